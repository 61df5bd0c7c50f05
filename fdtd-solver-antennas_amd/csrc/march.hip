// march.hip — one-pass leapfrog, z-marching xy tiles with an LDS-shared E plane.
//
// What the measurements of round 1 ask for: the byte saving of a fused E+H sweep (48 instead of
// 72 B/cell of field traffic) WITHOUT the register cost of per-thread recomputation (fused.hip: 221 VGPRs) and
// WITHOUT 3-D overlap (tile.hip: 28 % feeder threads whose halo reads miss L2).
//
// A block of 16 x 16 threads owns a 15 x 15 patch of cell groups (60 x 15 cells) in the xy plane and marches
// through a chunk of z planes.  Per plane k+1 every thread computes the new voltages E'(k+1) of its group
// (two-pass arithmetic; the k-1 neighbour is the plane it carried in registers), puts them into one of two LDS
// plane buffers and — owners only — into the next-step buffer; then the owners update the currents of plane k
// from E'(k) (own: registers; j+1 / i0+4 neighbours: LDS plane of the previous iteration), E'(k+1) (registers)
// and I(k) (registers).  One barrier per plane.  Redundancy: one feeder row/column per tile (12 %) and one extra
// E' plane per z-chunk; per plane a thread loads 8 float4 of fields instead of 13 + 13 in the two passes.
// Results go to the ping-pong buffers of the one-pass family (V, I, psi_E double-buffered; psi_H in place).
//
// Replaces, like kernels.hip, the stepping inside FDTD.Run(...) of the reference's external engine
// (antenna_sim/solver_fdtd_openems_fixed.py:280).  Single slab, class operator, no Mur.
#include <vector>

#include "kernel_common.hpp"

namespace {

constexpr int MX = 16, MY = 16, MT = MX * MY;   // threads; owners 15 x 15

template <int COEF>
__device__ __forceinline__ int4 mcls4(const uchar4 c, const int comp) {
  return COEF == 2 ? make_int4(3 * c.x + comp, 3 * c.y + comp, 3 * c.z + comp, 3 * c.w + comp) : make_int4(c.x, c.y, c.z, c.w);
}

__device__ __forceinline__ float4 mvnew4(const float2* lut, const int4 ci, const float4 ex, const float myz, const float4 v,
                                         const float4 d1, const float4 d2) {
  const float2 l0 = lut[ci.x], l1 = lut[ci.y], l2 = lut[ci.z], l3 = lut[ci.w];
  return upd4(make_float4(l0.x, l1.x, l2.x, l3.x), v,
              make_float4(l0.y * (ex.x * myz), l1.y * (ex.y * myz), l2.y * (ex.z * myz), l3.y * (ex.w * myz)), d1, d2);
}

// psi' = b*psi + c*d ; d <- d/kappa + psi'   (psi already in registers; returns psi')
__device__ __forceinline__ float4 cp4(float4& d, float4 ps, const float b, const float c, const float ik) {
  ps.x = __builtin_fmaf(b, ps.x, c * d.x);
  ps.y = __builtin_fmaf(b, ps.y, c * d.y);
  ps.z = __builtin_fmaf(b, ps.z, c * d.z);
  ps.w = __builtin_fmaf(b, ps.w, c * d.w);
  d.x = __builtin_fmaf(ik, d.x, ps.x);
  d.y = __builtin_fmaf(ik, d.y, ps.y);
  d.z = __builtin_fmaf(ik, d.z, ps.z);
  d.w = __builtin_fmaf(ik, d.w, ps.w);
  return ps;
}

__device__ __forceinline__ float4 cp4x(float4& d, float4 ps, const float4 b, const float4 c, const float4 ik) {
  ps.x = __builtin_fmaf(b.x, ps.x, c.x * d.x);
  ps.y = __builtin_fmaf(b.y, ps.y, c.y * d.y);
  ps.z = __builtin_fmaf(b.z, ps.z, c.z * d.z);
  ps.w = __builtin_fmaf(b.w, ps.w, c.w * d.w);
  d.x = __builtin_fmaf(ik.x, d.x, ps.x);
  d.y = __builtin_fmaf(ik.y, d.y, ps.y);
  d.z = __builtin_fmaf(ik.z, d.z, ps.z);
  d.w = __builtin_fmaf(ik.w, d.w, ps.w);
  return ps;
}

// Everything one E plane reads from memory.  The loop issues ALL loads of an iteration (this and HLd) before
// the first dependent instruction: with 2-3 waves per SIMD a z-marching thread cannot rely on other waves to
// cover three serial round trips (fields -> psi_E -> psi_H) per plane.
struct ELd {
  float4 ix, iy, iz, iz_jm, ix_jm, vx, vy, vz;
  float iz_im, iy_im;
  uchar4 c0, c1, c2;
  float4 psy0, psy1, psz0, psz1, psx0, psx1;
  float yb, yc, yk, zb, zc, zk, my0, my1, my2, mz0, mz1, mz2;
  int sz, off, oy, oz, ox;
};

template <int COEF, bool PML>
__device__ __forceinline__ void e_load(const DevParams& p, const int k, const int j, const int i0, const int sy, const int sx, ELd& L) {
  const int P = p.P;
  const int off = k * p.plane + j * P + i0;
  L.off = off;
  L.ix = ld4(p.I[0] + off); L.iy = ld4(p.I[1] + off); L.iz = ld4(p.I[2] + off);
  L.iz_jm = ld4(p.I[2] + off - P); L.ix_jm = ld4(p.I[0] + off - P);
  L.iz_im = p.I[2][off - 1]; L.iy_im = p.I[1][off - 1];
  L.vx = ld4(p.V[0] + off); L.vy = ld4(p.V[1] + off); L.vz = ld4(p.V[2] + off);
  if (COEF == 2) {
    L.c0 = L.c1 = L.c2 = *reinterpret_cast<const uchar4*>(p.ecls + off);
  } else {
    L.c0 = *reinterpret_cast<const uchar4*>(p.ecls + off);
    L.c1 = *reinterpret_cast<const uchar4*>(p.ecls + p.nloc + off);
    L.c2 = *reinterpret_cast<const uchar4*>(p.ecls + 2 * p.nloc + off);
  }
  L.my0 = p.emet[0][1][j]; L.my1 = p.emet[1][1][j]; L.my2 = p.emet[2][1][j];
  L.mz0 = p.emet[0][2][k]; L.mz1 = p.emet[1][2][k]; L.mz2 = p.emet[2][2][k];
  L.sz = -1;
  if (PML) {
    if (sy >= 0) {
      L.yb = p.cp[1][0][0][j]; L.yc = p.cp[1][0][1][j]; L.yk = p.cp[1][0][2][j];
      L.oy = (k * p.nslot[1] + sy) * P + i0;
      L.psy0 = ld4(p.psiE[0][0] + L.oy); L.psy1 = ld4(p.psiE[2][1] + L.oy);
    }
    L.sz = pml_slot(p, 2, k);
    if (L.sz >= 0) {
      L.zb = p.cp[2][0][0][k]; L.zc = p.cp[2][0][1][k]; L.zk = p.cp[2][0][2][k];
      L.oz = (L.sz * p.ny + j) * P + i0;
      L.psz0 = ld4(p.psiE[0][1] + L.oz); L.psz1 = ld4(p.psiE[1][0] + L.oz);
    }
    if (sx >= 0) {
      L.ox = (k * p.ny + j) * p.nslot[0] + sx;
      L.psx0 = ld4(p.psiE[1][1] + L.ox); L.psx1 = ld4(p.psiE[2][0] + L.ox);
    }
  }
}

// E half-step of one group from its loaded operands.  In: I(k-1) own x,y.  Out: the new voltages.
template <int COEF, bool PML>
__device__ __forceinline__ void e_compute(const DevParams& p, const float2* s_lut, const SrcStage& s_src, const int nsrc_t,
                                          const int i0, const int sy, const int sx, const bool store, const ELd& L, const float4 ix_km,
                                          const float4 iy_km, float4& vx, float4& vy, float4& vz) {
  float4 dx1 = sub4(L.iz, L.iz_jm), dx2 = sub4(L.iy, iy_km);
  float4 dy1 = sub4(L.ix, ix_km);
  float4 dy2 = make_float4(L.iz.x - L.iz_im, L.iz.y - L.iz.x, L.iz.z - L.iz.y, L.iz.w - L.iz.z);
  float4 dz1 = make_float4(L.iy.x - L.iy_im, L.iy.y - L.iy.x, L.iy.z - L.iy.y, L.iy.w - L.iy.z);
  float4 dz2 = sub4(L.ix, L.ix_jm);
  if (PML) {
    if (sy >= 0) {
      const float4 a = cp4(dx1, L.psy0, L.yb, L.yc, L.yk), b = cp4(dz2, L.psy1, L.yb, L.yc, L.yk);
      if (store) { st4(p.psiEn[0][0] + L.oy, a); st4(p.psiEn[2][1] + L.oy, b); }
    }
    if (L.sz >= 0) {
      const float4 a = cp4(dx2, L.psz0, L.zb, L.zc, L.zk), b = cp4(dy1, L.psz1, L.zb, L.zc, L.zk);
      if (store) { st4(p.psiEn[0][1] + L.oz, a); st4(p.psiEn[1][0] + L.oz, b); }
    }
    if (sx >= 0) {
      const float4 xb = ld4(p.cp[0][0][0] + i0), xc = ld4(p.cp[0][0][1] + i0), xk = ld4(p.cp[0][0][2] + i0);   // L1-resident tables
      const float4 a = cp4x(dy2, L.psx0, xb, xc, xk), b = cp4x(dz1, L.psx1, xb, xc, xk);
      if (store) { st4(p.psiEn[1][1] + L.ox, a); st4(p.psiEn[2][0] + L.ox, b); }
    }
  }
  const float4 ex0 = ld4(p.emet[0][0] + i0), ex1 = ld4(p.emet[1][0] + i0), ex2 = ld4(p.emet[2][0] + i0);
  vx = mvnew4(s_lut, mcls4<COEF>(L.c0, 0), ex0, L.my0 * L.mz0, L.vx, dx1, dx2);
  vy = mvnew4(s_lut, mcls4<COEF>(L.c1, 1), ex1, L.my1 * L.mz1, L.vy, dy1, dy2);
  vz = mvnew4(s_lut, mcls4<COEF>(L.c2, 2), ex2, L.my2 * L.mz2, L.vz, dz1, dz2);
  if (nsrc_t > 0) {
    apply_staged(s_src, nsrc_t, 0, L.off, vx);
    apply_staged(s_src, nsrc_t, 1, L.off, vy);
    apply_staged(s_src, nsrc_t, 2, L.off, vz);
  }
  if (store) {
    st4(p.Vn[0] + L.off, vx);
    st4(p.Vn[1] + L.off, vy);
    st4(p.Vn[2] + L.off, vz);
  }
}

// Everything the H update of one owner group reads from memory besides E' (registers / LDS).
struct HLd {
  float4 psy0, psy1, psz0, psz1, psx0, psx1;
  float yb, yc, yk, zb, zc, zk, my0, my1, my2, mz0, mz1, mz2;
  int sz, oy, oz, ox;
};

template <bool PML>
__device__ __forceinline__ void h_load(const DevParams& p, const int k, const int j, const int i0, const int sy, const int sx, HLd& H) {
  const int P = p.P;
  H.my0 = p.hmet[0][1][j]; H.my1 = p.hmet[1][1][j]; H.my2 = p.hmet[2][1][j];
  H.mz0 = p.hmet[0][2][k]; H.mz1 = p.hmet[1][2][k]; H.mz2 = p.hmet[2][2][k];
  H.sz = -1;
  if (PML) {
    if (sy >= 0) {
      H.yb = p.cp[1][1][0][j]; H.yc = p.cp[1][1][1][j]; H.yk = p.cp[1][1][2][j];
      H.oy = (k * p.nslot[1] + sy) * P + i0;
      H.psy0 = ld4(p.psiH[0][0] + H.oy); H.psy1 = ld4(p.psiH[2][1] + H.oy);
    }
    H.sz = pml_slot(p, 2, k);
    if (H.sz >= 0) {
      H.zb = p.cp[2][1][0][k]; H.zc = p.cp[2][1][1][k]; H.zk = p.cp[2][1][2][k];
      H.oz = (H.sz * p.ny + j) * P + i0;
      H.psz0 = ld4(p.psiH[0][1] + H.oz); H.psz1 = ld4(p.psiH[1][0] + H.oz);
    }
    if (sx >= 0) {
      H.ox = (k * p.ny + j) * p.nslot[0] + sx;
      H.psx0 = ld4(p.psiH[1][1] + H.ox); H.psx1 = ld4(p.psiH[2][0] + H.ox);
    }
  }
}

template <int COEF, bool PML>
__global__ __launch_bounds__(MT) void k_step_march(const DevParams p, const long long step, const int extra, const int ntx,
                                                   const int nty, const int kc) {
  __shared__ float4 s_e[2][3][MY][MX];   // E' of plane k and k+1 (double buffer): 24 KiB
  __shared__ float2 s_lut[COEF == 2 ? 768 : 256];
  __shared__ SrcStage s_src;
  __shared__ double s_red[FDTD_BLOCK];
  if (extra && blockIdx.x == gridDim.x - 1) {   // probe block: both probe kinds of the step just finished
    probe_block(p, FDTD_KIND_V, step - 1, s_red);
    probe_block(p, FDTD_KIND_I, step - 1, s_red);
    return;
  }
  // ---- block decode: XCD-aware remap; x fastest, then y, then z-chunk (neighbouring tiles march together) ----
  const unsigned nb = gridDim.x - (unsigned)extra, b = blockIdx.x;
  const unsigned q8 = nb >> 3, r8 = nb & 7u, xcd = b & 7u;
  const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int bx = (int)(v % (unsigned)ntx);
  const int by = (int)((v / (unsigned)ntx) % (unsigned)nty);
  const int bz = (int)(v / ((unsigned)ntx * (unsigned)nty));
  const int tx = threadIdx.x & (MX - 1), ty = threadIdx.x >> 4;
  const int i0 = (bx * (MX - 1) + tx) * 4, j = by * (MY - 1) + ty;
  const int kb = bz * kc, ke = min(kb + kc, p.nk);
  const bool inside = i0 < p.P && j < p.ny;
  const bool owner = inside && tx < MX - 1 && ty < MY - 1;

  for (int q = threadIdx.x; q < p.lut_n; q += MT) s_lut[q] = p.lut[q];
  int nsrc_t = 0;
  if (p.nsrc > 0) {   // sources inside this tile column (xy footprint incl. feeders, planes kb..ke), staged once
    const int2 srng = p.src_rng4[v];
    nsrc_t = min(srng.y - srng.x, FDTD_BLOCK);
    stage_sources(p, p.src_ids4, srng.x, nsrc_t, step, s_src);
  }
  __syncthreads();

  const int PL = p.plane;
  const int sy = (PML && inside) ? pml_slot(p, 1, j) : -1;
  const int sx = (PML && inside) ? pml_slot(p, 0, i0) : -1;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ixk = zero4, iyk = zero4, izk = zero4;   // I(k) own
  float4 exk = zero4, eyk = zero4, ezk = zero4;   // E'(k) own

  // ---- prologue: E'(kb) ----
  if (inside) {
    const int offm = (kb - 1) * PL + j * p.P + i0;
    const float4 ix_km = ld4(p.I[0] + offm), iy_km = ld4(p.I[1] + offm);
    ELd L;
    e_load<COEF, PML>(p, kb, j, i0, sy, sx, L);
    e_compute<COEF, PML>(p, s_lut, s_src, nsrc_t, i0, sy, sx, owner, L, ix_km, iy_km, exk, eyk, ezk);
    ixk = L.ix; iyk = L.iy; izk = L.iz;
  }
  s_e[kb & 1][0][ty][tx] = exk;
  s_e[kb & 1][1][ty][tx] = eyk;
  s_e[kb & 1][2][ty][tx] = ezk;
  __syncthreads();

  for (int k = kb; k < ke; ++k) {
    // ---- all loads of this iteration first ----
    const bool do_e = inside && k + 1 < p.nk;   // the plane above the slab does not exist: zero (meets zero coefficients)
    ELd L;
    HLd H;
    if (do_e) e_load<COEF, PML>(p, k + 1, j, i0, sy, sx, L);
    if (owner) h_load<PML>(p, k, j, i0, sy, sx, H);
    // ---- E'(k+1) ----
    float4 ix1 = zero4, iy1 = zero4, iz1 = zero4, ex1 = zero4, ey1 = zero4, ez1 = zero4;
    if (do_e) {
      e_compute<COEF, PML>(p, s_lut, s_src, nsrc_t, i0, sy, sx, owner && (k + 1 < ke), L, ixk, iyk, ex1, ey1, ez1);
      ix1 = L.ix; iy1 = L.iy; iz1 = L.iz;
    }
    const int nbuf = (k + 1) & 1, cbuf = k & 1;
    s_e[nbuf][0][ty][tx] = ex1;
    s_e[nbuf][1][ty][tx] = ey1;
    s_e[nbuf][2][ty][tx] = ez1;
    // ---- H'(k) on the owner cells ----
    if (owner) {
      const int off = k * PL + j * p.P + i0;
      const float4 vz_jp = s_e[cbuf][2][ty + 1][tx], vx_jp = s_e[cbuf][0][ty + 1][tx];
      const float vz_ip = s_e[cbuf][2][ty][tx + 1].x, vy_ip = s_e[cbuf][1][ty][tx + 1].x;
      float4 hx1 = sub4(ezk, vz_jp), hx2 = sub4(eyk, ey1);
      float4 hy1 = sub4(exk, ex1);
      float4 hy2 = make_float4(ezk.x - ezk.y, ezk.y - ezk.z, ezk.z - ezk.w, ezk.w - vz_ip);
      float4 hz1 = make_float4(eyk.x - eyk.y, eyk.y - eyk.z, eyk.z - eyk.w, eyk.w - vy_ip);
      float4 hz2 = sub4(exk, vx_jp);
      if (PML) {
        if (sy >= 0) {
          st4(p.psiH[0][0] + H.oy, cp4(hx1, H.psy0, H.yb, H.yc, H.yk));
          st4(p.psiH[2][1] + H.oy, cp4(hz2, H.psy1, H.yb, H.yc, H.yk));
        }
        if (H.sz >= 0) {
          st4(p.psiH[0][1] + H.oz, cp4(hx2, H.psz0, H.zb, H.zc, H.zk));
          st4(p.psiH[1][0] + H.oz, cp4(hy1, H.psz1, H.zb, H.zc, H.zk));
        }
        if (sx >= 0) {
          const float4 xb = ld4(p.cp[0][1][0] + i0), xc = ld4(p.cp[0][1][1] + i0), xk = ld4(p.cp[0][1][2] + i0);
          st4(p.psiH[1][1] + H.ox, cp4x(hy2, H.psx0, xb, xc, xk));
          st4(p.psiH[2][0] + H.ox, cp4x(hz1, H.psx1, xb, xc, xk));
        }
      }
      const float m0 = H.my0 * H.mz0, m1 = H.my1 * H.mz1, m2 = H.my2 * H.mz2;
      const float4 h0 = ld4(p.hmet[0][0] + i0), h1 = ld4(p.hmet[1][0] + i0), h2 = ld4(p.hmet[2][0] + i0);
      st4(p.In[0] + off, make_float4(ixk.x + (h0.x * m0) * (hx1.x - hx2.x), ixk.y + (h0.y * m0) * (hx1.y - hx2.y),
                                      ixk.z + (h0.z * m0) * (hx1.z - hx2.z), ixk.w + (h0.w * m0) * (hx1.w - hx2.w)));
      st4(p.In[1] + off, make_float4(iyk.x + (h1.x * m1) * (hy1.x - hy2.x), iyk.y + (h1.y * m1) * (hy1.y - hy2.y),
                                      iyk.z + (h1.z * m1) * (hy1.z - hy2.z), iyk.w + (h1.w * m1) * (hy1.w - hy2.w)));
      st4(p.In[2] + off, make_float4(izk.x + (h2.x * m2) * (hz1.x - hz2.x), izk.y + (h2.y * m2) * (hz1.y - hz2.y),
                                      izk.z + (h2.z * m2) * (hz1.z - hz2.z), izk.w + (h2.w * m2) * (hz1.w - hz2.w)));
    }
    ixk = ix1; iyk = iy1; izk = iz1;
    exk = ex1; eyk = ey1; ezk = ez1;
    __syncthreads();   // plane k+1 complete in LDS; plane k's buffer free for plane k+2
  }
}

}  // namespace

static int march_kc(const fdtd_ctx* c) { return c->march_kc > 0 ? c->march_kc : 10; }

void march_counts(const fdtd_ctx* c, int& ntx, int& nty, int& ntz, int& kc) {
  kc = march_kc(c);
  ntx = (c->p.P4 + (MX - 1) - 1) / (MX - 1);
  nty = (c->p.ny + (MY - 1) - 1) / (MY - 1);
  ntz = (c->p.nk + kc - 1) / kc;
}

// tile columns (linear id = (bz*nty + by)*ntx + bx) whose threads compute cell group (gx, j) in plane k
void march_tiles_of_cell(const fdtd_ctx* c, int gx, int j, int k, std::vector<int>& out) {
  int ntx, nty, ntz, kc;
  march_counts(c, ntx, nty, ntz, kc);
  auto cover = [](int q, int own, int nt, int* r) {
    int n = 0;
    const int t = q / own;
    if (t < nt) r[n++] = t;
    if (q % own == 0 && t > 0) r[n++] = t - 1;   // q is the feeder of the previous tile
    return n;
  };
  int rx[2], ry[2], rz[2], nz_ = 0;
  const int nx_ = cover(gx, MX - 1, ntx, rx), ny_ = cover(j, MY - 1, nty, ry);
  rz[nz_++] = k / kc;
  if (k % kc == 0 && k > 0) rz[nz_++] = k / kc - 1;   // plane ke of the chunk below (computed redundantly there)
  for (int d = 0; d < nz_; ++d)
    for (int a = 0; a < ny_; ++a)
      for (int b = 0; b < nx_; ++b) out.push_back((rz[d] * nty + ry[a]) * ntx + rx[b]);
}

void launch_step_march(fdtd_ctx* c, long long step, bool probe_block, hipStream_t s) {
  int ntx, nty, ntz, kc;
  march_counts(c, ntx, nty, ntz, kc);
  const int extra = probe_block ? 1 : 0;
  const dim3 grid((unsigned)(ntx * nty * ntz + extra)), block(MT);
  if (c->packed_op) {
    if (c->have_cpml) hipLaunchKernelGGL((k_step_march<2, true>), grid, block, 0, s, c->p, step, extra, ntx, nty, kc);
    else hipLaunchKernelGGL((k_step_march<2, false>), grid, block, 0, s, c->p, step, extra, ntx, nty, kc);
  } else {
    if (c->have_cpml) hipLaunchKernelGGL((k_step_march<1, true>), grid, block, 0, s, c->p, step, extra, ntx, nty, kc);
    else hipLaunchKernelGGL((k_step_march<1, false>), grid, block, 0, s, c->p, step, extra, ntx, nty, kc);
  }
}
