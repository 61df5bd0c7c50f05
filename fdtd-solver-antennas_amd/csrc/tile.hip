// tile.hip — one-pass leapfrog with an LDS-shared E tile (overlapped 3-D tiling).
//
// Same goal as fused.hip (one sweep over memory per timestep: 48 B/cell of field traffic instead of 72)
// but neighbours are shared through LDS instead of being recomputed per thread, which keeps the register
// count near the two-pass E kernel's (profiles/r01/pmc_fused_vs_twopass_NS.json: the register-recompute
// version needs 221 VGPRs -> 2 waves/SIMD and loses on occupancy although it moves 25 % fewer bytes).
//
// A block of 16 x 8 x 8 threads covers 16 groups (64 cells) in x, 8 rows, 8 planes.  EVERY thread computes
// the new voltages of its group (E half-step, exactly the two-pass arithmetic) and puts them into an LDS
// tile; after one barrier the 15 x 7 x 7 OWNER threads do the H half-step, reading V'(j+1), V'(k+1), V'(i0+4)
// from the tile.  Tiles overlap by one thread in each direction (the high-side threads only feed their
// neighbours), so there is no halo special-casing: all threads run the same code.  Results go to the
// ping-pong buffers of the fused path (V, I, psi_E double-buffered; psi_H in place).
//
// Replaces, like kernels.hip, the stepping inside FDTD.Run(...) of the reference's external engine
// (antenna_sim/solver_fdtd_openems_fixed.py:280).  Single slab, class operator, no Mur.
#include <vector>

#include "kernel_common.hpp"

namespace {

constexpr int TX = 16;   // threads along x (64 cells); TY x TZ is a template parameter: 8x8 (1024 thr), 4x8, 8x4 (512), 4x4 (256)

template <int COEF>
__device__ __forceinline__ int4 tcls4(const uchar4 c, const int comp) {
  return COEF == 2 ? make_int4(3 * c.x + comp, 3 * c.y + comp, 3 * c.z + comp, 3 * c.w + comp) : make_int4(c.x, c.y, c.z, c.w);
}

__device__ __forceinline__ float4 tvnew4(const float2* lut, const int4 ci, const float4 ex, const float myz, const float4 v,
                                         const float4 d1, const float4 d2) {
  const float2 l0 = lut[ci.x], l1 = lut[ci.y], l2 = lut[ci.z], l3 = lut[ci.w];
  return upd4(make_float4(l0.x, l1.x, l2.x, l3.x), v,
              make_float4(l0.y * (ex.x * myz), l1.y * (ex.y * myz), l2.y * (ex.z * myz), l3.y * (ex.w * myz)), d1, d2);
}

// psi' = b*psi + c*d ; d <- d/kappa + psi' ; psi' stored to `pout` by owners only
__device__ __forceinline__ void tcp_row4(float4& d, const float* pin, float* pout, const bool store, float b, float c, float ik) {
  float4 ps = ld4(pin);
  ps.x = __builtin_fmaf(b, ps.x, c * d.x);
  ps.y = __builtin_fmaf(b, ps.y, c * d.y);
  ps.z = __builtin_fmaf(b, ps.z, c * d.z);
  ps.w = __builtin_fmaf(b, ps.w, c * d.w);
  if (store) st4(pout, ps);
  d.x = __builtin_fmaf(ik, d.x, ps.x);
  d.y = __builtin_fmaf(ik, d.y, ps.y);
  d.z = __builtin_fmaf(ik, d.z, ps.z);
  d.w = __builtin_fmaf(ik, d.w, ps.w);
}

// x-directed layers with the float4-aligned slot layout: per-cell coefficient vectors
__device__ __forceinline__ void tcp_x4(float4& d, const float* pin, float* pout, const bool store, const float4 b, const float4 c,
                                       const float4 ik) {
  float4 ps = ld4(pin);
  ps.x = __builtin_fmaf(b.x, ps.x, c.x * d.x);
  ps.y = __builtin_fmaf(b.y, ps.y, c.y * d.y);
  ps.z = __builtin_fmaf(b.z, ps.z, c.z * d.z);
  ps.w = __builtin_fmaf(b.w, ps.w, c.w * d.w);
  if (store) st4(pout, ps);
  d.x = __builtin_fmaf(ik.x, d.x, ps.x);
  d.y = __builtin_fmaf(ik.y, d.y, ps.y);
  d.z = __builtin_fmaf(ik.z, d.z, ps.z);
  d.w = __builtin_fmaf(ik.w, d.w, ps.w);
}

template <int COEF, bool PML, int TY, int TZ>
__global__ __launch_bounds__(TX * TY * TZ) void k_step_tile(const DevParams p, const long long step, const int extra, const int ntx,
                                                             const int nty, const int ntz) {
  constexpr int NT = TX * TY * TZ;
  constexpr int LY = TY == 8 ? 3 : 2;     // log2(TY)
  __shared__ float4 s_v[3][TZ][TY][TX];   // new voltages of the tile (48 KiB at 8x8)
  __shared__ float2 s_lut[COEF == 2 ? 768 : 256];
  __shared__ SrcStage s_src;
  __shared__ double s_red[FDTD_BLOCK];
  // probe block: both probe kinds of step-1, reduced by the first FDTD_BLOCK threads
  const bool probe_blk = extra && blockIdx.x == gridDim.x - 1;
  if (probe_blk) {
    // all NT threads walk through the barriers of probe_block; only the first FDTD_BLOCK contribute
    for (int kind = 0; kind < 2; ++kind) {
      const long long st = step - 1;
      if (st < 0 || st >= p.max_steps) break;
      for (int q = 0; q < p.nprobe; ++q) {
        const DevProbe pr = p.probes[q];
        if (pr.kind != kind) continue;
        double s = 0.0;
        if (threadIdx.x < FDTD_BLOCK)
          for (int e = threadIdx.x; e < pr.n; e += FDTD_BLOCK) {
            const float* F = (kind == FDTD_KIND_V ? p.V[pr.comp[e]] : p.I[pr.comp[e]]);
            s = fma((double)pr.w[e], (double)F[pr.off[e]], s);
          }
        if (threadIdx.x < FDTD_BLOCK) s_red[threadIdx.x] = s;
        __syncthreads();
        for (int w = FDTD_BLOCK / 2; w > 0; w >>= 1) {
          if ((int)threadIdx.x < w) s_red[threadIdx.x] += s_red[threadIdx.x + w];
          __syncthreads();
        }
        if (threadIdx.x == 0) pr.series[st] = s_red[0];
        __syncthreads();
      }
    }
    return;
  }

  // ---- tile decode: XCD-aware remap, then z fastest (z-halo rows are the freshest in L2), x, y ----
  const unsigned nb = gridDim.x - (unsigned)extra, b = blockIdx.x;
  const unsigned q8 = nb >> 3, r8 = nb & 7u, xcd = b & 7u;
  const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int bz = (int)(v % (unsigned)ntz);
  const int bx = (int)((v / (unsigned)ntz) % (unsigned)ntx);
  const int by = (int)(v / ((unsigned)ntz * (unsigned)ntx));
  const int tx = threadIdx.x & (TX - 1), ty = (threadIdx.x >> 4) & (TY - 1), tz = threadIdx.x >> (4 + LY);
  const int i0 = (bx * (TX - 1) + tx) * 4, j = by * (TY - 1) + ty, k = bz * (TZ - 1) + tz;
  const bool inside = i0 < p.P && j < p.ny && k < p.nk;
  const bool owner = inside && tx < TX - 1 && ty < TY - 1 && tz < TZ - 1;

  for (int q = threadIdx.x; q < p.lut_n; q += NT) s_lut[q] = p.lut[q];
  int nsrc_t = 0;
  if (p.nsrc > 0) {
    const int2 srng = p.src_rng3[v];
    nsrc_t = min(srng.y - srng.x, FDTD_BLOCK);
    if (threadIdx.x < FDTD_BLOCK) stage_sources(p, p.src_ids3, srng.x, nsrc_t, step, s_src);
  }
  __syncthreads();

  const int P = p.P, PL = p.plane;
  const int off = k * PL + j * P + i0;
  float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vy = vx, vz = vx;
  float4 ix = vx, iy = vx, iz = vx;
  int sy = -1, sz = -1, sx = -1;
  if (inside) {
    // ================= E half-step (two-pass arithmetic) =================
    ix = ld4(p.I[0] + off); iy = ld4(p.I[1] + off); iz = ld4(p.I[2] + off);
    const float4 iz_jm = ld4(p.I[2] + off - P), ix_jm = ld4(p.I[0] + off - P);
    const float4 iy_km = ld4(p.I[1] + off - PL), ix_km = ld4(p.I[0] + off - PL);
    const float iz_im = p.I[2][off - 1], iy_im = p.I[1][off - 1];
    vx = ld4(p.V[0] + off); vy = ld4(p.V[1] + off); vz = ld4(p.V[2] + off);
    float4 dx1 = sub4(iz, iz_jm), dx2 = sub4(iy, iy_km);
    float4 dy1 = sub4(ix, ix_km);
    float4 dy2 = make_float4(iz.x - iz_im, iz.y - iz.x, iz.z - iz.y, iz.w - iz.z);
    float4 dz1 = make_float4(iy.x - iy_im, iy.y - iy.x, iy.z - iy.y, iy.w - iy.z);
    float4 dz2 = sub4(ix, ix_jm);
    if (PML) {
      sy = pml_slot(p, 1, j); sz = pml_slot(p, 2, k); sx = pml_slot(p, 0, i0);
      if (sy >= 0) {
        const float b_ = p.cp[1][0][0][j], c_ = p.cp[1][0][1][j], ik = p.cp[1][0][2][j];
        const int o = (k * p.nslot[1] + sy) * P + i0;
        tcp_row4(dx1, p.psiE[0][0] + o, p.psiEn[0][0] + o, owner, b_, c_, ik);
        tcp_row4(dz2, p.psiE[2][1] + o, p.psiEn[2][1] + o, owner, b_, c_, ik);
      }
      if (sz >= 0) {
        const float b_ = p.cp[2][0][0][k], c_ = p.cp[2][0][1][k], ik = p.cp[2][0][2][k];
        const int o = (sz * p.ny + j) * P + i0;
        tcp_row4(dx2, p.psiE[0][1] + o, p.psiEn[0][1] + o, owner, b_, c_, ik);
        tcp_row4(dy1, p.psiE[1][0] + o, p.psiEn[1][0] + o, owner, b_, c_, ik);
      }
      if (sx >= 0) {
        const float4 b_ = ld4(p.cp[0][0][0] + i0), c_ = ld4(p.cp[0][0][1] + i0), ik = ld4(p.cp[0][0][2] + i0);
        const int o = (k * p.ny + j) * p.nslot[0] + sx;
        tcp_x4(dy2, p.psiE[1][1] + o, p.psiEn[1][1] + o, owner, b_, c_, ik);
        tcp_x4(dz1, p.psiE[2][0] + o, p.psiEn[2][0] + o, owner, b_, c_, ik);
      }
    }
    uchar4 c0, c1, c2;
    if (COEF == 2) {
      c0 = c1 = c2 = *reinterpret_cast<const uchar4*>(p.ecls + off);
    } else {
      c0 = *reinterpret_cast<const uchar4*>(p.ecls + off);
      c1 = *reinterpret_cast<const uchar4*>(p.ecls + p.nloc + off);
      c2 = *reinterpret_cast<const uchar4*>(p.ecls + 2 * p.nloc + off);
    }
    const float4 ex0 = ld4(p.emet[0][0] + i0), ex1 = ld4(p.emet[1][0] + i0), ex2 = ld4(p.emet[2][0] + i0);
    vx = tvnew4(s_lut, tcls4<COEF>(c0, 0), ex0, p.emet[0][1][j] * p.emet[0][2][k], vx, dx1, dx2);
    vy = tvnew4(s_lut, tcls4<COEF>(c1, 1), ex1, p.emet[1][1][j] * p.emet[1][2][k], vy, dy1, dy2);
    vz = tvnew4(s_lut, tcls4<COEF>(c2, 2), ex2, p.emet[2][1][j] * p.emet[2][2][k], vz, dz1, dz2);
    if (nsrc_t > 0) {
      apply_staged(s_src, nsrc_t, 0, off, vx);
      apply_staged(s_src, nsrc_t, 1, off, vy);
      apply_staged(s_src, nsrc_t, 2, off, vz);
    }
    if (owner) {
      st4(p.Vn[0] + off, vx);
      st4(p.Vn[1] + off, vy);
      st4(p.Vn[2] + off, vz);
    }
  }
  // H-side loads are issued BEFORE the barrier so that their latency overlaps the tile hand-off
  float4 qy1 = vx, qy2 = vx, qz1 = vx, qz2 = vx, qx1 = vx, qx2 = vx, hxb = vx, hxc = vx, hxk = vx;
  float yb = 0.f, yc = 0.f, yk = 1.f, zb = 0.f, zc = 0.f, zk = 1.f;
  float4 h0 = vx, h1 = vx, h2 = vx;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  int oy = 0, oz = 0, ox = 0;
  if (owner) {
    if (PML) {
      if (sy >= 0) {
        oy = (k * p.nslot[1] + sy) * P + i0;
        qy1 = ld4(p.psiH[0][0] + oy); qy2 = ld4(p.psiH[2][1] + oy);
        yb = p.cp[1][1][0][j]; yc = p.cp[1][1][1][j]; yk = p.cp[1][1][2][j];
      }
      if (sz >= 0) {
        oz = (sz * p.ny + j) * P + i0;
        qz1 = ld4(p.psiH[0][1] + oz); qz2 = ld4(p.psiH[1][0] + oz);
        zb = p.cp[2][1][0][k]; zc = p.cp[2][1][1][k]; zk = p.cp[2][1][2][k];
      }
      if (sx >= 0) {
        ox = (k * p.ny + j) * p.nslot[0] + sx;
        qx1 = ld4(p.psiH[1][1] + ox); qx2 = ld4(p.psiH[2][0] + ox);
        hxb = ld4(p.cp[0][1][0] + i0); hxc = ld4(p.cp[0][1][1] + i0); hxk = ld4(p.cp[0][1][2] + i0);
      }
    }
    h0 = ld4(p.hmet[0][0] + i0); h1 = ld4(p.hmet[1][0] + i0); h2 = ld4(p.hmet[2][0] + i0);
    m0 = p.hmet[0][1][j] * p.hmet[0][2][k];
    m1 = p.hmet[1][1][j] * p.hmet[1][2][k];
    m2 = p.hmet[2][1][j] * p.hmet[2][2][k];
  }
  s_v[0][tz][ty][tx] = vx;
  s_v[1][tz][ty][tx] = vy;
  s_v[2][tz][ty][tx] = vz;
  __syncthreads();
  if (!owner) return;

  // ================= H half-step on the owner cells, neighbours from the LDS tile =================
  const float4 vz_jp = s_v[2][tz][ty + 1][tx], vx_jp = s_v[0][tz][ty + 1][tx];
  const float4 vy_kp = s_v[1][tz + 1][ty][tx], vx_kp = s_v[0][tz + 1][ty][tx];
  const float vz_ip = s_v[2][tz][ty][tx + 1].x, vy_ip = s_v[1][tz][ty][tx + 1].x;
  float4 hx1 = sub4(vz, vz_jp), hx2 = sub4(vy, vy_kp);
  float4 hy1 = sub4(vx, vx_kp);
  float4 hy2 = make_float4(vz.x - vz.y, vz.y - vz.z, vz.z - vz.w, vz.w - vz_ip);
  float4 hz1 = make_float4(vy.x - vy.y, vy.y - vy.z, vy.z - vy.w, vy.w - vy_ip);
  float4 hz2 = sub4(vx, vx_jp);
  auto cpr = [](float4& d, const float4 ps, float b, float c, float ik) {
    const float4 n = make_float4(__builtin_fmaf(b, ps.x, c * d.x), __builtin_fmaf(b, ps.y, c * d.y), __builtin_fmaf(b, ps.z, c * d.z),
                                 __builtin_fmaf(b, ps.w, c * d.w));
    d = make_float4(__builtin_fmaf(ik, d.x, n.x), __builtin_fmaf(ik, d.y, n.y), __builtin_fmaf(ik, d.z, n.z), __builtin_fmaf(ik, d.w, n.w));
    return n;
  };
  auto cpx = [](float4& d, const float4 ps, const float4 b, const float4 c, const float4 ik) {
    const float4 n = make_float4(__builtin_fmaf(b.x, ps.x, c.x * d.x), __builtin_fmaf(b.y, ps.y, c.y * d.y), __builtin_fmaf(b.z, ps.z, c.z * d.z),
                                 __builtin_fmaf(b.w, ps.w, c.w * d.w));
    d = make_float4(__builtin_fmaf(ik.x, d.x, n.x), __builtin_fmaf(ik.y, d.y, n.y), __builtin_fmaf(ik.z, d.z, n.z), __builtin_fmaf(ik.w, d.w, n.w));
    return n;
  };
  if (PML) {
    if (sy >= 0) { st4(p.psiH[0][0] + oy, cpr(hx1, qy1, yb, yc, yk)); st4(p.psiH[2][1] + oy, cpr(hz2, qy2, yb, yc, yk)); }
    if (sz >= 0) { st4(p.psiH[0][1] + oz, cpr(hx2, qz1, zb, zc, zk)); st4(p.psiH[1][0] + oz, cpr(hy1, qz2, zb, zc, zk)); }
    if (sx >= 0) { st4(p.psiH[1][1] + ox, cpx(hy2, qx1, hxb, hxc, hxk)); st4(p.psiH[2][0] + ox, cpx(hz1, qx2, hxb, hxc, hxk)); }
  }
  st4(p.In[0] + off, make_float4(ix.x + (h0.x * m0) * (hx1.x - hx2.x), ix.y + (h0.y * m0) * (hx1.y - hx2.y),
                                  ix.z + (h0.z * m0) * (hx1.z - hx2.z), ix.w + (h0.w * m0) * (hx1.w - hx2.w)));
  st4(p.In[1] + off, make_float4(iy.x + (h1.x * m1) * (hy1.x - hy2.x), iy.y + (h1.y * m1) * (hy1.y - hy2.y),
                                  iy.z + (h1.z * m1) * (hy1.z - hy2.z), iy.w + (h1.w * m1) * (hy1.w - hy2.w)));
  st4(p.In[2] + off, make_float4(iz.x + (h2.x * m2) * (hz1.x - hz2.x), iz.y + (h2.y * m2) * (hz1.y - hz2.y),
                                  iz.z + (h2.z * m2) * (hz1.z - hz2.z), iz.w + (h2.w * m2) * (hz1.w - hz2.w)));
}

}  // namespace

static void tile_shape(const fdtd_ctx* c, int& ty, int& tz) {
  ty = (c->tile_shape >> 4) & 0xF;
  tz = c->tile_shape & 0xF;
}

void tile_counts(const fdtd_ctx* c, int& ntx, int& nty, int& ntz) {
  int TY, TZ;
  tile_shape(c, TY, TZ);
  ntx = (c->p.P4 + (TX - 1) - 1) / (TX - 1);
  nty = (c->p.ny + (TY - 1) - 1) / (TY - 1);
  ntz = (c->p.nk + (TZ - 1) - 1) / (TZ - 1);
}

// tiles (linear id = (by*ntx + bx)*ntz + bz) whose threads compute the cell group (gx, j, k)
void tiles_of_cell(const fdtd_ctx* c, int gx, int j, int k, std::vector<int>& out) {
  int ntx, nty, ntz, TY, TZ;
  tile_counts(c, ntx, nty, ntz);
  tile_shape(c, TY, TZ);
  auto cover = [](int q, int own, int nt, int* r) {   // tiles along one axis containing index q (own = threads-1)
    int n = 0;
    const int t = q / own;
    if (t < nt) r[n++] = t;
    if (q % own == 0 && t > 0) r[n++] = t - 1;        // q is the high-side (feeder) thread of the previous tile
    return n;
  };
  int rx[2], ry[2], rz[2];
  const int nx_ = cover(gx, TX - 1, ntx, rx), ny_ = cover(j, TY - 1, nty, ry), nz_ = cover(k, TZ - 1, ntz, rz);
  for (int a = 0; a < ny_; ++a)
    for (int b = 0; b < nx_; ++b)
      for (int d = 0; d < nz_; ++d) out.push_back((ry[a] * ntx + rx[b]) * ntz + rz[d]);
}

template <int COEF, bool PML>
static void launch_shape(fdtd_ctx* c, dim3 grid, long long step, int extra, int ntx, int nty, int ntz, hipStream_t s) {
  switch (c->tile_shape) {
    case 0x88: hipLaunchKernelGGL((k_step_tile<COEF, PML, 8, 8>), grid, dim3(TX * 64), 0, s, c->p, step, extra, ntx, nty, ntz); break;
    case 0x48: hipLaunchKernelGGL((k_step_tile<COEF, PML, 4, 8>), grid, dim3(TX * 32), 0, s, c->p, step, extra, ntx, nty, ntz); break;
    case 0x84: hipLaunchKernelGGL((k_step_tile<COEF, PML, 8, 4>), grid, dim3(TX * 32), 0, s, c->p, step, extra, ntx, nty, ntz); break;
    default:   hipLaunchKernelGGL((k_step_tile<COEF, PML, 4, 4>), grid, dim3(TX * 16), 0, s, c->p, step, extra, ntx, nty, ntz); break;
  }
}

void launch_step_tile(fdtd_ctx* c, long long step, bool probe_block, hipStream_t s) {
  int ntx, nty, ntz;
  tile_counts(c, ntx, nty, ntz);
  const int extra = probe_block ? 1 : 0;
  const dim3 grid((unsigned)(ntx * nty * ntz + extra));
  if (c->packed_op) {
    if (c->have_cpml) launch_shape<2, true>(c, grid, step, extra, ntx, nty, ntz, s);
    else launch_shape<2, false>(c, grid, step, extra, ntx, nty, ntz, s);
  } else {
    if (c->have_cpml) launch_shape<1, true>(c, grid, step, extra, ntx, nty, ntz, s);
    else launch_shape<1, false>(c, grid, step, extra, ntx, nty, ntz, s);
  }
}
