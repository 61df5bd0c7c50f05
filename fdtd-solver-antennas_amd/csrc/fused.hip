// fused.hip — one-pass leapfrog: E half-step and H half-step of a timestep in ONE sweep over memory.
//
// The two-pass kernels (kernels.hip) already run at the chip's streaming ceiling for the bytes they move
// (profiles/r01: 246 MB per E launch at ~5.7-6.2 TB/s), so the only way up is fewer bytes: a two-pass
// leapfrog reads and writes every field twice per timestep (72 B/cell), a fused sweep once (48 B/cell).
//
// For a thread's four cells at (i0..i0+3, j, k) the H update needs the NEW voltages at (j+1), (k+1) and
// (i0+4) as well.  Instead of staging an E tile in LDS and synchronising, every thread recomputes those
// few neighbour values itself (same fmaf sequence => bit-identical to what the owning thread stores), so
// there is no barrier and the strip-major / XCD-aware mapping of the two-pass kernels is kept; the extra
// loads are L1/L2 hits on rows that the neighbouring threads load anyway.  Because neighbouring blocks
// still read time-level-n values of cells this block owns, results go to a second buffer set
// (ping-pong): V, I and the E-side CPML psi are double-buffered, psi_H is only touched by its owner and
// is updated in place.  Ghost planes, pads and table slack make every neighbour access in-bounds; values
// beyond the grid only ever meet zero coefficients.
//
// Single-slab, CPML/PEC scenes with a class-compressed operator; everything else (Mur, raw operator,
// multi-rank halo choreography) uses the two-pass path.  Replaces, like kernels.hip, the stepping inside
// FDTD.Run(...) of the reference's external engine (antenna_sim/solver_fdtd_openems_fixed.py:280).
#include "kernel_common.hpp"

namespace {

__device__ __forceinline__ int pml_slot_in(const DevParams& p, int a, int q, int n) {
  return q < n ? pml_slot(p, a, q) : -1;
}

// psi' = b*psi + c*d ; d <- d/kappa + psi'   (row-uniform coefficients); STORE writes psi' to the next buffer
template <bool STORE>
__device__ __forceinline__ void cp_row4(float4& d, const float* pin, float* pout, float b, float c, float ik) {
  float4 ps = ld4(pin);
  ps.x = __builtin_fmaf(b, ps.x, c * d.x);
  ps.y = __builtin_fmaf(b, ps.y, c * d.y);
  ps.z = __builtin_fmaf(b, ps.z, c * d.z);
  ps.w = __builtin_fmaf(b, ps.w, c * d.w);
  if (STORE) st4(pout, ps);
  d.x = __builtin_fmaf(ik, d.x, ps.x);
  d.y = __builtin_fmaf(ik, d.y, ps.y);
  d.z = __builtin_fmaf(ik, d.z, ps.z);
  d.w = __builtin_fmaf(ik, d.w, ps.w);
}

__device__ __forceinline__ float cp_1(float d, const float* pin, float b, float c, float ik) {
  const float ps = __builtin_fmaf(b, *pin, c * d);
  return __builtin_fmaf(ik, d, ps);
}

// x-directed layers (per-cell coefficients, psi [k][j][nslot_x]) for two difference vectors
template <bool STORE>
__device__ __forceinline__ void cp_x4(const DevParams& p, int eh, int i0, int rowslot, float4& da, const float* ain,
                                      float* aout, float4& db, const float* bin, float* bout) {
  float a[4] = {da.x, da.y, da.z, da.w};
  float bb[4] = {db.x, db.y, db.z, db.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = i0 + e;
    const int sx = pml_slot_in(p, 0, i, p.nx);
    if (sx >= 0) {
      const float b = p.cp[0][eh][0][i], c = p.cp[0][eh][1][i], ik = p.cp[0][eh][2][i];
      const int o = rowslot + sx;
      float ps = __builtin_fmaf(b, ain[o], c * a[e]);
      if (STORE) aout[o] = ps;
      a[e] = __builtin_fmaf(ik, a[e], ps);
      ps = __builtin_fmaf(b, bin[o], c * bb[e]);
      if (STORE) bout[o] = ps;
      bb[e] = __builtin_fmaf(ik, bb[e], ps);
    }
  }
  da = make_float4(a[0], a[1], a[2], a[3]);
  db = make_float4(bb[0], bb[1], bb[2], bb[3]);
}

// same, one difference vector (neighbour rows only need one of the two x-differenced components)
__device__ __forceinline__ void cp_x4_one(const DevParams& p, int i0, int rowslot, float4& da, const float* ain) {
  float a[4] = {da.x, da.y, da.z, da.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = i0 + e;
    const int sx = pml_slot_in(p, 0, i, p.nx);
    if (sx >= 0) {
      const float b = p.cp[0][0][0][i], c = p.cp[0][0][1][i], ik = p.cp[0][0][2][i];
      const float ps = __builtin_fmaf(b, ain[rowslot + sx], c * a[e]);
      a[e] = __builtin_fmaf(ik, a[e], ps);
    }
  }
  da = make_float4(a[0], a[1], a[2], a[3]);
}

template <int COEF>
__device__ __forceinline__ int4 cls4(const DevParams& p, int comp, int off) {
  if (COEF == 2) {
    const uchar4 c = *reinterpret_cast<const uchar4*>(p.ecls + off);
    return make_int4(3 * c.x + comp, 3 * c.y + comp, 3 * c.z + comp, 3 * c.w + comp);
  }
  const uchar4 c = *reinterpret_cast<const uchar4*>(p.ecls + comp * p.nloc + off);
  return make_int4(c.x, c.y, c.z, c.w);
}

template <int COEF>
__device__ __forceinline__ int cls1(const DevParams& p, int comp, int off) {
  return COEF == 2 ? 3 * (int)p.ecls[off] + comp : (int)p.ecls[comp * p.nloc + off];
}

// V' = vv*V + (m*(ex*myz))*(d1 - d2) for four cells
__device__ __forceinline__ float4 vnew4(const float2* lut, const int4 ci, const float4 ex, const float myz,
                                        const float4 v, const float4 d1, const float4 d2) {
  const float2 l0 = lut[ci.x], l1 = lut[ci.y], l2 = lut[ci.z], l3 = lut[ci.w];
  return upd4(make_float4(l0.x, l1.x, l2.x, l3.x), v,
              make_float4(l0.y * (ex.x * myz), l1.y * (ex.y * myz), l2.y * (ex.z * myz), l3.y * (ex.w * myz)), d1, d2);
}

__device__ __forceinline__ float vnew1(const float2* lut, const int ci, const float ex, const float myz, const float v,
                                       const float d1, const float d2) {
  const float2 l = lut[ci];
  return __builtin_fmaf(l.x, v, (l.y * (ex * myz)) * (d1 - d2));
}

// soft source contribution for edge (comp, flat offset) at this step, 0 if none
__device__ __forceinline__ float src_amount(const DevParams& p, const long long step, const int comp, const int off) {
  float a = 0.f;
  for (int e = 0; e < p.nsrc; ++e)
    if (p.src_off[e] == off && p.src_comp[e] == comp) {
      const long long t = step - p.src_delay[e];
      if (t >= 0 && t < p.nsig) a = p.src_amp[e] * p.sig[t];
    }
  return a;
}

__device__ __forceinline__ void src4(const DevParams& p, const long long step, const int comp, const int off, float4& v) {
  const float a0 = src_amount(p, step, comp, off), a1 = src_amount(p, step, comp, off + 1);
  const float a2 = src_amount(p, step, comp, off + 2), a3 = src_amount(p, step, comp, off + 3);
  if (a0 != 0.f) v.x = v.x + a0;
  if (a1 != 0.f) v.y = v.y + a1;
  if (a2 != 0.f) v.z = v.z + a2;
  if (a3 != 0.f) v.w = v.w + a3;
}

template <int COEF, bool PML>
__global__ __launch_bounds__(FDTD_BLOCK) void k_step_fused(const DevParams p, const long long step, const int extra) {
  __shared__ float2 s_lut[COEF == 2 ? 768 : 256];
  __shared__ double s_red[FDTD_BLOCK];
  if (extra && blockIdx.x == gridDim.x - 1) {   // probe block: both probe kinds of the step just finished
    probe_block(p, FDTD_KIND_V, step - 1, s_red);
    probe_block(p, FDTD_KIND_I, step - 1, s_red);
    return;
  }
  s_lut[threadIdx.x] = p.lut[threadIdx.x];
  if (COEF == 2) {
    s_lut[threadIdx.x + 256] = p.lut[threadIdx.x + 256];
    s_lut[threadIdx.x + 512] = p.lut[threadIdx.x + 512];
  }
  __syncthreads();
  int k, j, i0, strip;
  if (!decode(p, 0, p.nk, extra, k, j, i0, strip)) return;
  const int P = p.P, PL = p.plane;
  const int off = k * PL + j * P + i0;
  const int oj = off + P, ok = off + PL, oi = off + 4;   // row j+1, plane k+1, cell i0+4

  // ---- time level n: currents ----
  const float4 ix = ld4(p.I[0] + off), iy = ld4(p.I[1] + off), iz = ld4(p.I[2] + off);
  const float4 ix_jm = ld4(p.I[0] + off - P), iz_jm = ld4(p.I[2] + off - P);
  const float4 ix_km = ld4(p.I[0] + off - PL), iy_km = ld4(p.I[1] + off - PL);
  const float iy_im = p.I[1][off - 1], iz_im = p.I[2][off - 1];
  const float4 ix_jp = ld4(p.I[0] + oj), iy_jp = ld4(p.I[1] + oj), iz_jp = ld4(p.I[2] + oj);
  const float4 iy_jpkm = ld4(p.I[1] + oj - PL);
  const float iy_jp_im = p.I[1][oj - 1];
  const float4 ix_kp = ld4(p.I[0] + ok), iy_kp = ld4(p.I[1] + ok), iz_kp = ld4(p.I[2] + ok);
  const float4 iz_kpjm = ld4(p.I[2] + ok - P);
  const float iz_kp_im = p.I[2][ok - 1];
  const float ix_ip = p.I[0][oi], iy_ip = p.I[1][oi], iz_ip = p.I[2][oi];
  const float ix_ip_km = p.I[0][oi - PL], ix_ip_jm = p.I[0][oi - P];
  // ---- time level n: voltages ----
  float4 vx = ld4(p.V[0] + off), vy = ld4(p.V[1] + off), vz = ld4(p.V[2] + off);
  float4 vx_jp = ld4(p.V[0] + oj), vz_jp = ld4(p.V[2] + oj);
  float4 vx_kp = ld4(p.V[0] + ok), vy_kp = ld4(p.V[1] + ok);
  float vy_ip = p.V[1][oi], vz_ip = p.V[2][oi];

  // ---- differences of I (component c: d1 along axis c+1, d2 along axis c+2) ----
  // A: own cells
  float4 ax1 = sub4(iz, iz_jm), ax2 = sub4(iy, iy_km);
  float4 ay1 = sub4(ix, ix_km), ay2 = make_float4(iz.x - iz_im, iz.y - iz.x, iz.z - iz.y, iz.w - iz.z);
  float4 az1 = make_float4(iy.x - iy_im, iy.y - iy.x, iy.z - iy.y, iy.w - iy.z), az2 = sub4(ix, ix_jm);
  // B: row j+1, components x and z
  float4 bx1 = sub4(iz_jp, iz), bx2 = sub4(iy_jp, iy_jpkm);
  float4 bz1 = make_float4(iy_jp.x - iy_jp_im, iy_jp.y - iy_jp.x, iy_jp.z - iy_jp.y, iy_jp.w - iy_jp.z), bz2 = sub4(ix_jp, ix);
  // C: plane k+1, components x and y
  float4 cx1 = sub4(iz_kp, iz_kpjm), cx2 = sub4(iy_kp, iy);
  float4 cy1 = sub4(ix_kp, ix), cy2 = make_float4(iz_kp.x - iz_kp_im, iz_kp.y - iz_kp.x, iz_kp.z - iz_kp.y, iz_kp.w - iz_kp.z);
  // D: cell i0+4, components y and z
  float dy1 = ix_ip - ix_ip_km, dy2 = iz_ip - iz.w;
  float dz1 = iy_ip - iy.w, dz2 = ix_ip - ix_ip_jm;

  if (PML) {
    const int sy = pml_slot(p, 1, j), sy1 = pml_slot_in(p, 1, j + 1, p.ny);
    const int sz = pml_slot(p, 2, k), sz1 = pml_slot_in(p, 2, k + 1, p.nk);
    if (sy >= 0) {
      const float b = p.cp[1][0][0][j], c = p.cp[1][0][1][j], ik = p.cp[1][0][2][j];
      const int o = (k * p.nslot[1] + sy) * P + i0;
      cp_row4<true>(ax1, p.psiE[0][0] + o, p.psiEn[0][0] + o, b, c, ik);
      cp_row4<true>(az2, p.psiE[2][1] + o, p.psiEn[2][1] + o, b, c, ik);
      dz2 = cp_1(dz2, p.psiE[2][1] + o + 4, b, c, ik);
      if (k + 1 < p.nk) {
        const int o1 = o + p.nslot[1] * P;
        cp_row4<false>(cx1, p.psiE[0][0] + o1, nullptr, b, c, ik);
      }
    }
    if (sy1 >= 0) {
      const float b = p.cp[1][0][0][j + 1], c = p.cp[1][0][1][j + 1], ik = p.cp[1][0][2][j + 1];
      const int o = (k * p.nslot[1] + sy1) * P + i0;
      cp_row4<false>(bx1, p.psiE[0][0] + o, nullptr, b, c, ik);
      cp_row4<false>(bz2, p.psiE[2][1] + o, nullptr, b, c, ik);
    }
    if (sz >= 0) {
      const float b = p.cp[2][0][0][k], c = p.cp[2][0][1][k], ik = p.cp[2][0][2][k];
      const int o = (sz * p.ny + j) * P + i0;
      cp_row4<true>(ax2, p.psiE[0][1] + o, p.psiEn[0][1] + o, b, c, ik);
      cp_row4<true>(ay1, p.psiE[1][0] + o, p.psiEn[1][0] + o, b, c, ik);
      dy1 = cp_1(dy1, p.psiE[1][0] + o + 4, b, c, ik);
      if (j + 1 < p.ny) cp_row4<false>(bx2, p.psiE[0][1] + o + P, nullptr, b, c, ik);
    }
    if (sz1 >= 0) {
      const float b = p.cp[2][0][0][k + 1], c = p.cp[2][0][1][k + 1], ik = p.cp[2][0][2][k + 1];
      const int o = (sz1 * p.ny + j) * P + i0;
      cp_row4<false>(cx2, p.psiE[0][1] + o, nullptr, b, c, ik);
      cp_row4<false>(cy1, p.psiE[1][0] + o, nullptr, b, c, ik);
    }
    if (i0 < p.pml_lo[0] || i0 + 4 >= p.pml_hi[0]) {
      const int rs = (k * p.ny + j) * p.nslot[0];
      cp_x4<true>(p, 0, i0, rs, ay2, p.psiE[1][1], p.psiEn[1][1], az1, p.psiE[2][0], p.psiEn[2][0]);
      if (j + 1 < p.ny) cp_x4_one(p, i0, rs + p.nslot[0], bz1, p.psiE[2][0]);
      if (k + 1 < p.nk) cp_x4_one(p, i0, rs + p.ny * p.nslot[0], cy2, p.psiE[1][1]);
      const int sx4 = pml_slot_in(p, 0, i0 + 4, p.nx);
      if (sx4 >= 0) {
        const float b = p.cp[0][0][0][i0 + 4], c = p.cp[0][0][1][i0 + 4], ik = p.cp[0][0][2][i0 + 4];
        dy2 = cp_1(dy2, p.psiE[1][1] + rs + sx4, b, c, ik);
        dz1 = cp_1(dz1, p.psiE[2][0] + rs + sx4, b, c, ik);
      }
    }
  }

  // ---- E half-step: new voltages at own cells and at the three neighbour positions ----
  {
    const float4 ex0 = ld4(p.emet[0][0] + i0), ex1 = ld4(p.emet[1][0] + i0), ex2 = ld4(p.emet[2][0] + i0);
    const float ex1_ip = p.emet[1][0][i0 + 4], ex2_ip = p.emet[2][0][i0 + 4];
    const float y0 = p.emet[0][1][j], y1 = p.emet[1][1][j], y2 = p.emet[2][1][j];
    const float z0 = p.emet[0][2][k], z1 = p.emet[1][2][k], z2 = p.emet[2][2][k];
    const float y0p = p.emet[0][1][j + 1], y2p = p.emet[2][1][j + 1];
    const float z0p = p.emet[0][2][k + 1], z1p = p.emet[1][2][k + 1];
    vx = vnew4(s_lut, cls4<COEF>(p, 0, off), ex0, y0 * z0, vx, ax1, ax2);
    vy = vnew4(s_lut, cls4<COEF>(p, 1, off), ex1, y1 * z1, vy, ay1, ay2);
    vz = vnew4(s_lut, cls4<COEF>(p, 2, off), ex2, y2 * z2, vz, az1, az2);
    vx_jp = vnew4(s_lut, cls4<COEF>(p, 0, oj), ex0, y0p * z0, vx_jp, bx1, bx2);
    vz_jp = vnew4(s_lut, cls4<COEF>(p, 2, oj), ex2, y2p * z2, vz_jp, bz1, bz2);
    vx_kp = vnew4(s_lut, cls4<COEF>(p, 0, ok), ex0, y0 * z0p, vx_kp, cx1, cx2);
    vy_kp = vnew4(s_lut, cls4<COEF>(p, 1, ok), ex1, y1 * z1p, vy_kp, cy1, cy2);
    vy_ip = vnew1(s_lut, cls1<COEF>(p, 1, oi), ex1_ip, y1 * z1, vy_ip, dy1, dy2);
    vz_ip = vnew1(s_lut, cls1<COEF>(p, 2, oi), ex2_ip, y2 * z2, vz_ip, dz1, dz2);
  }
  if (p.nsrc > 0 && p.src_flag2[k * p.nstrips + strip]) {
    src4(p, step, 0, off, vx); src4(p, step, 1, off, vy); src4(p, step, 2, off, vz);
    src4(p, step, 0, oj, vx_jp); src4(p, step, 2, oj, vz_jp);
    src4(p, step, 0, ok, vx_kp); src4(p, step, 1, ok, vy_kp);
    const float a1 = src_amount(p, step, 1, oi), a2 = src_amount(p, step, 2, oi);
    if (a1 != 0.f) vy_ip = vy_ip + a1;
    if (a2 != 0.f) vz_ip = vz_ip + a2;
  }
  st4(p.Vn[0] + off, vx);
  st4(p.Vn[1] + off, vy);
  st4(p.Vn[2] + off, vz);

  // ---- H half-step on the own cells, from the new voltages ----
  float4 hx1 = sub4(vz, vz_jp), hx2 = sub4(vy, vy_kp);
  float4 hy1 = sub4(vx, vx_kp), hy2 = make_float4(vz.x - vz.y, vz.y - vz.z, vz.z - vz.w, vz.w - vz_ip);
  float4 hz1 = make_float4(vy.x - vy.y, vy.y - vy.z, vy.z - vy.w, vy.w - vy_ip), hz2 = sub4(vx, vx_jp);
  if (PML) {
    const int sy = pml_slot(p, 1, j);
    if (sy >= 0) {
      const float b = p.cp[1][1][0][j], c = p.cp[1][1][1][j], ik = p.cp[1][1][2][j];
      const int o = (k * p.nslot[1] + sy) * P + i0;
      cpml_row4(hx1, p.psiH[0][0] + o, b, c, ik);
      cpml_row4(hz2, p.psiH[2][1] + o, b, c, ik);
    }
    const int sz = pml_slot(p, 2, k);
    if (sz >= 0) {
      const float b = p.cp[2][1][0][k], c = p.cp[2][1][1][k], ik = p.cp[2][1][2][k];
      const int o = (sz * p.ny + j) * P + i0;
      cpml_row4(hx2, p.psiH[0][1] + o, b, c, ik);
      cpml_row4(hy1, p.psiH[1][0] + o, b, c, ik);
    }
    if (i0 < p.pml_lo[0] || i0 + 3 >= p.pml_hi[0])
      cpml_x4(p, 1, i0, (k * p.ny + j) * p.nslot[0], hy2, p.psiH[1][1], hz1, p.psiH[2][0]);
  }
  {
    const float4 h0 = ld4(p.hmet[0][0] + i0), h1 = ld4(p.hmet[1][0] + i0), h2 = ld4(p.hmet[2][0] + i0);
    const float m0 = p.hmet[0][1][j] * p.hmet[0][2][k];
    const float m1 = p.hmet[1][1][j] * p.hmet[1][2][k];
    const float m2 = p.hmet[2][1][j] * p.hmet[2][2][k];
    const float4 nx_ = make_float4(ix.x + (h0.x * m0) * (hx1.x - hx2.x), ix.y + (h0.y * m0) * (hx1.y - hx2.y),
                                   ix.z + (h0.z * m0) * (hx1.z - hx2.z), ix.w + (h0.w * m0) * (hx1.w - hx2.w));
    const float4 ny_ = make_float4(iy.x + (h1.x * m1) * (hy1.x - hy2.x), iy.y + (h1.y * m1) * (hy1.y - hy2.y),
                                   iy.z + (h1.z * m1) * (hy1.z - hy2.z), iy.w + (h1.w * m1) * (hy1.w - hy2.w));
    const float4 nz_ = make_float4(iz.x + (h2.x * m2) * (hz1.x - hz2.x), iz.y + (h2.y * m2) * (hz1.y - hz2.y),
                                   iz.z + (h2.z * m2) * (hz1.z - hz2.z), iz.w + (h2.w * m2) * (hz1.w - hz2.w));
    st4(p.In[0] + off, nx_);
    st4(p.In[1] + off, ny_);
    st4(p.In[2] + off, nz_);
  }
}

}  // namespace

void launch_step_fused(fdtd_ctx* c, long long step, bool probe_block, hipStream_t s) {
  const int extra = probe_block ? 1 : 0;
  const dim3 grid((unsigned)(c->p.nstrips * c->p.nk * c->p.nbs + extra)), block(FDTD_BLOCK);
  if (c->packed_op) {
    if (c->have_cpml) hipLaunchKernelGGL((k_step_fused<2, true>), grid, block, 0, s, c->p, step, extra);
    else hipLaunchKernelGGL((k_step_fused<2, false>), grid, block, 0, s, c->p, step, extra);
  } else {
    if (c->have_cpml) hipLaunchKernelGGL((k_step_fused<1, true>), grid, block, 0, s, c->p, step, extra);
    else hipLaunchKernelGGL((k_step_fused<1, false>), grid, block, 0, s, c->p, step, extra);
  }
}
