// fused.hip — one-pass leapfrog: E half-step and H half-step of a timestep in ONE sweep over memory.
//
// The two-pass kernels (kernels.hip) already run at the chip's streaming ceiling for the bytes they move
// (profiles/r01: 246 MB per E launch at ~5.7-6.2 TB/s), so the only way up is fewer bytes: a two-pass
// leapfrog reads and writes every field twice per timestep (72 B/cell), a fused sweep once (48 B/cell).
//
// For a thread's four cells at (i0..i0+3, j, k) the H update needs the NEW voltages at row j+1, plane k+1
// and cell i0+4 as well.  Row j+1 and plane k+1 are recomputed by the thread itself (same fmaf sequence =>
// bit-identical to what the owning thread stores; the extra loads are L1/L2 hits on rows its neighbours
// load anyway); cell i0+4 comes from the next lane through a wave shuffle — waves overlap by one lane
// (63 owners + 1 helper lane that only computes E), so there is no LDS tile and no barrier.
// Because neighbouring blocks still read time-level-n values of cells this block owns, results go to a
// second buffer set (ping-pong): V, I and the E-side CPML psi are double-buffered; psi_H is only touched
// by its owner and is updated in place.  Ghost planes, pads and table slack make every neighbour access
// in-bounds; values beyond the grid only ever meet zero coefficients.
//
// Instruction economy (the first version of this kernel was issue/latency bound): every global load is
// issued in one phase ahead of a compiler barrier; CPML is gated per block (z: uniform, y: strip rows) and
// lanes outside a layer run the same code with identity coefficients and a clamped psi address; the
// x-directed psi arrays use a float4-aligned internal slot layout.
//
// Single-slab, CPML/PEC scenes with a class-compressed operator; everything else (Mur, raw operator,
// multi-rank halo choreography) uses the two-pass path.  Replaces, like kernels.hip, the stepping inside
// FDTD.Run(...) of the reference's external engine (antenna_sim/solver_fdtd_openems_fixed.py:280).
#include "kernel_common.hpp"

namespace {

constexpr int OWN = 63;                 // owner lanes per wave; lane 63 recomputes the next wave's first group
constexpr int GROUPS = 4 * OWN;         // cell groups (4 cells each) per 256-thread block

struct Cp { float b, c, ik; };
struct Cp4 { float4 b, c, ik; };

// psi' = b*psi + c*d ; d <- d/kappa + psi'
__device__ __forceinline__ float4 cp4(float4& d, const float4 ps, const Cp q) {
  const float4 n = make_float4(__builtin_fmaf(q.b, ps.x, q.c * d.x), __builtin_fmaf(q.b, ps.y, q.c * d.y),
                               __builtin_fmaf(q.b, ps.z, q.c * d.z), __builtin_fmaf(q.b, ps.w, q.c * d.w));
  d = make_float4(__builtin_fmaf(q.ik, d.x, n.x), __builtin_fmaf(q.ik, d.y, n.y), __builtin_fmaf(q.ik, d.z, n.z),
                  __builtin_fmaf(q.ik, d.w, n.w));
  return n;
}
__device__ __forceinline__ float4 cp4v(float4& d, const float4 ps, const Cp4& q) {
  const float4 n = make_float4(__builtin_fmaf(q.b.x, ps.x, q.c.x * d.x), __builtin_fmaf(q.b.y, ps.y, q.c.y * d.y),
                               __builtin_fmaf(q.b.z, ps.z, q.c.z * d.z), __builtin_fmaf(q.b.w, ps.w, q.c.w * d.w));
  d = make_float4(__builtin_fmaf(q.ik.x, d.x, n.x), __builtin_fmaf(q.ik.y, d.y, n.y), __builtin_fmaf(q.ik.z, d.z, n.z),
                  __builtin_fmaf(q.ik.w, d.w, n.w));
  return n;
}

template <int COEF>
__device__ __forceinline__ int4 cls4(const uchar4 c, const int comp) {
  return COEF == 2 ? make_int4(3 * c.x + comp, 3 * c.y + comp, 3 * c.z + comp, 3 * c.w + comp) : make_int4(c.x, c.y, c.z, c.w);
}

// V' = vv*V + (m*(ex*myz))*(d1 - d2) for four cells
__device__ __forceinline__ float4 vnew4(const float2* lut, const int4 ci, const float4 ex, const float myz,
                                        const float4 v, const float4 d1, const float4 d2) {
  const float2 l0 = lut[ci.x], l1 = lut[ci.y], l2 = lut[ci.z], l3 = lut[ci.w];
  return upd4(make_float4(l0.x, l1.x, l2.x, l3.x), v,
              make_float4(l0.y * (ex.x * myz), l1.y * (ex.y * myz), l2.y * (ex.z * myz), l3.y * (ex.w * myz)), d1, d2);
}

__device__ __forceinline__ float4 shift_lo(const float4 a, const float prev) {   // a(i) - a(i-1)
  return make_float4(a.x - prev, a.y - a.x, a.z - a.y, a.w - a.z);
}
__device__ __forceinline__ float4 shift_hi(const float4 a, const float next) {   // a(i) - a(i+1)
  return make_float4(a.x - a.y, a.y - a.z, a.z - a.w, a.w - next);
}

template <int COEF, bool PML>
__global__ __launch_bounds__(FDTD_BLOCK, 2) void k_step_fused(const DevParams p, const long long step, const int extra) {
  __shared__ float2 s_lut[COEF == 2 ? 768 : 256];
  __shared__ double s_red[FDTD_BLOCK];
  __shared__ SrcStage s_src;
  if (extra && blockIdx.x == gridDim.x - 1) {   // probe block: both probe kinds of the step just finished
    probe_block(p, FDTD_KIND_V, step - 1, s_red);
    probe_block(p, FDTD_KIND_I, step - 1, s_red);
    return;
  }
  // ---- XCD-aware strip-major decode; 63 owner groups + 1 helper per wave ----
  int strip, k, pb;
  decode_block(p.nbs2, p.nk, extra, strip, k, pb);
  for (int q = threadIdx.x; q < p.lut_n; q += FDTD_BLOCK) s_lut[q] = p.lut[q];
  int2 srng = make_int2(0, 0);
  if (p.nsrc > 0) {   // sources touching this strip-plane's rows [j0, j0+rows] x planes [k, k+1]
    srng = p.src_rng2[k * p.nstrips2 + strip];
    stage_sources(p, p.src_ids2, srng.x, min(srng.y - srng.x, FDTD_BLOCK), step, s_src);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = strip * p.tys2;
  const int rows = min(p.tys2, p.ny - j0);
  const int ngrp = rows * p.P4;
  const int tt = (pb * 4 + wave) * OWN + lane;
  if (tt > ngrp) return;                                  // tt == ngrp: helper for the strip's last group
  const bool owner = tt < ngrp && lane < OWN;
  const int jj = tt / p.P4;
  const int j = j0 + jj, i0 = (tt - jj * p.P4) * 4;
  const int P = p.P, PL = p.plane;
  const int off = k * PL + j * P + i0;
  const int oj = off + P, ok = off + PL;

  // ---- CPML gating: z uniform per block, y per strip, x per lane ----
  const bool zany = PML && p.nslot[2] > 0 && (k < p.pml_lo[2] || k + 1 >= p.pml_hi[2]);
  const bool yany = PML && p.nslot[1] > 0 && (j0 < p.pml_lo[1] || j0 + rows + 1 >= p.pml_hi[1]);
  const bool xany = PML && p.nslot[0] > 0;
  const int sy = PML ? pml_slot(p, 1, j) : -1, sy1 = PML && j + 1 < p.ny ? pml_slot(p, 1, j + 1) : -1;
  const int sz = PML ? pml_slot(p, 2, k) : -1, sz1 = PML && k + 1 < p.nk ? pml_slot(p, 2, k + 1) : -1;
  const int sx = PML ? pml_slot(p, 0, i0) : -1;            // aligned layout: all four cells share the range
  // psi offsets (clamped to slot 0 outside the layers: any valid address, the coefficients are identity there)
  const int oy = (k * p.nslot[1] + max(sy, 0)) * P + i0;            // y layout [k][sy][P]
  const int oy1 = (k * p.nslot[1] + max(sy1, 0)) * P + i0;
  const int oyk = oy + p.nslot[1] * P;                               // same row slot, plane k+1
  const int oz = (max(sz, 0) * p.ny + j) * P + i0;                  // z layout [sz][ny][P]
  const int oz1 = (max(sz1, 0) * p.ny + j) * P + i0;
  const int ox = (k * p.ny + j) * p.nslot[0] + max(sx, 0);          // x layout [k][j][nslot_x]

  // ================= load phase =================
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
  float4 vx = ld4(p.V[0] + off), vy = ld4(p.V[1] + off), vz = ld4(p.V[2] + off);
  float4 vx_jp = ld4(p.V[0] + oj), vz_jp = ld4(p.V[2] + oj);
  float4 vx_kp = ld4(p.V[0] + ok), vy_kp = ld4(p.V[1] + ok);
  uchar4 c_own[3], c_jp[3], c_kp[3];
  if (COEF == 2) {
    c_own[0] = c_own[1] = c_own[2] = *reinterpret_cast<const uchar4*>(p.ecls + off);
    c_jp[0] = c_jp[2] = *reinterpret_cast<const uchar4*>(p.ecls + oj);
    c_kp[0] = c_kp[1] = *reinterpret_cast<const uchar4*>(p.ecls + ok);
  } else {
    c_own[0] = *reinterpret_cast<const uchar4*>(p.ecls + off);
    c_own[1] = *reinterpret_cast<const uchar4*>(p.ecls + p.nloc + off);
    c_own[2] = *reinterpret_cast<const uchar4*>(p.ecls + 2 * p.nloc + off);
    c_jp[0] = *reinterpret_cast<const uchar4*>(p.ecls + oj);
    c_jp[2] = *reinterpret_cast<const uchar4*>(p.ecls + 2 * p.nloc + oj);
    c_kp[0] = *reinterpret_cast<const uchar4*>(p.ecls + ok);
    c_kp[1] = *reinterpret_cast<const uchar4*>(p.ecls + p.nloc + ok);
  }
  const float4 ex0 = ld4(p.emet[0][0] + i0), ex1 = ld4(p.emet[1][0] + i0), ex2 = ld4(p.emet[2][0] + i0);
  const float y0 = p.emet[0][1][j], y1 = p.emet[1][1][j], y2 = p.emet[2][1][j];
  const float y0p = p.emet[0][1][j + 1], y2p = p.emet[2][1][j + 1];
  const float z0 = p.emet[0][2][k], z1 = p.emet[1][2][k], z2 = p.emet[2][2][k];
  const float z0p = p.emet[0][2][k + 1], z1p = p.emet[1][2][k + 1];
  // E-side psi (current buffers) + coefficients
  float4 qy_ax1, qy_az2, qy_cx1, qy_bx1, qy_bz2, qz_ax2, qz_ay1, qz_bx2, qz_cx2, qz_cy1, qx_ay2, qx_az1, qx_bz1, qx_cy2;
  Cp cy_j, cy_j1, cz_k, cz_k1;
  Cp4 cx_e;
  if (yany) {
    qy_ax1 = ld4(p.psiE[0][0] + oy); qy_az2 = ld4(p.psiE[2][1] + oy); qy_cx1 = ld4(p.psiE[0][0] + (k + 1 < p.nk ? oyk : oy));
    qy_bx1 = ld4(p.psiE[0][0] + oy1); qy_bz2 = ld4(p.psiE[2][1] + oy1);
    cy_j = Cp{p.cp[1][0][0][j], p.cp[1][0][1][j], p.cp[1][0][2][j]};
    cy_j1 = Cp{p.cp[1][0][0][j + 1], p.cp[1][0][1][j + 1], p.cp[1][0][2][j + 1]};
  }
  if (zany) {
    qz_ax2 = ld4(p.psiE[0][1] + oz); qz_ay1 = ld4(p.psiE[1][0] + oz); qz_bx2 = ld4(p.psiE[0][1] + (j + 1 < p.ny ? oz + P : oz));
    qz_cx2 = ld4(p.psiE[0][1] + oz1); qz_cy1 = ld4(p.psiE[1][0] + oz1);
    cz_k = Cp{p.cp[2][0][0][k], p.cp[2][0][1][k], p.cp[2][0][2][k]};
    cz_k1 = Cp{p.cp[2][0][0][k + 1], p.cp[2][0][1][k + 1], p.cp[2][0][2][k + 1]};
  }
  if (xany) {
    qx_ay2 = ld4(p.psiE[1][1] + ox); qx_az1 = ld4(p.psiE[2][0] + ox);
    qx_bz1 = ld4(p.psiE[2][0] + (j + 1 < p.ny ? ox + p.nslot[0] : ox));
    qx_cy2 = ld4(p.psiE[1][1] + (k + 1 < p.nk ? ox + p.ny * p.nslot[0] : ox));
    cx_e = Cp4{ld4(p.cp[0][0][0] + i0), ld4(p.cp[0][0][1] + i0), ld4(p.cp[0][0][2] + i0)};
  }
  asm volatile("" ::: "memory");   // keep every load above the arithmetic: one memory round trip per thread

  // ================= E half-step =================
  // component c: d1 along axis c+1, d2 along axis c+2.  A: own cells, B: row j+1 (x,z), C: plane k+1 (x,y)
  float4 ax1 = sub4(iz, iz_jm), ax2 = sub4(iy, iy_km);
  float4 ay1 = sub4(ix, ix_km), ay2 = shift_lo(iz, iz_im);
  float4 az1 = shift_lo(iy, iy_im), az2 = sub4(ix, ix_jm);
  float4 bx1 = sub4(iz_jp, iz), bx2 = sub4(iy_jp, iy_jpkm);
  float4 bz1 = shift_lo(iy_jp, iy_jp_im), bz2 = sub4(ix_jp, ix);
  float4 cx1 = sub4(iz_kp, iz_kpjm), cx2 = sub4(iy_kp, iy);
  float4 cy1 = sub4(ix_kp, ix), cy2 = shift_lo(iz_kp, iz_kp_im);
  if (yany) {
    const float4 n1 = cp4(ax1, qy_ax1, cy_j), n2 = cp4(az2, qy_az2, cy_j);
    (void)cp4(cx1, qy_cx1, cy_j);
    (void)cp4(bx1, qy_bx1, cy_j1);
    (void)cp4(bz2, qy_bz2, cy_j1);
    if (owner && sy >= 0) { st4(p.psiEn[0][0] + oy, n1); st4(p.psiEn[2][1] + oy, n2); }
  }
  if (zany) {
    const float4 n1 = cp4(ax2, qz_ax2, cz_k), n2 = cp4(ay1, qz_ay1, cz_k);
    (void)cp4(bx2, qz_bx2, cz_k);
    (void)cp4(cx2, qz_cx2, cz_k1);
    (void)cp4(cy1, qz_cy1, cz_k1);
    if (owner && sz >= 0) { st4(p.psiEn[0][1] + oz, n1); st4(p.psiEn[1][0] + oz, n2); }
  }
  if (xany) {
    const float4 n1 = cp4v(ay2, qx_ay2, cx_e), n2 = cp4v(az1, qx_az1, cx_e);
    (void)cp4v(bz1, qx_bz1, cx_e);
    (void)cp4v(cy2, qx_cy2, cx_e);
    if (owner && sx >= 0) { st4(p.psiEn[1][1] + ox, n1); st4(p.psiEn[2][0] + ox, n2); }
  }
  vx = vnew4(s_lut, cls4<COEF>(c_own[0], 0), ex0, y0 * z0, vx, ax1, ax2);
  vy = vnew4(s_lut, cls4<COEF>(c_own[1], 1), ex1, y1 * z1, vy, ay1, ay2);
  vz = vnew4(s_lut, cls4<COEF>(c_own[2], 2), ex2, y2 * z2, vz, az1, az2);
  vx_jp = vnew4(s_lut, cls4<COEF>(c_jp[0], 0), ex0, y0p * z0, vx_jp, bx1, bx2);
  vz_jp = vnew4(s_lut, cls4<COEF>(c_jp[2], 2), ex2, y2p * z2, vz_jp, bz1, bz2);
  vx_kp = vnew4(s_lut, cls4<COEF>(c_kp[0], 0), ex0, y0 * z0p, vx_kp, cx1, cx2);
  vy_kp = vnew4(s_lut, cls4<COEF>(c_kp[1], 1), ex1, y1 * z1p, vy_kp, cy1, cy2);
  if (srng.y > srng.x) {
    const int n = min(srng.y - srng.x, FDTD_BLOCK);   // (more than FDTD_BLOCK sources per strip-plane: see ensure_fused)
    apply_staged(s_src, n, 0, off, vx); apply_staged(s_src, n, 1, off, vy); apply_staged(s_src, n, 2, off, vz);
    apply_staged(s_src, n, 0, oj, vx_jp); apply_staged(s_src, n, 2, oj, vz_jp);
    apply_staged(s_src, n, 0, ok, vx_kp); apply_staged(s_src, n, 1, ok, vy_kp);
  }
  // new voltages of cell i0+4 = first cell of the next lane's group (lane 63 of every wave is that helper)
  const float vy_ip = __shfl_down(vy.x, 1), vz_ip = __shfl_down(vz.x, 1);
  if (!owner) return;
  st4(p.Vn[0] + off, vx);
  st4(p.Vn[1] + off, vy);
  st4(p.Vn[2] + off, vz);

  // ================= H half-step on the own cells, from the new voltages =================
  // second load phase (kept out of the first one to stay below 256 VGPRs without spilling)
  const float4 h0 = ld4(p.hmet[0][0] + i0), h1 = ld4(p.hmet[1][0] + i0), h2 = ld4(p.hmet[2][0] + i0);
  const float hm0 = p.hmet[0][1][j] * p.hmet[0][2][k];
  const float hm1 = p.hmet[1][1][j] * p.hmet[1][2][k];
  const float hm2 = p.hmet[2][1][j] * p.hmet[2][2][k];
  // H-side psi (in place) + coefficients
  float4 ry_hx1, ry_hz2, rz_hx2, rz_hy1, rx_hy2, rx_hz1;
  Cp dy_j, dz_k;
  Cp4 dx_h;
  if (yany) {
    ry_hx1 = ld4(p.psiH[0][0] + oy); ry_hz2 = ld4(p.psiH[2][1] + oy);
    dy_j = Cp{p.cp[1][1][0][j], p.cp[1][1][1][j], p.cp[1][1][2][j]};
  }
  if (zany) {
    rz_hx2 = ld4(p.psiH[0][1] + oz); rz_hy1 = ld4(p.psiH[1][0] + oz);
    dz_k = Cp{p.cp[2][1][0][k], p.cp[2][1][1][k], p.cp[2][1][2][k]};
  }
  if (xany) {
    rx_hy2 = ld4(p.psiH[1][1] + ox); rx_hz1 = ld4(p.psiH[2][0] + ox);
    dx_h = Cp4{ld4(p.cp[0][1][0] + i0), ld4(p.cp[0][1][1] + i0), ld4(p.cp[0][1][2] + i0)};
  }
  asm volatile("" ::: "memory");
  float4 hx1 = sub4(vz, vz_jp), hx2 = sub4(vy, vy_kp);
  float4 hy1 = sub4(vx, vx_kp), hy2 = shift_hi(vz, vz_ip);
  float4 hz1 = shift_hi(vy, vy_ip), hz2 = sub4(vx, vx_jp);
  if (yany) {
    const float4 n1 = cp4(hx1, ry_hx1, dy_j), n2 = cp4(hz2, ry_hz2, dy_j);
    if (sy >= 0) { st4(p.psiH[0][0] + oy, n1); st4(p.psiH[2][1] + oy, n2); }
  }
  if (zany) {
    const float4 n1 = cp4(hx2, rz_hx2, dz_k), n2 = cp4(hy1, rz_hy1, dz_k);
    if (sz >= 0) { st4(p.psiH[0][1] + oz, n1); st4(p.psiH[1][0] + oz, n2); }
  }
  if (xany) {
    const float4 n1 = cp4v(hy2, rx_hy2, dx_h), n2 = cp4v(hz1, rx_hz1, dx_h);
    if (sx >= 0) { st4(p.psiH[1][1] + ox, n1); st4(p.psiH[2][0] + ox, n2); }
  }
  st4(p.In[0] + off, make_float4(ix.x + (h0.x * hm0) * (hx1.x - hx2.x), ix.y + (h0.y * hm0) * (hx1.y - hx2.y),
                                  ix.z + (h0.z * hm0) * (hx1.z - hx2.z), ix.w + (h0.w * hm0) * (hx1.w - hx2.w)));
  st4(p.In[1] + off, make_float4(iy.x + (h1.x * hm1) * (hy1.x - hy2.x), iy.y + (h1.y * hm1) * (hy1.y - hy2.y),
                                  iy.z + (h1.z * hm1) * (hy1.z - hy2.z), iy.w + (h1.w * hm1) * (hy1.w - hy2.w)));
  st4(p.In[2] + off, make_float4(iz.x + (h2.x * hm2) * (hz1.x - hz2.x), iz.y + (h2.y * hm2) * (hz1.y - hz2.y),
                                  iz.z + (h2.z * hm2) * (hz1.z - hz2.z), iz.w + (h2.w * hm2) * (hz1.w - hz2.w)));
}

}  // namespace

// rows per strip for blocks of 4 x 63 owner groups
void choose_tiling_fused(fdtd_ctx* c) {
  const int P4 = c->p.P4, ny = c->p.ny;
  int best = 1; double best_cost = 1e30;
  for (int tys = 4; tys <= 40; ++tys) {
    const int rows = tys < ny ? tys : ny;
    const int t = rows * P4 + 1;                      // +1: the helper of the strip's last group
    const int nbs = (t + GROUPS - 1) / GROUPS;
    const double idle = (double)(nbs * GROUPS - t) / (nbs * GROUPS);
    const double cost = idle + 0.15 / rows;
    if (cost < best_cost - 1e-12) { best_cost = cost; best = rows; }
  }
  c->p.tys2 = best;
  c->p.nbs2 = (best * P4 + 1 + GROUPS - 1) / GROUPS;
  c->p.nstrips2 = (ny + best - 1) / best;
}

void launch_step_fused(fdtd_ctx* c, long long step, bool probe_block, hipStream_t s) {
  const int extra = probe_block ? 1 : 0;
  const dim3 grid((unsigned)(c->p.nstrips2 * c->p.nk * c->p.nbs2 + extra)), block(FDTD_BLOCK);
  if (c->packed_op) {
    if (c->have_cpml) hipLaunchKernelGGL((k_step_fused<2, true>), grid, block, 0, s, c->p, step, extra);
    else hipLaunchKernelGGL((k_step_fused<2, false>), grid, block, 0, s, c->p, step, extra);
  } else {
    if (c->have_cpml) hipLaunchKernelGGL((k_step_fused<1, true>), grid, block, 0, s, c->p, step, extra);
    else hipLaunchKernelGGL((k_step_fused<1, false>), grid, block, 0, s, c->p, step, extra);
  }
}
