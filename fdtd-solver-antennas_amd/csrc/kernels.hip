// kernels.hip — hand-written gfx950 kernels of the EC-FDTD hot path.
//
//   K1 k_update_E   V <- vv*V + vi*curl(I)   + fused CPML psi update      
//   K2 k_update_H   I <- ii*I + iv*curl(V)   + fused CPML psi update      (N2, N7)
//   K3 sources/probes: fused into K1/K2 (source add inside update_E, probes in one extra block);
//      k_post is the stand-alone form used with Mur boundaries and by fdtd_half_step     (N4, N8)
//   K4 k_mur        first-order Mur pre / post / apply on the six faces   (N6)
//   K6 k_dft        running DFT of field boxes (NF2FF surfaces)           (N9)
//   K8 k_energy     sum V^2, sum I^2 (per-block partials, added in block order)   (N11)
//   K1+K2 k_step    both half-steps of a timestep in ONE launch behind per-block flags; on cache-resident
//                   single slabs SEVERAL timesteps per launch (device-scope loads / write-through stores)
//
// These replace what the reference runs inside FDTD.Run(...) of the external openEMS engine
// (antenna_sim/solver_fdtd_openems_fixed.py:280 and the four sibling call sites).
//
// Roofline: 6-point-neighbour stencil, ~0.5 flop/B -> HBM bound, no MFMA.  Algorithmic traffic
// is 36 B per cell per half-step (read 3+3 fields, write 3).  Design rules applied:
//   * one thread = 4 consecutive x-cells = one dwordx4 per field component; consecutive lanes
//     read consecutive memory (rows are stored gap-free), so every wave access is 1 KiB coalesced;
//   * operator coefficients are NOT streamed: one class byte per CELL (1 B/cell; 3 B/cell when a
//     scene has more than 256 distinct edge-class triples) indexes a (vv, m) table staged in LDS, the
//     mesh metric comes from 1-D tables in L1;
//   * blockIdx is remapped so that each XCD (own 4 MiB L2) sweeps one contiguous part of the slab,
//     ordered strip-by-strip through z so the k+-1 and j+-1 neighbour rows are L2 hits; the eight parts
//     are equal in COST (CPML rows / planes weigh more), then corrected by measured finish times;
//   * CPML psi arrays exist only inside the layers; interior threads pay three compares.
//
// Float32 operation order is pinned with explicit fmaf (compiled with -ffp-contract=off) and is
// identical to oracle/fdtd_oracle.c, so results are compared bit for bit.
#include <hip/hip_ext.h>
#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

#include "kernel_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// K1: E half-step
// ------------------------------------------------------------------------------------------------
// Mur "post" pass (mode 1 of k_mur: cd <- st + coeff * V_inner with the freshly updated V) for the face points whose inner
// point this thread has just computed — the thread owns cells i0..i0+3 of row j in plane k and holds the new Vx, Vy, Vz.
// A face point is touched by exactly one thread, so the (single) update per point is that of k_mur, bit for bit.
// Two phases.  The st values of the x faces are LOADED with the field loads at the top of the kernel (PHASE 0) and used at its end
// (PHASE 1): every wave holds a cell next to an x face, and loaded at the end they were a second, exposed memory round trip for
// all of them.  Those of the y and z faces — whole rows and planes, i.e. few waves — are loaded where they are used: kept from
// the top they were 16 more registers in EVERY wave (82 ... 90 VGPRs: five waves per SIMD where the plain kernel has seven).
// st and cd rows have stride P and zero pad cells, like field rows: whole float4 accesses.  Per axis a thread lies next to at
// most one face (the host takes the separate launch for grids under 5 nodes along an axis).
struct MurVals { float x0, x1; };
__device__ __forceinline__ float4 mur_post4(const float c, const float4& v, const float4& s) {   // k_mur mode 1, four cells
  return make_float4(__builtin_fmaf(c, v.x, s.x), __builtin_fmaf(c, v.y, s.y), __builtin_fmaf(c, v.z, s.z), __builtin_fmaf(c, v.w, s.w));
}
// DEV (one-launch schedule): st was written by H blocks of the previous timestep of the same launch and cd is read by H blocks of this
// timestep on other CUs — device-scope loads, write-through stores; `par` selects the cd copy of this timestep (0 elsewhere).
template <int PHASE, bool DEV = false>
__device__ __forceinline__ void mur_post_inline(const DevParams& p, const MurDev& m, const int k, const int j, const int i0, MurVals& mv,
                                                const float4& vx, const float4& vy, const float4& vz, const int par = 0) {
  auto ld = [](const float* q, const int o) { return DEV ? ldb4_dev(dev_buf(q), (unsigned)o << 2, 0u) : ld4(q + o); };
  auto st = [](float* q, const int o, const float4& v) { if (DEV) sto4_dev(q, (unsigned)o, v); else st4(q + o, v); };
#pragma unroll
  for (int fi = 0; fi < 6; ++fi) {
    const MurDevFace& f = m.f[fi];
    if (!f.on) continue;
    const int a = fi >> 1, pc = par * f.cdn;
    if (a == 2) {          // z faces: tangential x, y; rows j of a plane (block-uniform test)
      if (PHASE == 0 || k != f.in) continue;
      const int o = j * p.P + i0;
      st(f.cd[0] + pc, o, mur_post4(f.coeff, vx, ld(f.st[0], o)));
      st(f.cd[1] + pc, o, mur_post4(f.coeff, vy, ld(f.st[1], o)));
    } else if (a == 1) {   // y faces: comp[0] = z, comp[1] = x; one row per plane
      if (PHASE == 0 || j != f.in) continue;
      const int o = k * p.P + i0;
      st(f.cd[0] + pc, o, mur_post4(f.coeff, vz, ld(f.st[0], o)));
      st(f.cd[1] + pc, o, mur_post4(f.coeff, vx, ld(f.st[1], o)));
    } else {               // x faces: comp[0] = y, comp[1] = z; [k][j]; one of the thread's four cells at most
      const int e = f.in - i0;
      if (e < 0 || e > 3) continue;
      const int o = k * p.ny + j;
      if (PHASE == 0) {
        if (DEV) { mv.x0 = ldb1_dev(dev_buf(f.st[0]), (unsigned)o << 2, 0u); mv.x1 = ldb1_dev(dev_buf(f.st[1]), (unsigned)o << 2, 0u); }
        else { mv.x0 = f.st[0][o]; mv.x1 = f.st[1][o]; }
      } else {   // (bit masks: selecting the component with a computed index sends the vectors through scratch)
        const unsigned m0 = 0u - (unsigned)(e == 0), m1 = 0u - (unsigned)(e == 1), m2 = 0u - (unsigned)(e == 2), m3 = 0u - (unsigned)(e == 3);
        const float ny_ = __uint_as_float((__float_as_uint(vy.x) & m0) | (__float_as_uint(vy.y) & m1) | (__float_as_uint(vy.z) & m2) | (__float_as_uint(vy.w) & m3));
        const float nz_ = __uint_as_float((__float_as_uint(vz.x) & m0) | (__float_as_uint(vz.y) & m1) | (__float_as_uint(vz.z) & m2) | (__float_as_uint(vz.w) & m3));
        const float c0 = __builtin_fmaf(f.coeff, ny_, mv.x0), c1 = __builtin_fmaf(f.coeff, nz_, mv.x1);
        if (DEV) { sto1_dev(f.cd[0] + pc, (unsigned)o, c0); sto1_dev(f.cd[1] + pc, (unsigned)o, c1); }
        else { f.cd[0][pc + o] = c0; f.cd[1][pc + o] = c1; }
      }
    }
  }
}

// COEF: 0 raw arrays, 1 class byte per edge, 2 one packed class byte per cell
// WF: the block is part of a one-launch-per-timestep wavefront (k_step below): V goes out write-through (sc1) and the block
// publishes its flag when every wave's stores have been acknowledged.
// MUR: the block also runs the Mur "post" pass (S += coeff * V_new on the plane next to each Mur face) for the points it owns.
// MULTI: the block is part of a launch of SEVERAL timesteps (k_step): what it reads was written — and what it overwrites was
// read — by workgroups of the previous timestep of the same launch, so it first waits for their flags (back_target != 0) and
// every load of mutable data is a device-scope load, every store a write-through store (fields and psi alike).
template <int COEF, bool PML, bool FUSE, bool P2P, bool WF, bool MUR = false, bool MULTI = false>
__device__ __forceinline__ void body_E(const DevParams& p, const int strip, const int k, const int pb, const long long step,
                                       float2* const s_lut, float4* const s_psi, float* const s_xc, SrcStage& s_src, const unsigned wf_target,
                                       const MurDev* const mur = nullptr, const unsigned back_target = 0u) {
  // coefficient table -> LDS by LDS-DMA, issued FIRST: no staging registers (the kernel has none to spare), and since
  // vector-memory operations retire in order a counted wait below leaves the field loads behind it in flight.
  // Thread t moves entries 2t, 2t+1 (16 bytes; the destination of an LDS-DMA load is lane-linear, so the table lands
  // in order); 768 entries take a second round of the first 128 threads.  The buffer is allocated for the full table.
  if (COEF != 0) {
    const unsigned w_lds = __builtin_amdgcn_readfirstlane(lds_off(s_lut) + (threadIdx.x >> 6) * 1024u);
    const float* lsrc = reinterpret_cast<const float*>(p.lut) + 4 * (int)threadIdx.x;
    if (2 * (int)threadIdx.x < p.lut_n) glds16(lsrc, w_lds);
    if (COEF == 2 && 2 * ((int)threadIdx.x + FDTD_BLOCK) < p.lut_n) glds16(lsrc + 4 * FDTD_BLOCK, w_lds + 4096u);
  }
  // (b, c, 1/kappa) of the x-layer cells in psi-slot order: a compact table the host laid out (fdtd_set_cpml), 3 x XC_MAX
  // floats, to LDS by LDS-DMA as well (96 threads x 16 bytes)
  const bool xc_lds = PML && FDTD_PSI_STAGE && p.xc_tab != nullptr;
  if (xc_lds && (int)threadIdx.x < 3 * XC_MAX / 4)
    glds16(p.xc_tab + 4 * (int)threadIdx.x, __builtin_amdgcn_readfirstlane(lds_off(s_xc) + (threadIdx.x >> 6) * 1024u));
  // the field loads are issued BEFORE the barrier that publishes the LDS tables, so that their latency overlaps
  // the table staging instead of following it (out-of-range threads of the last block read their block's first
  // group, in range by construction, and leave after the barrier)
  int j = 0, i0 = 0;
  const bool valid = decode_thread(p, strip, pb, j, i0);
  const int off = k * p.plane + (valid ? j * p.P + i0 : 0);
  constexpr bool LS = FDTD_LANE_SHIFT && LANE_SHIFT_FITS(WF, MULTI);
  // ONE per-lane offset for all thirteen loads: the neighbour displacements (k-1, j-1, i-1) go into the SCALAR base
  // pointers (SGPRs are plentiful, VGPRs decide the occupancy); bases start one plane below plane 0 (the ghost plane), so
  // that the offset stays unsigned whatever the displacement
  const unsigned uo = (unsigned)(off + p.plane);
  // P2P: the k-1 neighbours of the bottom plane are the lower rank's top plane and come from the mailbox (below)
  const bool dep_in = P2P && k == 0 && p.mb_in_H != nullptr;
  float4 ix, iy, iz, iz_jm, ix_jm, vx, vy, vz;
  float4 iy_km = make_float4(0.f, 0.f, 0.f, 0.f), ix_km = iy_km;
  float iz_im, iy_im;
  if (!MULTI) {
    const float *I0 = p.I[0] - p.plane, *I1 = p.I[1] - p.plane, *I2 = p.I[2] - p.plane;
    ix = ldo4(I0, uo); iy = ldo4(I1, uo); iz = ldo4(I2, uo);
    iz_jm = ldo4(I2 - p.P, uo); ix_jm = ldo4(I0 - p.P, uo);
    if (!dep_in) { iy_km = ldo4(I1 - p.plane, uo); ix_km = ldo4(I0 - p.plane, uo); }
    {   // (FDTD_LANE_SHIFT: only lane 0 needs these from memory — the others take lane 0's address, one segment for the whole wave, and lane_prev below)
      const unsigned ue = (!LS || (threadIdx.x & 63u) == 0u) ? uo : (unsigned)__builtin_amdgcn_readfirstlane((int)uo);
      iz_im = ldo1(I2 - 1, ue); iy_im = ldo1(I1 - 1, ue);
    }
    vx = ldo4(p.V[0] - p.plane, uo); vy = ldo4(p.V[1] - p.plane, uo); vz = ldo4(p.V[2] - p.plane, uo);
  } else {
    // the H blocks of the previous timestep that wrote what this block reads have published (and the probe blocks that sample
    // V cells of this strip-plane have read them); then device-scope loads: one buffer resource per array, based one plane
    // below plane 0, the neighbour displacements in the scalar offset
    if (back_target) {
      wf_wait_back(p, k, strip, pb, back_target);
      if (p.wf_prbV_sp != nullptr && sload_int(p.wf_prbV_sp + (k * p.nstrips + strip)) != 0) wf_wait_probes(p, FDTD_KIND_V, back_target);
    }
    const DevRsrc r0 = dev_buf(p.I[0] - 2 * p.plane), r1 = dev_buf(p.I[1] - 2 * p.plane), r2 = dev_buf(p.I[2] - 2 * p.plane);
    const unsigned bo = (unsigned)off << 2, d0 = (unsigned)(2 * p.plane) << 2, dj = (unsigned)(2 * p.plane - p.P) << 2,
                   dk = (unsigned)p.plane << 2, di = (unsigned)(2 * p.plane - 1) << 2;
    ix = ldb4_dev(r0, bo, d0); iy = ldb4_dev(r1, bo, d0); iz = ldb4_dev(r2, bo, d0);
    iz_jm = ldb4_dev(r2, bo, dj); ix_jm = ldb4_dev(r0, bo, dj);
    iy_km = ldb4_dev(r1, bo, dk); ix_km = ldb4_dev(r0, bo, dk);
    {
      const unsigned be = (!LS || (threadIdx.x & 63u) == 0u) ? bo : (unsigned)__builtin_amdgcn_readfirstlane((int)bo);
      iz_im = ldb1_dev(r2, be, di); iy_im = ldb1_dev(r1, be, di);
    }
    vx = ldb4_dev(dev_buf(p.V[0]), bo, 0u); vy = ldb4_dev(dev_buf(p.V[1]), bo, 0u); vz = ldb4_dev(dev_buf(p.V[2]), bo, 0u);
  }
  MurVals mv;
  if (MUR && valid) mur_post_inline<0, WF>(p, *mur, k, j, i0, mv, vx, vy, vz);   // the st values of the x faces travel with the field loads
  // Soft sources inside this strip-plane (block-uniform range; almost always empty).  The range comes by an explicit SCALAR
  // load: left to the compiler this uniform load sits behind the LDS-DMA statements (asm, "memory"), cannot be proven
  // unclobbered and becomes a vector load + s_waitcnt vmcnt(0) in the middle of the load phase — every wave then waited for
  // all its field loads before it issued the staged psi loads, the second round trip the staging exists to avoid.
  // A strip-plane with MANY source edges (the reference's multi-patch scene draws lumped ports of 1 350 edges each,
  // solver_fdtd_openems_microstrip_multi_3d.py:472-541: 5 400 in its 2 x 2 array) takes the DENSE form below: every thread scanning the list cost
  // that scene 175 us per timestep instead of 21.
  int2 srng = make_int2(0, 0);
  bool dense = false;
  if (FUSE && p.nsrc > 0) {
    srng = sload_int2(p.src_rng + (k * p.nstrips + strip));
    dense = srng.y - srng.x > SRC_SCAN_MAX && p.src_dense_ok;   // (two sources on ONE edge keep the scanning path: it adds them in order)
    if (srng.y > srng.x && !dense) stage_sources(p, p.src_ids, srng.x, srng.y - srng.x, step, s_src);
  }
  if (PML && FDTD_PSI_STAGE)   // psi of the x / z layers: LDS-DMA right behind the field loads (kernel_common.hpp)
    psi_stage_issue<MULTI>(p, p.psiE, __builtin_amdgcn_readfirstlane(lds_off(s_psi) + (threadIdx.x >> 6) * (PSI_SLOTS * 1024u)), valid, k, j, i0);
  if (dep_in) {   // H halo of step-1 (tag = step + 1; for step 0 the neighbour's INITIAL top plane, pushed by k_p2p_prime); slot of the parity of the step that produced it
    const float* mb = p.mb_in_H + (size_t)((step + 1) & 1) * 2 * mb_slot_words(p);
    mb_pull2(mb, mb + mb_slot_words(p), (unsigned)(j * p.P + i0), (unsigned)step + 1u + p.p2p_tag_bias, ix_km, iy_km, p.p2p_err, p.p2p_limit,
             (WF ? 0x20000000u : 0u) | ((unsigned)strip << 14) | (unsigned)pb);   // who: E half-step (bit 28 clear), one-launch schedule (bit 29), strip, block
  }

  // component x: d1 along y (of Iz), d2 along z (of Iy); y: d1 along z (Ix), d2 along x (Iz);
  // z: d1 along x (Iy), d2 along y (Ix)
  if (LS) {   // every lane active here (out-of-range threads leave after the barrier below)
    const float zp = lane_prev(iz.w), yp = lane_prev(iy.w);
    if ((threadIdx.x & 63u) != 0u) { iz_im = zp; iy_im = yp; }
  }
  float4 dx1 = sub4(iz, iz_jm), dx2 = sub4(iy, iy_km);
  float4 dy1 = sub4(ix, ix_km);
  float4 dy2 = make_float4(iz.x - iz_im, iz.y - iz.x, iz.z - iz.y, iz.w - iz.z);
  float4 dz1 = make_float4(iy.x - iy_im, iy.y - iy.x, iy.z - iy.y, iy.w - iy.z);
  float4 dz2 = sub4(ix, ix_jm);

  // ONE barrier, here: the differences above consumed every field load, so all older vector-memory operations — the
  // LDS-DMA loads of the coefficient tables and of the staged psi — have landed as well (in-order retirement; the explicit
  // wait is for the ones the compiler does not count), and no wave waits at the barrier for data it would not have
  // waited for anyway.  Out-of-range threads computed on their block's first group and leave now.
  if (COEF != 0 || (FUSE && p.nsrc > 0) || (PML && FDTD_PSI_STAGE)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (!WF && !dense && !valid) return;   // (a wavefront block meets once more, at its flag; a block with dense sources at their barriers)
  if (valid) {
  if (PML) {
    if (FDTD_PSI_STAGE) {
      psi_stage_apply<MULTI>(p, p.psiE, 0, s_psi, s_xc, xc_lds, k, j, i0, dx2, dy1, dy2, dz1, dx1, dz2);
    } else {
      const int sy = pml_slot(p, 1, j);
      if (sy >= 0) {
        const float b = p.cp[1][0][0][j], c = p.cp[1][0][1][j], ik = p.cp[1][0][2][j];
        const int o = (k * p.nslot[1] + sy) * p.P + i0;
        cpml_row4(dx1, p.psiE[0][0], (unsigned)o, b, c, ik);
        cpml_row4(dz2, p.psiE[2][1], (unsigned)o, b, c, ik);
      }
      const int sz = pml_slot(p, 2, k);
      if (sz >= 0) {
        const float b = p.cp[2][0][0][k], c = p.cp[2][0][1][k], ik = p.cp[2][0][2][k];
        const int o = (sz * p.ny + j) * p.P + i0;
        cpml_row4(dx2, p.psiE[0][1], (unsigned)o, b, c, ik);
        cpml_row4(dy1, p.psiE[1][0], (unsigned)o, b, c, ik);
      }
      if (i0 < p.pml_lo[0] || i0 >= p.pml_hi[0])   // both bounds are multiples of 4: all four cells or none
        cpml_x4(p, 0, i0, k * p.xplane + j * p.xrs, dy2, p.psiE[1][1], dz1, p.psiE[2][0]);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the coefficient loads of the update below out of the CPML section (registers)
  }
  }   // valid
  // Dense sources: the block's cells as an LDS image [component][thread][4 cells] of this timestep's source values (the psi staging area
  // is free by now).  Every thread clears its own twelve slots, the strip-plane's sources are dealt round over the threads and each
  // lands in the slot of the cell it excites (those of other blocks of the strip-plane fall outside), every thread adds its slots.
  // (Edges are distinct: two sources on one edge would overwrite each other here — DevParams::src_dense_ok, set by fdtd_add_source.)
  float* const s_dense = reinterpret_cast<float*>(s_psi);
  if (FUSE && dense) {
    if (PML && FDTD_PSI_STAGE) __syncthreads();   // everybody has read its staged psi
    float4* const d4 = reinterpret_cast<float4*>(s_dense);
    d4[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); d4[FDTD_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); d4[2 * FDTD_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int t_first = pb * FDTD_BLOCK;
    for (int q = srng.x + (int)threadIdx.x; q < srng.y; q += FDTD_BLOCK) {
      const int e = p.src_ids[q];
      const int r2 = p.src_off[e] - k * p.plane, sj = r2 / p.P, si = r2 - sj * p.P;
      const int tl = (sj - strip * p.tys) * p.P4 + (si >> 2) - t_first;
      if ((unsigned)tl < (unsigned)FDTD_BLOCK) {
        const long long ts = step - p.src_delay[e];
        if (ts >= 0 && ts < p.nsig) s_dense[(p.src_comp[e] * FDTD_BLOCK + tl) * 4 + (si & 3)] = p.src_amp[e] * p.sig[ts];
      }
    }
    __syncthreads();
  }
  if (valid) {

  // One component at a time — coefficients, update, sources, store — with a scheduling fence between the components:
  // left to itself the scheduler interleaves all three (12 LDS table reads, 3 metric float4, 6 coefficient float4 live
  // at once), which costs the variant without CPML branches more registers than its occupancy target has (it spilled).
  const int nsrc_t = (FUSE && !dense) ? srng.y - srng.x : 0;
  uchar4 cc = make_uchar4(0, 0, 0, 0);
  if (COEF == 2) cc = *reinterpret_cast<const uchar4*>(p.ecls + off);
#pragma unroll
  for (int comp = 0; comp < 3; ++comp) {
    float4& v = comp == 0 ? vx : (comp == 1 ? vy : vz);
    const float4& d1 = comp == 0 ? dx1 : (comp == 1 ? dy1 : dz1);
    const float4& d2 = comp == 0 ? dx2 : (comp == 1 ? dy2 : dz2);
    float4 a, b;
    if (COEF == 0) {
      a = ld4(p.vv + comp * p.nloc + off); b = ld4(p.vi + comp * p.nloc + off);
    } else {
      const float4 ex = ld4(p.emet[comp][0] + i0);
      const float m = p.emet[comp][1][j] * p.emet[comp][2][k];
      float2 l0, l1, l2, l3;
      if (COEF == 1) {
        const uchar4 c1 = *reinterpret_cast<const uchar4*>(p.ecls + comp * p.nloc + off);
        l0 = s_lut[c1.x]; l1 = s_lut[c1.y]; l2 = s_lut[c1.z]; l3 = s_lut[c1.w];
      } else {
        l0 = s_lut[3 * cc.x + comp]; l1 = s_lut[3 * cc.y + comp]; l2 = s_lut[3 * cc.z + comp]; l3 = s_lut[3 * cc.w + comp];
      }
      a = make_float4(l0.x, l1.x, l2.x, l3.x);
      b = make_float4(l0.y * (ex.x * m), l1.y * (ex.y * m), l2.y * (ex.z * m), l3.y * (ex.w * m));
    }
    v = upd4(a, v, b, d1, d2);
    if (FUSE && nsrc_t > 0) apply_staged(s_src, nsrc_t, comp, off, v);   // V += amp * sig[step - delay] on the (few) source edges of this strip-plane
    if (FUSE && dense) {
      const float4 a = reinterpret_cast<const float4*>(s_dense)[comp * FDTD_BLOCK + threadIdx.x];
      if (a.x != 0.f) v.x = v.x + a.x;
      if (a.y != 0.f) v.y = v.y + a.y;
      if (a.z != 0.f) v.z = v.z + a.z;
      if (a.w != 0.f) v.w = v.w + a.w;
    }
    if (WF) sto4_dev(p.V[comp], (unsigned)off, v);
    else sto4s(p.nt, p.V[comp], (unsigned)off, v);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (MUR) mur_post_inline<1, WF>(p, *mur, k, j, i0, mv, vx, vy, vz, WF ? (int)(step & 1) : 0);
  if (P2P && k == 0 && p.mb_out_E != nullptr) {   // push the new Vx, Vy of the bottom plane into the lower rank's mailbox
    float* mb = p.mb_out_E + (size_t)(step & 1) * 2 * mb_slot_words(p);
    mb_push(mb, (unsigned)(j * p.P + i0), (unsigned)step + 1u, vx);
    mb_push(mb + mb_slot_words(p), (unsigned)(j * p.P + i0), (unsigned)step + 1u, vy);
  }
  }   // valid
  if (WF) wf_publish(p, p.wf_flags, k, strip, pb, wf_target);
}

template <int COEF, bool PML, bool FUSE, bool P2P>
__global__ __launch_bounds__(FDTD_BLOCK, (PML || P2P) ? FDTD_E_MINBLOCKS - 1 : FDTD_E_MINBLOCKS) void k_update_E(const DevParams p, const int k_begin, const FastDiv fd_ps,
                                                                            const long long step, const int extra, const unsigned nb_main) {
  extern __shared__ float2 s_lut[];   // coefficient table, lut_n entries (dynamic: scenes use a few dozen of the up to 768)
  // CPML psi staging (LDS-DMA, 16 KiB); the probe block borrows it for its reduction
  __shared__ float4 s_psi[(PML && FDTD_PSI_STAGE) ? FDTD_BLOCK * PSI_SLOTS : (FUSE ? 3 * FDTD_BLOCK : 1)];   // (fused sources: the dense image, 12 KiB)
  __shared__ float s_xc[(PML && FDTD_PSI_STAGE) ? 3 * XC_MAX : 1];   // (b, c, 1/kappa) of the x-layer cells, by psi slot
  double* const s_red = reinterpret_cast<double*>(s_psi);
  __shared__ SrcStage s_src;
  if (FUSE && extra && blockIdx.x >= gridDim.x - (unsigned)extra) {   // probe blocks (one per probe): H-probes of the step just finished
    probe_block(p, FDTD_KIND_I, step - 1, s_red, (int)(blockIdx.x - (gridDim.x - (unsigned)extra)));
    return;
  }
  int strip, kk, pb, k;
  if (P2P) {   // all planes in one launch, the halo-dependent bottom plane first (or last) in dispatch order
    if (!decode_block_p2p(p, fd_ps, p.fd_nbs, nb_main, (unsigned)(p.nstrips * p.nbs), 0, 1, p.p2p_dep_first, strip, k, pb)) return;
  } else {
    if (!decode_block_fd(p, fd_ps, p.fd_nbs, 0, strip, kk, pb)) return;
    k = k_begin + kk;
  }
  body_E<COEF, PML, FUSE, P2P, false>(p, strip, k, pb, step, s_lut, s_psi, s_xc, s_src, 0u);
}

// update_E of a single slab with Mur faces, sources fused: the "post" pass rides along (one launch less per timestep).
template <int COEF, bool PML>
__global__ __launch_bounds__(FDTD_BLOCK, FDTD_E_MINBLOCKS) void k_update_E_mur(const DevParams p, const int k_begin, const FastDiv fd_ps,
                                                                                const long long step, const int extra, const MurDev m) {
  extern __shared__ float2 s_lut[];
  __shared__ float4 s_psi[(PML && FDTD_PSI_STAGE) ? FDTD_BLOCK * PSI_SLOTS : 3 * FDTD_BLOCK];
  __shared__ float s_xc[(PML && FDTD_PSI_STAGE) ? 3 * XC_MAX : 1];
  __shared__ SrcStage s_src;
  if (extra && blockIdx.x >= gridDim.x - (unsigned)extra) {   // probe blocks (one per probe): H-probes of the step just finished
    probe_block(p, FDTD_KIND_I, step - 1, reinterpret_cast<double*>(s_psi), (int)(blockIdx.x - (gridDim.x - (unsigned)extra)));
    return;
  }
  int strip, kk, pb;
  if (!decode_block_fd(p, fd_ps, p.fd_nbs, 0, strip, kk, pb)) return;
  body_E<COEF, PML, true, false, false, true>(p, strip, k_begin + kk, pb, step, s_lut, s_psi, s_xc, s_src, 0u, &m);
}

// ---- Mur without an apply pass (mur_direct) ----------------------------------------------------------------------------
// The candidates cd (what the apply pass would store on the boundary nodes) live BEHIND each voltage array, in the same allocation
// (build_mur_table): rows of stride P for y and z faces — a z face's candidates look like a field plane, a y face's like one row
// per plane — and [k][j] for x faces.  So a thread that needs a boundary voltage loads the candidate by OFFSET from the same
// base: no second pointer, and for whole rows / planes no extra load at all (mur_load_V).
// Element e (run-time) of a vector replaced / read with BIT MASKS: chains of selects on the element index become an insertelement /
// extractelement with a variable index, which the backend serves from scratch (48 - 64 bytes per lane in the k_step<..., MUR> variants).
__device__ __forceinline__ float f1_pick(const float old, const float a, const bool take) {
  const unsigned m = 0u - (unsigned)take;
  return __uint_as_float(__float_as_uint(old) ^ ((__float_as_uint(old) ^ __float_as_uint(a)) & m));
}
__device__ __forceinline__ void f4_put(float4& v, const int e, const float c) {
  v.x = f1_pick(v.x, c, e == 0); v.y = f1_pick(v.y, c, e == 1); v.z = f1_pick(v.z, c, e == 2); v.w = f1_pick(v.w, c, e == 3);
}
// The nine voltage loads of an H thread (cells i0..i0+3 of row j in plane k; rows j+1, plane k+1 and cell i0+4 beside them) with
// every node of a Mur face taken from the candidates.  z faces win over y faces over x faces (the apply order).  Rows, planes
// and cells beyond the grid keep the addresses of the plain kernel (their values meet zero weights).  DEV: device-scope loads.
// Two phases: mur_load_V issues the loads; mur_finish_V, called once the caller has issued its other loads as well (I, psi), puts
// the x-face candidates into their cells — it is the first use of the loaded values, i.e. where the wave waits.
struct MurX { float c_vy, c_vz, c_vzjp, c_vykp; };   // the x-face candidates of the thread's boundary cell
__device__ __forceinline__ void mur_finish_V(const DevParams& p, const MurH& m, const MurX& x, const int k, const int j, const int i0,
                                             float4& vy, float4& vz, float4& vz_jp, float4& vy_kp) {
  // which cell, and whether a later face has the node (then the load took that face's candidates already): worked out again
  // rather than kept from mur_load_V — compares against scalars cost less than registers here
  int e = -1;
  if (m.b[0] == 0 && i0 == 0) e = 0;
  if (m.b[1] >= 0 && (unsigned)(m.b[1] - i0) < 4u) e = m.b[1] - i0;
  const bool zk = k == m.b[4] || k == m.b[5], zk1 = (k + 1 == m.b[4] || k + 1 == m.b[5]) && k + 1 < p.nk;
  const bool yj = j == m.b[2] || j == m.b[3], yj1 = (j + 1 == m.b[2] || j + 1 == m.b[3]) && j + 1 < p.ny;
  f4_put(vy, zk ? -1 : e, x.c_vy);
  f4_put(vz, yj ? -1 : e, x.c_vz);
  f4_put(vz_jp, (yj1 || j + 1 >= p.ny) ? -1 : e, x.c_vzjp);
  f4_put(vy_kp, (zk1 || k + 1 >= p.nk) ? -1 : e, x.c_vykp);
}
template <bool DEV>
__device__ __forceinline__ MurX mur_load_V(const DevParams& p, const int k, const int j, const int i0, const unsigned uo, const bool ip_load,
                                           float4& vx, float4& vy, float4& vz, float4& vz_jp, float4& vx_jp, float4& vy_kp, float4& vx_kp,
                                           float& vz_ip, float& vy_ip, const MurH& m, const int par = 0) {
  const unsigned P = (unsigned)p.P, rj = (unsigned)(j * p.P + i0), rk = (unsigned)(k * p.P + i0), xrow = (unsigned)(k * p.ny + j);
  // (one-launch schedule: the cd copy of this timestep)
  const int c00 = m.co[0][0] + par * m.cdn[0], c01 = m.co[0][1] + par * m.cdn[0], c10 = m.co[1][0] + par * m.cdn[1], c11 = m.co[1][1] + par * m.cdn[1];
  const int c20 = m.co[2][0] + par * m.cdn[2], c21 = m.co[2][1] + par * m.cdn[2], c30 = m.co[3][0] + par * m.cdn[3], c31 = m.co[3][1] + par * m.cdn[3];
  const int c40 = m.co[4][0] + par * m.cdn[4], c41 = m.co[4][1] + par * m.cdn[4], c50 = m.co[5][0] + par * m.cdn[5], c51 = m.co[5][1] + par * m.cdn[5];
  const bool jp = j + 1 < p.ny, kp = k + 1 < p.nk;
  // (selects, not branches: a face that is off has b = -1, never equal to a plane, row or cell index)
  // z faces: candidate offsets of Vx, Vy for plane k / plane k + 1 (block-uniform: scalar selects)
  const int zx = k == m.b[4] ? c40 : k == m.b[5] ? c50 : -1, zy = k == m.b[4] ? c41 : k == m.b[5] ? c51 : -1;
  const int zx1 = k + 1 == m.b[4] ? c40 : k + 1 == m.b[5] ? c50 : -1, zy1 = k + 1 == m.b[4] ? c41 : k + 1 == m.b[5] ? c51 : -1;
  // y faces: of Vz, Vx for row j / row j + 1
  const bool y2 = j == m.b[2], y3 = j == m.b[3], y2p = j + 1 == m.b[2], y3p = j + 1 == m.b[3];
  const int yz = y2 ? c20 : y3 ? c30 : -1, yx = y2 ? c21 : y3 ? c31 : -1;
  const int yz1 = y2p ? c20 : y3p ? c30 : -1, yx1 = y2p ? c21 : y3p ? c31 : -1;
  MurX x;
  // x faces: one of the thread's four cells at most (nx >= 6); its four candidates are loads of their own, issued first
  const bool xlo = m.b[0] == 0 && i0 == 0, xhi = m.b[1] >= 0 && (unsigned)(m.b[1] - i0) < 4u;
  const int e = xhi ? m.b[1] - i0 : xlo ? 0 : -1;
  const int cxy = xhi ? c10 : c00, cxz = xhi ? c11 : c01;
  const bool xh_ip = m.b[1] >= 0 && i0 + 4 == m.b[1];
  const unsigned o_vx = zx >= 0 ? zx + rj : yx >= 0 ? yx + rk : uo;
  const unsigned o_vy = zy >= 0 ? zy + rj : uo;
  const unsigned o_vz = yz >= 0 ? yz + rk : uo;
  const unsigned o_vzjp = (yz1 >= 0 && jp) ? yz1 + rk : uo + P;
  const unsigned o_vxjp = (zx >= 0 && jp) ? zx + rj + P : (yx1 >= 0 && jp) ? yx1 + rk : uo + P;
  const unsigned o_vykp = (zy1 >= 0 && kp) ? zy1 + rj : uo + (unsigned)p.plane;
  const unsigned o_vxkp = (zx1 >= 0 && kp) ? zx1 + rj : (yx >= 0 && kp) ? yx + rk + P : uo + (unsigned)p.plane;
  const bool in = i0 + 4 < p.nx;
  unsigned o_vzip = (in && yz >= 0) ? yz + rk + 4u : xh_ip ? c11 + xrow : uo + 4u;
  unsigned o_vyip = (in && zy >= 0) ? zy + rj + 4u : xh_ip ? c10 + xrow : uo + 4u;
  if (!ip_load) o_vzip = o_vyip = (unsigned)__builtin_amdgcn_readfirstlane((int)uo);   // (lane shift: the value comes from the neighbour lane)
  float c_vy = 0.f, c_vz = 0.f, c_vzjp = 0.f, c_vykp = 0.f;
  if (DEV) {
    const DevRsrc b0 = dev_buf(p.V[0]), b1 = dev_buf(p.V[1]), b2 = dev_buf(p.V[2]);
    if (e >= 0) {
      c_vy = ldb1_dev(b1, (cxy + xrow) << 2, 0u); c_vz = ldb1_dev(b2, (cxz + xrow) << 2, 0u);
      c_vzjp = ldb1_dev(b2, (cxz + xrow + (jp ? 1u : 0u)) << 2, 0u); c_vykp = ldb1_dev(b1, (cxy + xrow + (kp ? (unsigned)p.ny : 0u)) << 2, 0u);
    }
    vx = ldb4_dev(b0, o_vx << 2, 0u); vy = ldb4_dev(b1, o_vy << 2, 0u); vz = ldb4_dev(b2, o_vz << 2, 0u);
    vz_jp = ldb4_dev(b2, o_vzjp << 2, 0u); vx_jp = ldb4_dev(b0, o_vxjp << 2, 0u);
    vy_kp = ldb4_dev(b1, o_vykp << 2, 0u); vx_kp = ldb4_dev(b0, o_vxkp << 2, 0u);
    vz_ip = ldb1_dev(b2, o_vzip << 2, 0u); vy_ip = ldb1_dev(b1, o_vyip << 2, 0u);
  } else {
    if (e >= 0) {
      c_vy = ldo1(p.V[1], cxy + xrow); c_vz = ldo1(p.V[2], cxz + xrow);
      c_vzjp = ldo1(p.V[2], cxz + xrow + (jp ? 1u : 0u)); c_vykp = ldo1(p.V[1], cxy + xrow + (kp ? (unsigned)p.ny : 0u));
    }
    vx = ldo4(p.V[0], o_vx); vy = ldo4(p.V[1], o_vy); vz = ldo4(p.V[2], o_vz);
    vz_jp = ldo4(p.V[2], o_vzjp); vx_jp = ldo4(p.V[0], o_vxjp);
    vy_kp = ldo4(p.V[1], o_vykp); vx_kp = ldo4(p.V[0], o_vxkp);
    vz_ip = ldo1(p.V[2], o_vzip); vy_ip = ldo1(p.V[1], o_vyip);
  }
  x.c_vy = c_vy; x.c_vz = c_vz; x.c_vzjp = c_vzjp; x.c_vykp = c_vykp;
  return x;
}

__device__ __forceinline__ float f4_get(const float4& v, const int e) {
  const unsigned m0 = 0u - (unsigned)(e == 0), m1 = 0u - (unsigned)(e == 1), m2 = 0u - (unsigned)(e == 2), m3 = 0u - (unsigned)(e == 3);
  return __uint_as_float((__float_as_uint(v.x) & m0) | (__float_as_uint(v.y) & m1) | (__float_as_uint(v.z) & m2) | (__float_as_uint(v.w) & m3));
}
__device__ __forceinline__ float4 mur_pre4(const float c, const float4& vb, const float4& vi) {   // k_mur mode 0, four cells
  return make_float4(__builtin_fmaf(-c, vb.x, vi.x), __builtin_fmaf(-c, vb.y, vi.y), __builtin_fmaf(-c, vb.z, vi.z), __builtin_fmaf(-c, vb.w, vi.w));
}
// What is left of the apply pass and the pre pass of the next timestep, done by the H threads that hold the values anyway (all
// of them "applied": mur_load_V): (i) a thread whose own cells lie on a face stores their boundary voltages — nobody in this
// launch reads them there, the next update_E does; (ii) st = V_inner - coeff * V_boundary by the thread that has both: for a
// lower face the one on the boundary (its j + 1 / k + 1 neighbour is the inner node), for an upper face the one on the inner
// node.  As a pass of its own (one thread per face point, 12 rows of blocks behind the main ones) this cost 2.4 / 5.6 / 11.5 us
// on 200x200x40 / 300x300x60 / 400x400x80; here it is a few stores of the boundary threads.
template <bool DEV = false>   // DEV (one-launch schedule): the E blocks of the next timestep of the same launch read these: write-through stores
__device__ __forceinline__ void mur_pre_store(const DevParams& p, const MurH& m, const int k, const int j, const int i0, const unsigned uo,
                                              const float4& vx, const float4& vy, const float4& vz, const float4& vz_jp, const float4& vx_jp,
                                              const float4& vy_kp, const float4& vx_kp, const float vz_ip, const float vy_ip) {
#ifdef FDTD_DIAG_MUR_NO_PRE   // timing experiments only (tools/build_variant.sh): what the main blocks cost without this — wrong fields
  return;
#endif
  const unsigned rj = (unsigned)(j * p.P + i0), rk = (unsigned)(k * p.P + i0), xrow = (unsigned)(k * p.ny + j);
  auto S4 = [](float* base, const unsigned e, const float4& v) { if (DEV) sto4_dev(base, e, v); else sto4s(0, base, e, v); };
  auto S1 = [](float* base, const unsigned e, const float v) { if (DEV) sto1_dev(base, e, v); else base[e] = v; };
  // (which faces the thread touches is worked out again rather than kept from mur_load_V: compares against scalars cost less than registers here)
  const bool onz = k == m.b[4] || k == m.b[5], ony = j == m.b[2] || j == m.b[3];
  const bool onx = (m.b[0] == 0 && i0 == 0) || (m.b[1] >= 0 && (unsigned)(m.b[1] - i0) < 4u);
  if (onz || ony) S4(p.V[0], uo, vx);
  if (onz || onx) S4(p.V[1], uo, vy);
  if (ony || onx) S4(p.V[2], uo, vz);
  if (k == m.b[4]) { S4(p.V[0], m.so[4][0] + rj, mur_pre4(m.coeff[4], vx, vx_kp)); S4(p.V[1], m.so[4][1] + rj, mur_pre4(m.coeff[4], vy, vy_kp)); }
  if (k + 1 == m.b[5]) { S4(p.V[0], m.so[5][0] + rj, mur_pre4(m.coeff[5], vx_kp, vx)); S4(p.V[1], m.so[5][1] + rj, mur_pre4(m.coeff[5], vy_kp, vy)); }
  if (j == m.b[2]) { S4(p.V[2], m.so[2][0] + rk, mur_pre4(m.coeff[2], vz, vz_jp)); S4(p.V[0], m.so[2][1] + rk, mur_pre4(m.coeff[2], vx, vx_jp)); }
  if (j + 1 == m.b[3]) { S4(p.V[2], m.so[3][0] + rk, mur_pre4(m.coeff[3], vz_jp, vz)); S4(p.V[0], m.so[3][1] + rk, mur_pre4(m.coeff[3], vx_jp, vx)); }
  if (m.b[0] == 0 && i0 == 0) {
    S1(p.V[1], m.so[0][0] + xrow, __builtin_fmaf(-m.coeff[0], vy.x, vy.y));
    S1(p.V[2], m.so[0][1] + xrow, __builtin_fmaf(-m.coeff[0], vz.x, vz.y));
  }
  const int ei = m.b[1] - 1 - i0;   // the inner node of the upper x face among the thread's cells
  if (m.b[1] >= 0 && (unsigned)ei < 4u) {
    const float by = ei < 3 ? f4_get(vy, ei + 1) : vy_ip, bz = ei < 3 ? f4_get(vz, ei + 1) : vz_ip;
    S1(p.V[1], m.so[1][0] + xrow, __builtin_fmaf(-m.coeff[1], by, f4_get(vy, ei)));
    S1(p.V[2], m.so[1][1] + xrow, __builtin_fmaf(-m.coeff[1], bz, f4_get(vz, ei)));
  }
}

// Mur "pre" pass (mode 0 of k_mur) of one block: S = V_inner - coeff * V_boundary on the values BEFORE the next E update.
// e = block index among the Mur blocks of the launch: (face * 2 + tangential component) * mur_nbx + block within the face.
__device__ __forceinline__ void mur_pre_block(const DevParams& p, const unsigned e) {
  const MurDev& m = *p.mur;
  const unsigned row = e / (unsigned)p.mur_nbx, bx = e - row * (unsigned)p.mur_nbx;
  const int fi = (int)(row >> 1), t = (int)(row & 1u);
  const MurDevFace& f = m.f[fi];
  if (!f.on) return;
  const int s = (int)(bx * FDTD_BLOCK + threadIdx.x);
  if (s >= f.du * f.dv) return;
  const int iv = s / f.du, iu = s - iv * f.du;
  const int stride[3] = {1, p.P, p.plane};
  int pos[3];
  pos[f.a] = f.b; pos[f.ua] = iu; pos[f.va] = iv;
  const float* V = p.V[f.comp[t]];
  const int ob = pos[0] + pos[1] * p.P + pos[2] * p.plane;
  const int oi = ob + (f.in - f.b) * stride[f.a];
  f.st[t][iv * f.cs + iu] = __builtin_fmaf(-f.coeff, V[ob], V[oi]);
}

// ------------------------------------------------------------------------------------------------
// K2: H half-step
// ------------------------------------------------------------------------------------------------
// MUR: Mur faces without an apply pass — the block takes the boundary voltages it reads from the candidates (mur_sub4).
template <bool RAW, bool PML, bool P2P, bool WF, bool MULTI = false, bool MUR = false>
__device__ __forceinline__ void body_H(const DevParams& p, const int strip, const int k, const int pb, const long long step,
                                       float4* const s_psi, float* const s_xc, const unsigned wf_target, const unsigned back_target = 0u,
                                       const MurH* const mh = nullptr) {
  static_assert(!(MULTI && P2P), "several timesteps per launch: single slabs only");
  int j = 0, i0 = 0;
  const bool staged = PML && FDTD_PSI_STAGE;
  const bool xc_lds = staged && p.xc_tab != nullptr;
  if (xc_lds && (int)threadIdx.x < 3 * XC_MAX / 4)   // H-located (b, c, 1/kappa) of the x-layer cells: second half of the table
    glds16(p.xc_tab + 3 * XC_MAX + 4 * (int)threadIdx.x, __builtin_amdgcn_readfirstlane(lds_off(s_xc) + (threadIdx.x >> 6) * 1024u));
  // with psi staging the block meets at one barrier (x-layer coefficient table): out-of-range threads of a strip's last
  // block read their block's first group (in range by construction) and leave after it
  const bool valid = decode_thread(p, strip, pb, j, i0);
  if (!WF && !staged && !valid) return;
  const int off = k * p.plane + (valid ? j * p.P + i0 : 0);

  const unsigned uo = (unsigned)off;     // H reads planes k, k+1 only: offsets from plane 0 are never negative
  const bool dep_in = P2P && k == p.nk - 1 && p.mb_in_E != nullptr;   // k+1 is the upper rank's bottom plane: mailbox
  float4 vx, vy, vz, vz_jp, vx_jp, ix, iy, iz;
  float4 vy_kp = make_float4(0.f, 0.f, 0.f, 0.f), vx_kp = vy_kp;
  float vz_ip, vy_ip;
  MurX mx;
  // (lane shift) the thread whose right-hand neighbour thread is in another wave, or does not exist (end of the strip-plane), loads
  constexpr bool LS = FDTD_LANE_SHIFT && LANE_SHIFT_FITS(WF, MULTI);
  const bool ip_load = !LS || (threadIdx.x & 63u) == 63u ||
                       pb * FDTD_BLOCK + (int)threadIdx.x + 1 >= min(p.tys, p.ny - strip * p.tys) * p.P4;
  if (!WF) {
    if (MUR) {
      mx = mur_load_V<false>(p, k, j, i0, uo, ip_load, vx, vy, vz, vz_jp, vx_jp, vy_kp, vx_kp, vz_ip, vy_ip, *mh);
    } else {
    vx = ldo4(p.V[0], uo); vy = ldo4(p.V[1], uo); vz = ldo4(p.V[2], uo);
    vz_jp = ldo4(p.V[2] + p.P, uo); vx_jp = ldo4(p.V[0] + p.P, uo);   // neighbour displacements in the scalar bases: one offset VGPR
    if (!dep_in) { vy_kp = ldo4(p.V[1] + p.plane, uo); vx_kp = ldo4(p.V[0] + p.plane, uo); }
    {
      const unsigned ue = ip_load ? uo : (unsigned)__builtin_amdgcn_readfirstlane((int)uo);
      vz_ip = ldo1(p.V[2] + 4, ue); vy_ip = ldo1(p.V[1] + 4, ue);
    }
    }
    ix = ldo4(p.I[0], uo); iy = ldo4(p.I[1], uo); iz = ldo4(p.I[2], uo);
    if (staged)
      psi_stage_issue(p, p.psiH, __builtin_amdgcn_readfirstlane(lds_off(s_psi) + (threadIdx.x >> 6) * (PSI_SLOTS * 1024u)), valid, k, j, i0);
  } else {
    // wavefront block: everything that does not depend on this launch's E blocks first (I, psi), then the flags of the E
    // blocks this block reads from, then V with device-scope (sc1) loads — they bypass this CU's L1, which no other CU's
    // store ever refreshes, and are served by L2 / the Infinity Cache, where the write-through stores of body_E land
    if (!MULTI) {
      ix = ldo4(p.I[0], uo); iy = ldo4(p.I[1], uo); iz = ldo4(p.I[2], uo);
      if (staged)
        psi_stage_issue(p, p.psiH, __builtin_amdgcn_readfirstlane(lds_off(s_psi) + (threadIdx.x >> 6) * (PSI_SLOTS * 1024u)), valid, k, j, i0);
    }
    // MUR: the E blocks whose CANDIDATES this block loads, too.  Away from the y and z faces these are E blocks it waits for anyway — a candidate
    // it reads was written by the thread of the same cells (x faces; the row / plane in front of an upper face) —; a block that holds a row or a
    // plane ON a y / z face (candidates from the row / plane behind it, or one further ahead than the plain set reaches), and every block of a grid
    // whose upper x face starts a four-cell group (the inner node belongs to the thread before), takes the wide set (wf_wait_mur).  The wide set for
    // ALL blocks cost 2.3 us per timestep on 143x129x89 (the E -> H turn of a grid of one residency round).
    bool wide = false;
    if (MUR) {
      const MurH& m = *mh;
      const int rows = min(p.tys, p.ny - strip * p.tys);
      const int jlo = strip * p.tys + (int)fd_div((unsigned)(pb * FDTD_BLOCK), p.fd_P4);
      const int jhi = strip * p.tys + min(rows - 1, (int)fd_div((unsigned)(pb * FDTD_BLOCK + FDTD_BLOCK - 1), p.fd_P4));
      wide = k == m.b[4] || k == m.b[5] || (m.b[2] >= jlo && m.b[2] <= jhi) || (m.b[3] >= jlo && m.b[3] <= jhi) ||
             (m.b[1] >= 0 && (m.b[1] & 3) == 0);
    }
    if (wide) wf_wait_mur(p, k, strip, wf_target + p.wf_wait_bias);
    else wf_wait(p, k, strip, pb, wf_target + p.wf_wait_bias);
    if (MULTI) {
      // Several timesteps per launch: this block's I and psi were written by another workgroup of the SAME launch — the H block
      // of these cells one timestep ago.  That it has finished is only known HERE: the E block of these cells waited for its
      // flag, and this block has just seen the E block's.  Device-scope loads; and the probe blocks of the previous timestep
      // have read the I cells of this strip-plane before it overwrites them.
      if (back_target && p.wf_prb_sp != nullptr && sload_int(p.wf_prb_sp + (k * p.nstrips + strip)) != 0) wf_wait_probes(p, FDTD_KIND_I, back_target);
      ix = ldb4_dev(dev_buf(p.I[0]), uo << 2, 0u); iy = ldb4_dev(dev_buf(p.I[1]), uo << 2, 0u); iz = ldb4_dev(dev_buf(p.I[2]), uo << 2, 0u);
      if (staged)
        psi_stage_issue<true>(p, p.psiH, __builtin_amdgcn_readfirstlane(lds_off(s_psi) + (threadIdx.x >> 6) * (PSI_SLOTS * 1024u)), valid, k, j, i0);
    }
    if (MUR) {
      mx = mur_load_V<true>(p, k, j, i0, uo, ip_load, vx, vy, vz, vz_jp, vx_jp, vy_kp, vx_kp, vz_ip, vy_ip, *mh, (int)(step & 1));
    } else {
    const DevRsrc b0 = dev_buf(p.V[0]), b1 = dev_buf(p.V[1]), b2 = dev_buf(p.V[2]);
    const unsigned bo = uo << 2;
    vx = ldb4_dev(b0, bo, 0u); vy = ldb4_dev(b1, bo, 0u); vz = ldb4_dev(b2, bo, 0u);
    vz_jp = ldb4_dev(b2, bo, (unsigned)p.P << 2); vx_jp = ldb4_dev(b0, bo, (unsigned)p.P << 2);
    if (!dep_in) { vy_kp = ldb4_dev(b1, bo, (unsigned)p.plane << 2); vx_kp = ldb4_dev(b0, bo, (unsigned)p.plane << 2); }
    {
      const unsigned be = ip_load ? bo : (unsigned)__builtin_amdgcn_readfirstlane((int)bo);
      vz_ip = ldb1_dev(b2, be, 16u); vy_ip = ldb1_dev(b1, be, 16u);
    }
    }
  }
  if (dep_in) {   // E halo of this step (tag = step + 1)
    const float* mb = p.mb_in_E + (size_t)(step & 1) * 2 * mb_slot_words(p);
    mb_pull2(mb, mb + mb_slot_words(p), (unsigned)(j * p.P + i0), (unsigned)step + 1u + p.p2p_tag_bias, vx_kp, vy_kp, p.p2p_err, p.p2p_limit,
             0x10000000u | (WF ? 0x20000000u : 0u) | ((unsigned)strip << 14) | (unsigned)pb);   // who: H half-step (bit 28)
  }

  if (MUR) mur_finish_V(p, *mh, mx, k, j, i0, vy, vz, vz_jp, vy_kp);
  if (LS) {
    const float zn = lane_next(vz.x), yn = lane_next(vy.x);
    if (!ip_load) { vz_ip = zn; vy_ip = yn; }
  }
  if (MUR && valid) mur_pre_store<WF>(p, *mh, k, j, i0, uo, vx, vy, vz, vz_jp, vx_jp, vy_kp, vx_kp, vz_ip, vy_ip);
  float4 dx1 = sub4(vz, vz_jp), dx2 = sub4(vy, vy_kp);
  float4 dy1 = sub4(vx, vx_kp);
  float4 dy2 = make_float4(vz.x - vz.y, vz.y - vz.z, vz.z - vz.w, vz.w - vz_ip);
  float4 dz1 = make_float4(vy.x - vy.y, vy.y - vy.z, vy.z - vy.w, vy.w - vy_ip);
  float4 dz2 = sub4(vx, vx_jp);

  if (staged) {   // one barrier, where every wave has its loads anyway (see update_E): x-layer coefficient table + staged psi are in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!WF && !valid) return;
  }
  // wavefront block of a strip-plane that holds I-probe cells: its I goes out write-through and it publishes a flag of its
  // own, for the probe blocks at the end of the launch (wf_probe_tail); such a block meets once more, so nobody leaves early
  // (several timesteps per launch: EVERY H block does — the E blocks of the next timestep wait for these flags)
  const bool pub = MULTI || (WF && p.wf_prb_sp != nullptr && sload_int(p.wf_prb_sp + (k * p.nstrips + strip)) != 0);
  if (!WF || valid) {

  if (PML) {
    if (FDTD_PSI_STAGE) {
      psi_stage_apply<MULTI>(p, p.psiH, 1, s_psi, s_xc, xc_lds, k, j, i0, dx2, dy1, dy2, dz1, dx1, dz2);
    } else {
      const int sy = pml_slot(p, 1, j);
      if (sy >= 0) {
        const float b = p.cp[1][1][0][j], c = p.cp[1][1][1][j], ik = p.cp[1][1][2][j];
        const int o = (k * p.nslot[1] + sy) * p.P + i0;
        cpml_row4(dx1, p.psiH[0][0], (unsigned)o, b, c, ik);
        cpml_row4(dz2, p.psiH[2][1], (unsigned)o, b, c, ik);
      }
      const int sz = pml_slot(p, 2, k);
      if (sz >= 0) {
        const float b = p.cp[2][1][0][k], c = p.cp[2][1][1][k], ik = p.cp[2][1][2][k];
        const int o = (sz * p.ny + j) * p.P + i0;
        cpml_row4(dx2, p.psiH[0][1], (unsigned)o, b, c, ik);
        cpml_row4(dy1, p.psiH[1][0], (unsigned)o, b, c, ik);
      }
      if (i0 < p.pml_lo[0] || i0 >= p.pml_hi[0])
        cpml_x4(p, 1, i0, k * p.xplane + j * p.xrs, dy2, p.psiH[1][1], dz1, p.psiH[2][0]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  if (RAW) {
    const float4 ax = ld4(p.ii + off), bx = ld4(p.iv + off);
    const float4 ay = ld4(p.ii + p.nloc + off), by = ld4(p.iv + p.nloc + off);
    const float4 az = ld4(p.ii + 2 * p.nloc + off), bz = ld4(p.iv + 2 * p.nloc + off);
    ix = upd4(ax, ix, bx, dx1, dx2);
    iy = upd4(ay, iy, by, dy1, dy2);
    iz = upd4(az, iz, bz, dz1, dz2);
  } else {
    // ii = 1: fmaf(1, I, t) == I + t exactly
    const float4 hx0 = ld4(p.hmet[0][0] + i0), hx1 = ld4(p.hmet[1][0] + i0), hx2 = ld4(p.hmet[2][0] + i0);
    const float m0 = p.hmet[0][1][j] * p.hmet[0][2][k];
    const float m1 = p.hmet[1][1][j] * p.hmet[1][2][k];
    const float m2 = p.hmet[2][1][j] * p.hmet[2][2][k];
    ix = make_float4(ix.x + (hx0.x * m0) * (dx1.x - dx2.x), ix.y + (hx0.y * m0) * (dx1.y - dx2.y),
                     ix.z + (hx0.z * m0) * (dx1.z - dx2.z), ix.w + (hx0.w * m0) * (dx1.w - dx2.w));
    iy = make_float4(iy.x + (hx1.x * m1) * (dy1.x - dy2.x), iy.y + (hx1.y * m1) * (dy1.y - dy2.y),
                     iy.z + (hx1.z * m1) * (dy1.z - dy2.z), iy.w + (hx1.w * m1) * (dy1.w - dy2.w));
    iz = make_float4(iz.x + (hx2.x * m2) * (dz1.x - dz2.x), iz.y + (hx2.y * m2) * (dz1.y - dz2.y),
                     iz.z + (hx2.z * m2) * (dz1.z - dz2.z), iz.w + (hx2.w * m2) * (dz1.w - dz2.w));
  }
  if (pub) { sto4_dev(p.I[0], uo, ix); sto4_dev(p.I[1], uo, iy); sto4_dev(p.I[2], uo, iz); }
  else { sto4s(p.nt, p.I[0], uo, ix); sto4s(p.nt, p.I[1], uo, iy); sto4s(p.nt, p.I[2], uo, iz); }
  if (P2P && k == p.nk - 1 && p.mb_out_H != nullptr) {   // push the new Ix, Iy of the top plane into the upper rank's mailbox
    float* mb = p.mb_out_H + (size_t)(step & 1) * 2 * mb_slot_words(p);
    mb_push(mb, (unsigned)(j * p.P + i0), (unsigned)step + 2u, ix);
    mb_push(mb + mb_slot_words(p), (unsigned)(j * p.P + i0), (unsigned)step + 2u, iy);
  }
  }   // valid
  if (pub) wf_publish(p, p.wf_flagsH, k, strip, pb, wf_target);
}

template <bool RAW, bool PML, bool P2P, bool MUR = false>
__global__ __launch_bounds__(FDTD_BLOCK, (RAW || PML) ? FDTD_H_MINBLOCKS - 1 : FDTD_H_MINBLOCKS) void k_update_H(const DevParams p, const int k_begin, const FastDiv fd_ps,
                                                                            const long long step, const int extra, const unsigned nb_main, const MurH mh) {
  // CPML psi staging (LDS-DMA, 16 KiB); the probe block borrows it for its reduction
  __shared__ float4 s_psi[(PML && FDTD_PSI_STAGE) ? FDTD_BLOCK * PSI_SLOTS : FDTD_BLOCK / 2];
  __shared__ float s_xc[(PML && FDTD_PSI_STAGE) ? 3 * XC_MAX : 1];
  double* const s_red = reinterpret_cast<double*>(s_psi);
  if (extra && blockIdx.x >= gridDim.x - (unsigned)extra) {   // the extra blocks at the end of the grid:
    const unsigned e = blockIdx.x - (gridDim.x - (unsigned)extra);
    if (e < (unsigned)p.mur_nb) mur_pre_block(p, e);          // Mur pre pass of the next step (V is final, H not read); MUR: the main blocks do it
    else probe_block(p, FDTD_KIND_V, step, s_red, (int)e - p.mur_nb);   // the last ones, one per probe: V-probes of this step
    return;
  }
  int strip, kk, pb, k;
  if (P2P) {   // all planes in one launch, the halo-dependent top plane first (or last) in dispatch order
    if (!decode_block_p2p(p, fd_ps, p.fd_nbs, nb_main, (unsigned)(p.nstrips * p.nbs), p.nk - 1, 0, p.p2p_dep_first, strip, k, pb)) return;
  } else {
    if (!decode_block_fd(p, fd_ps, p.fd_nbs, p.sweep_rev, strip, kk, pb)) return;
    k = k_begin + kk;
  }
  body_H<RAW, PML, P2P, false, false, MUR>(p, strip, k, pb, step, s_psi, s_xc, 0u, 0u, &mh);
}

// ------------------------------------------------------------------------------------------------
// K1+K2 in ONE launch per timestep: the E sweep runs `lag` planes ahead of the H sweep, coupled by per-block flags.
// Why: on grids beyond the 256 MiB Infinity Cache the two sweeps of a step each stream all six fields from HBM (72 B per
// cell and step).  Here H finds the V planes E has just written — and the I planes E has just read — in the Infinity
// Cache: HBM sees 48 B per cell and step (tools/streams/eh_interleave_probe.hip: -18..-20 % time at 512x512x128 and
// 800x800x120, nothing at 400x400x80, a loss on cache-resident grids, where the two-launch schedule stays).
// Dispatch order (block index = 8 * position + XCD group): per XCD group and plane group g: its share of the E blocks of
// plane g, then its share of the H blocks of plane g - lag.  An H block (k, strip, pb) reads V — and overwrites I values
// that E blocks read — of: its own cells, the rows above (same strip, or the next strip's first rows) and plane k + 1; it
// waits for exactly those E blocks' flags (RAW and WAR are the same set).  Every flag it waits for belongs to a block
// EARLIER in dispatch order (lag >= 1), so with in-order dispatch the wait cannot deadlock; it is bounded all the same
// (wf_limit ticks, then the error word: the run fails with FDTD_E_DEVICE instead of hanging).
// Visibility follows MI355X_MICROARCH.md "inter-workgroup visibility": every V store of an E block is write-through (sc1),
// every wave drains its stores (vmcnt(0)), the block meets, ONE lane publishes the flag with an sc1 store; the consumer's
// polling wave reads the flags with sc1 loads, the block meets, and EVERY load of V is an sc1 load to registers.
// ------------------------------------------------------------------------------------------------
#ifdef FDTD_XCD_TRACE
#define FDTD_XCD_TRACE_MAX 65536
__device__ unsigned long long g_xcd_trace[2 * FDTD_XCD_TRACE_MAX * 4];   // [step & 1][block][start, end, XCC_ID | E/H << 3 | valid << 4, step]
#endif
// MULTI: `fd_per.d` blocks per timestep (the main blocks, then the probe blocks, padded to a multiple of eight so that a share
// keeps its XCD from timestep to timestep), timestep after timestep in ONE launch: no kernel boundary, the E blocks of
// timestep s + 1 start while the H blocks of timestep s drain.  Only the order "all E blocks, then all H blocks" (lag < 0),
// every timestep in the same direction (an E block waits for the H block of its cells one timestep earlier: half a timestep's
// blocks back in dispatch order, long finished — walking backwards it would be the block dispatched last).
// MUR (single slabs, all E blocks then all H blocks): first-order Mur faces inside the launch — the post pass in the E blocks, no apply pass: the H blocks
// take the boundary voltages from the candidates (two copies alternating with the timestep: an E block of the next timestep must not overwrite what an H
// block of this one still reads), store them and run the pre pass (mur_load_V, mur_pre_store, wf_wait_mur).
template <int COEF, bool PML, bool P2P, bool MULTI = false, bool MUR = false>
__global__ __launch_bounds__(FDTD_BLOCK, (P2P || MUR) ? FDTD_WF_MINBLOCKS - 1 : FDTD_WF_MINBLOCKS) void k_step(const DevParams p, const long long step0, const int lag, const unsigned wf_target0,
                                                                        const unsigned nbp, const FastDiv fd_2m, const int down, const unsigned nmain,
                                                                        const FastDiv fd_per, const MurDev mur, const MurH mh) {
  static_assert(!(MUR && P2P), "Mur faces inside one launch: single slabs only");
  extern __shared__ float2 s_lut[];
  __shared__ float4 s_psi[(PML && FDTD_PSI_STAGE) ? FDTD_BLOCK * PSI_SLOTS : 3 * FDTD_BLOCK];   // (the probe blocks borrow it for their reduction; dense sources for their image)
  __shared__ float s_xc[(PML && FDTD_PSI_STAGE) ? 3 * XC_MAX : 1];
  __shared__ SrcStage s_src;
  unsigned b = blockIdx.x;
  long long step = step0;
  unsigned wf_target = wf_target0, ts = 0u;
  if (MULTI) {   // which timestep of the launch
    ts = fd_div(b, fd_per);
    b -= ts * fd_per.d;
    step += ts;
    wf_target += ts;
  }
  const unsigned x = b & 7u, pos = b >> 3;
  if (b >= nmain) {   // the last blocks of the timestep: one per probe
    if ((int)(b - nmain) < p.nprobe) wf_probe_tail<MULTI>(p, (int)(b - nmain), step, wf_target, reinterpret_cast<double*>(s_psi));
    return;
  }
  if (p.xstamp && b < 8u && threadIdx.x == 0) p.xstamp[p.xstamp_n + b] = wall_clock64();   // calibration launch: when the shares started
  bool is_h;
  int k, pb;
  unsigned strip;
  if (lag < 0) {
    // Cache-resident slab: ALL E blocks, then ALL H blocks, each half in the two-launch kernels' own order (XCD-contiguous,
    // strip-major; fd_2m is the divider nk * nbs here; odd steps walk every XCD's range backwards).  No empty positions: on
    // planes of a few dozen blocks the plane-group order below dispatches more empty blocks than real ones.
    const unsigned nE = nmain >> 1;   // = p.xgrid: eight cost-weighted XCD shares, padded to the largest
    is_h = b >= nE;
    unsigned v;
    if (!xcd_position(p.xs, is_h ? b - nE : b, down, v)) return;
    strip = fd_div(v, fd_2m);
    const unsigned rem = v - strip * fd_2m.d;
    const unsigned kk = fd_div(rem, p.fd_nbs);
    k = (int)kk; pb = (int)(rem - kk * p.fd_nbs.d);
  } else {
    const unsigned m = fd_2m.d >> 1;                    // positions per role in a plane group (the largest XCD share of a plane)
    const unsigned grp = fd_div(pos, fd_2m), w = pos - grp * fd_2m.d;
    is_h = w >= m;
    const unsigned r = is_h ? w - m : w;
    k = is_h ? (int)grp - lag : (int)grp;
    // this XCD group's contiguous share of the plane's nbp = nstrips * nbs blocks (equal in cost: the y-layer strips count more)
    const unsigned first = p.ps[x], cnt = p.ps[x + 1] - first;
    if (k < 0 || k >= p.nk || r >= cnt) return;
    // odd steps walk the planes downwards: a step starts on the planes the previous one touched last (still in the Infinity
    // Cache).  The flags an H block waits for — planes k and k + 1 — are earlier in dispatch order in either direction.
    if (down) k = p.nk - 1 - k;
    const unsigned v = first + r;
    strip = fd_div(v, p.fd_nbs);
    pb = (int)(v - strip * p.fd_nbs.d);
  }
  // P2P (mailbox halo transport, upwards only): E of plane 0 — the first blocks of the launch — takes its k-1 neighbours from
  // the lower rank's mailbox and pushes its result down; H of the top plane — the last H blocks — takes k+1 from the upper
  // rank's mailbox (that rank's E blocks of plane 0 are the first of ITS launch) and pushes its result up.
#ifdef FDTD_XCD_TRACE   // diagnostic builds only (tools/xcd_trace.py): every block leaves {start, end (its first wave), XCC_ID, E/H} in its own
  // 32-byte slot of a per-launch table (two tables, alternating with the step): plain stores, no atomics, no extra barrier
  const unsigned long long t_begin = wall_clock64();
#endif
  if (!is_h) body_E<COEF, PML, true, P2P, true, MUR, MULTI>(p, (int)strip, k, pb, step, s_lut, s_psi, s_xc, s_src, wf_target, MUR ? &mur : nullptr, ts ? wf_target - 1u : 0u);
  else body_H<COEF == 0, PML, P2P, true, MULTI, MUR>(p, (int)strip, k, pb, step, s_psi, s_xc, wf_target, ts ? wf_target - 1u : 0u, MUR ? &mh : nullptr);
  if (p.xstamp && threadIdx.x == 0) p.xstamp[b] = wall_clock64();   // ... and when this block (its first wave) was done
#ifdef FDTD_XCD_TRACE
  if (threadIdx.x == 0 && b < FDTD_XCD_TRACE_MAX) {
    unsigned xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    unsigned long long* const slot = g_xcd_trace + ((size_t)(step & 1) * FDTD_XCD_TRACE_MAX + b) * 4;
    slot[0] = t_begin; slot[1] = wall_clock64(); slot[2] = (xid & 7u) | (is_h ? 8u : 0u) | 16u; slot[3] = (unsigned long long)step;
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// K4: Mur.  One thread per face point and tangential component; blockIdx.y = face*2 + t.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FDTD_BLOCK) void k_mur(const DevParams p, const MurDev m, const int mode) {
  const int fi = blockIdx.y >> 1, t = blockIdx.y & 1;
  const MurDevFace& f = m.f[fi];
  if (!f.on) return;
  const int s = blockIdx.x * FDTD_BLOCK + threadIdx.x;
  if (s >= f.du * f.dv) return;
  const int iv = s / f.du, iu = s - iv * f.du;
  const int stride[3] = {1, p.P, p.plane};
  int pos[3];
  pos[f.a] = f.b; pos[f.ua] = iu; pos[f.va] = iv;
  const int comp = f.comp[t];
  float* V = p.V[comp];
  const int ob = pos[0] + pos[1] * p.P + pos[2] * p.plane;
  const int oi = ob + (f.in - f.b) * stride[f.a];
  const int ci = iv * f.cs + iu;   // st and cd: rows of stride cs (P; x faces: ny)
  if (mode == 0) {
    f.st[t][ci] = __builtin_fmaf(-f.coeff, V[ob], V[oi]);
  } else if (mode == 1) {
    f.cd[t][ci] = __builtin_fmaf(f.coeff, V[oi], f.st[t][ci]);
  } else {
    // faces are applied in order 0..5 by the reference order; a later face wins on shared edges
    for (int g = fi + 1; g < 6; ++g) {
      if (!m.f[g].on) continue;
      const int ga = g >> 1;
      if (ga != comp && pos[ga] == m.bnd[g]) return;
    }
    V[ob] = f.cd[t][ci];
  }
}

// ------------------------------------------------------------------------------------------------
// K3: sources + probes (+ step++ after the H half-step).  Single block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FDTD_BLOCK) void k_post(const DevParams p, const int kind, const int sources,
                                                     const long long step) {
  __shared__ double red[FDTD_BLOCK];
  if (sources) {
    for (int e = threadIdx.x; e < p.nsrc; e += FDTD_BLOCK) {
      const long long t = step - p.src_delay[e];
      if (t >= 0 && t < p.nsig) {
        float* v = p.V[p.src_comp[e]] + p.src_off[e];
        *v = *v + p.src_amp[e] * p.sig[t];
      }
    }
    __threadfence_block();
    __syncthreads();  // single block: sources land before the probes read
  }
  probe_block(p, kind, step, red);
}

// Before the first timestep of a decomposed run: the E sweep of step 0 reads, as its k-1 neighbours of plane 0, the INITIAL
// Ix, Iy of the lower rank's top plane.  Every rank pushes them up (tag 1, the slot of "step -1"), so seeded initial fields
// decompose like zero ones; a zero tag is never valid, i.e. an untouched mailbox can never be mistaken for a halo.
__global__ __launch_bounds__(FDTD_BLOCK) void k_p2p_prime(const DevParams p) {
  const unsigned t = blockIdx.x * FDTD_BLOCK + threadIdx.x;
  if (t >= (unsigned)p.plane / 4u || !p.mb_out_H) return;
  const unsigned o = 4u * t;
  const size_t top = (size_t)(p.nk - 1) * p.plane + o;
  float* mb = p.mb_out_H + (size_t)1 * 2 * mb_slot_words(p);
  mb_push(mb, o, 1u, ld4(p.I[0] + top));
  mb_push(mb + mb_slot_words(p), o, 1u, ld4(p.I[1] + top));
}

// V- and I-probes of one step in one launch, one block per probe (wavefront schedule: both fields are final when the step's
// launch has ended; the probes' cells are cold then — one block for all probes took 6 us per step at 400x400x80 with two
// probes and 20 us at 800x800x120 with eight, one memory round trip after the other)
__global__ __launch_bounds__(FDTD_BLOCK) void k_probes(const DevParams p, const long long step) {
  __shared__ double red[FDTD_BLOCK];
  if (step < 0 || step >= p.max_steps) return;
  const DevProbe pr = p.probes[blockIdx.x];
  double s = 0.0;
  for (int e = threadIdx.x; e < pr.n; e += FDTD_BLOCK) {
    const float* F = (pr.kind == FDTD_KIND_V ? p.V[pr.comp[e]] : p.I[pr.comp[e]]);
    s = fma((double)pr.w[e], (double)F[pr.off[e]], s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = FDTD_BLOCK / 2; w > 0; w >>= 1) {   // the reduction tree of probe_block: identical sums
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) pr.series[step] = red[0];
}

// ------------------------------------------------------------------------------------------------
// K6: running DFT over registered boxes.  blockIdx.y = box, grid-stride over its points.
// ------------------------------------------------------------------------------------------------
// kind < 0: the boxes of BOTH kinds in one launch (the one-launch schedules sample V and I of a timestep together)
__global__ __launch_bounds__(FDTD_BLOCK) void k_dft(const DevParams p, const int kind, const DevBox* __restrict__ boxes,
                                                    const int nfreq, const int every, const int nsamples,
                                                    const double* __restrict__ tw_v, const double* __restrict__ tw_i, const long long step) {
  if (step % every != 0) return;
  const long long smp = step / every;
  if (smp >= nsamples) return;
  const DevBox bx = boxes[blockIdx.y];
  if ((kind >= 0 && bx.kind != kind) || bx.npts == 0) return;
  const float* F = (bx.kind == FDTD_KIND_V ? p.V[bx.comp] : p.I[bx.comp]);
  const double* w = (bx.kind == FDTD_KIND_V ? tw_v : tw_i) + smp * nfreq * 2;
  for (long pt = (long)blockIdx.x * FDTD_BLOCK + threadIdx.x; pt < bx.npts; pt += (long)gridDim.x * FDTD_BLOCK) {
    const int ii = (int)(pt % bx.ni);
    const long r = pt / bx.ni;
    const int jj = (int)(r % bx.nj), kk = (int)(r / bx.nj);
    const double v = (double)F[(bx.lo[2] + kk) * p.plane + (bx.lo[1] + jj) * p.P + bx.lo[0] + ii];
    for (int f = 0; f < nfreq; ++f) {
      double* a = bx.acc + ((long)f * bx.npts + pt) * 2;
      a[0] = fma(v, w[2 * f], a[0]);
      a[1] = fma(v, w[2 * f + 1], a[1]);
    }
  }
}

// K6': time-domain recording of the same boxes (fdtd_set_recorder): sample `smp` of every box of `kind` -> rec[smp][pt].
// What the reference's engine dumps for an NF2FF box (CreateNF2FFBox, solver_fdtd_openems_fixed.py:220), kept in HBM.
__global__ __launch_bounds__(FDTD_BLOCK) void k_rec(const DevParams p, const int kind, const DevBox* __restrict__ boxes,
                                                    const int every, const int nsamples, const long long step) {
  const long long smp = step / every;
  if (step % every != 0 || smp >= nsamples) return;
  const DevBox bx = boxes[blockIdx.y];
  if ((kind >= 0 && bx.kind != kind) || bx.npts == 0) return;   // kind < 0: the boxes of both kinds
  const float* F = (bx.kind == FDTD_KIND_V ? p.V[bx.comp] : p.I[bx.comp]);
  float* dst = bx.rec + (size_t)smp * bx.npts;
  for (long pt = (long)blockIdx.x * FDTD_BLOCK + threadIdx.x; pt < bx.npts; pt += (long)gridDim.x * FDTD_BLOCK) {
    const int ii = (int)(pt % bx.ni);
    const long r = pt / bx.ni;
    const int jj = (int)(r % bx.nj), kk = (int)(r / bx.nj);
    dst[pt] = F[(bx.lo[2] + kk) * p.plane + (bx.lo[1] + jj) * p.P + bx.lo[0] + ii];
  }
}

// ... and their transform to any set of frequencies after the run (fdtd_rec_transform; what nf2ff.CalcNF2FF does with the
// dumps, fixed.py:296): out[f][pt] = sum_s rec[s][pt] * tw[s][f], samples in order, one float64 fma chain per (f, pt) —
// the same chain the running DFT builds, so both give identical bits.  NF frequencies per pass over the samples.
template <int NF>
__global__ __launch_bounds__(FDTD_BLOCK) void k_rec_dft(const float* __restrict__ rec, const long npts, const int ns, const int nfreq,
                                                        const int f0, const double* __restrict__ tw, double* __restrict__ out) {
  const long pt = (long)blockIdx.x * FDTD_BLOCK + threadIdx.x;
  if (pt >= npts) return;
  double ar[NF], ai[NF];
#pragma unroll
  for (int q = 0; q < NF; ++q) ar[q] = ai[q] = 0.0;
  for (int s = 0; s < ns; ++s) {
    const double v = (double)rec[(size_t)s * npts + pt];
    const double* w = tw + ((size_t)s * nfreq + f0) * 2;
#pragma unroll
    for (int q = 0; q < NF; ++q)
      if (f0 + q < nfreq) { ar[q] = fma(v, w[2 * q], ar[q]); ai[q] = fma(v, w[2 * q + 1], ai[q]); }
  }
#pragma unroll
  for (int q = 0; q < NF; ++q)
    if (f0 + q < nfreq) { double* o = out + ((size_t)(f0 + q) * npts + pt) * 2; o[0] = ar[q]; o[1] = ai[q]; }
}

// ------------------------------------------------------------------------------------------------
// K8: energy sums over the owned planes (pads are zero, ghosts excluded).
// ------------------------------------------------------------------------------------------------
// Every block leaves its two partial sums in part[2 * block]; the LAST block to arrive (a counter) adds them in block order:
// the same bits whatever the arrival order — the end criterion of a run is reproducible (atomicAdd on doubles is not).
__global__ __launch_bounds__(FDTD_BLOCK) void k_energy(const DevParams p, double* out, double* part, unsigned* arrived) {
  __shared__ double rv[FDTD_BLOCK], ri[FDTD_BLOCK];
  double sv = 0.0, si = 0.0;
  const int n4 = p.nloc / 4;
  for (int t = blockIdx.x * FDTD_BLOCK + threadIdx.x; t < n4; t += gridDim.x * FDTD_BLOCK) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float4 v = ld4(p.V[c] + 4 * t), i = ld4(p.I[c] + 4 * t);
      sv += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      si += (double)i.x * i.x + (double)i.y * i.y + (double)i.z * i.z + (double)i.w * i.w;
    }
  }
  rv[threadIdx.x] = sv; ri[threadIdx.x] = si;
  __syncthreads();
  for (int w = FDTD_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { rv[threadIdx.x] += rv[threadIdx.x + w]; ri[threadIdx.x] += ri[threadIdx.x + w]; }
    __syncthreads();
  }
  __shared__ unsigned last;
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = rv[0]; part[2 * blockIdx.x + 1] = ri[0];
    __threadfence();   // the partials are visible device-wide before the arrival is counted
    last = atomicAdd(arrived, 1u) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double sv2 = 0.0, si2 = 0.0;   // fixed order: thread t takes blocks t, t + 256, ...; then the tree above
  for (unsigned b = threadIdx.x; b < gridDim.x; b += FDTD_BLOCK) {
    sv2 += __hip_atomic_load(part + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    si2 += __hip_atomic_load(part + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  rv[threadIdx.x] = sv2; ri[threadIdx.x] = si2;
  __syncthreads();
  for (int w = FDTD_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { rv[threadIdx.x] += rv[threadIdx.x + w]; ri[threadIdx.x] += ri[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = rv[0]; out[1] = ri[0]; *arrived = 0u; }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void choose_tiling(fdtd_ctx* c) {
  // rows per strip: minimise idle lanes in the last block of a strip-plane, prefer ~16 rows
  const int P4 = c->p.P4, ny = c->p.ny;
  // One launch per timestep (k_step): every XCD group takes an equal share of each plane's blocks and the groups advance in
  // step (flags), so what counts there is the plane as a whole — blocks wasted on a short last strip and shares that differ
  // by a block stall everybody (400x400x80: 23-row strips = 18 strips, the last of 9 rows, 162 blocks = shares of 20 and 21:
  // 66 Gcells/s; 40-row strips = 160 blocks = 8 x 20: 70).
  const unsigned sel = c->d.flags & FDTD_FLAG_KERNEL_MASK;
  const bool wf_big = (size_t)(c->d.nk + 2) * c->plane * 6 * sizeof(float) > ((size_t)FDTD_WF_AUTO_MIB << 20);
  // (cache-resident slabs run all E blocks, then all H blocks, each half in the two-launch order: the two-launch rule fits them —
  // 300x300x60 in one launch: 75.7 Gcells/s with its 17-row strips, 72-74 with 10, 15, 20, 25, 30 or 40 rows)
  const bool wf_like = wf_big && sel != FDTD_FLAG_KERNEL_DIRECT;
  int best = 1; double best_cost = 1e30;
  for (int tys = 4; tys <= 40; ++tys) {
    const int rows = tys < ny ? tys : ny;
    const int t = rows * P4;
    const int nbs = (t + FDTD_BLOCK - 1) / FDTD_BLOCK;
    const double halo = 1.0 / tys * 0.15;   // j-neighbour rows re-read at strip seams (L2-served)
    double cost;
    if (wf_like) {
      // the plane as a whole: lanes wasted in every strip's last block AND on a short last strip (whose trailing blocks are
      // dispatched empty), a short last strip as such (measured: 300 rows in strips of 17 / 20 rows = the same 90 blocks per
      // plane, 73.5 / 74.8 Gcells/s), and — beyond the Infinity Cache, where the XCD groups advance in step — uneven shares
      const int nstrips = (ny + rows - 1) / rows;
      const double nbp = (double)nstrips * nbs;
      const double waste = 1.0 - (double)ny * P4 / (nbp * FDTD_BLOCK);
      const double ragged = (ny % rows) ? 0.02 : 0.0;
      const double uneven = wf_big ? (8.0 * std::ceil(nbp / 8.0) - nbp) / nbp : 0.0;
      cost = waste + ragged + uneven + halo;
    } else {
      cost = (double)(nbs * FDTD_BLOCK - t) / (nbs * FDTD_BLOCK) + halo;
    }
    if (cost < best_cost - 1e-12) { best_cost = cost; best = rows; }
  }
  if (const char* e = getenv("FDTD_TYS")) { const int v = atoi(e); if (v >= 1 && v <= 65536) best = v < ny ? v : ny; }   // experiments
  c->p.tys = best;
  c->p.nbs = (best * P4 + FDTD_BLOCK - 1) / FDTD_BLOCK;
  c->p.nstrips = (ny + best - 1) / best;
  c->p.fd_nbs = make_fastdiv((unsigned)c->p.nbs);
  c->p.fd_P4 = make_fastdiv((unsigned)P4);
}

// ---- cost-weighted XCD shares ------------------------------------------------------------------------------------------
// The launch's blocks in strip-major order, v = (strip * planes + kk) * nbs + pb, cut into eight contiguous ranges of equal
// COST, one per XCD (kernel_common.hpp: xcd_position).  Cost of a block = its active threads, a thread in a y-layer row
// counting 1 + w_y, in a z-layer plane 1 + w_z, in both 1 + w_y + w_z + w_yz (two more psi arrays read and written per
// layer: +16 of 37 bytes per cell and half-step; where two layers meet the second pair is a direct load after the
// differences).  Every block also pays a constant (dispatch, the LDS tables, the barrier).  The x layers touch every
// wave alike and do not enter.
static const double XCD_BLOCK_CONST = 0.06;   // of a full block's thread cost
static void strip_block_costs(const fdtd_ctx* c, std::vector<double>& act, std::vector<double>& ylay) {
  const DevParams& p = c->p;
  act.assign((size_t)p.nstrips * p.nbs, 0.0);
  ylay.assign((size_t)p.nstrips * p.nbs, 0.0);
  for (int s = 0; s < p.nstrips; ++s) {
    const int rows = std::min(p.tys, p.ny - s * p.tys);
    for (int jj = 0; jj < rows; ++jj) {
      const int j = s * p.tys + jj;
      const bool yl = c->have_cpml && (j < p.pml_lo[1] || j >= p.pml_hi[1]);
      // threads [jj * P4, (jj + 1) * P4) of the strip-plane, spread over the blocks they fall into
      int t0 = jj * p.P4;
      const int t1 = t0 + p.P4;
      while (t0 < t1) {
        const int pb = t0 / FDTD_BLOCK, upto = std::min(t1, (pb + 1) * FDTD_BLOCK);
        act[(size_t)s * p.nbs + pb] += upto - t0;
        if (yl) ylay[(size_t)s * p.nbs + pb] += upto - t0;
        t0 = upto;
      }
    }
  }
}
static void cut_shares(const std::vector<double>& cost, const double (&frac)[8], unsigned (&xs)[9], unsigned& grid) {
  const size_t n = cost.size();
  double total = 0.0;
  for (double v : cost) total += v;
  double run = 0.0, upto = 0.0;
  size_t v = 0;
  xs[0] = 0u;
  for (int x = 1; x < 8; ++x) {
    upto += frac[x - 1];
    const double target = total * upto;
    // the block that straddles the target goes to whichever side leaves the smaller error
    while (v < n && run + cost[v] <= target) run += cost[v++];
    if (v < n && target - run > run + cost[v] - target) run += cost[v++];
    xs[x] = (unsigned)v;
  }
  xs[8] = (unsigned)n;
  unsigned m = 0;
  for (int x = 0; x < 8; ++x) m = std::max(m, xs[x + 1] - xs[x]);
  grid = 8u * m;
}
static void equal_shares(unsigned n, unsigned (&xs)[9], unsigned& grid) {
  const unsigned q = n >> 3, r = n & 7u;
  for (unsigned x = 0; x <= 8; ++x) xs[x] = x * q + std::min(x, r);
  grid = 8u * (q + (r ? 1u : 0u));
}
void xcd_shares_reset(fdtd_ctx* c) { c->xcd_cache.clear(); }
// shares of a launch over planes [k_first, k_first + planes) -> c->p.xs / c->p.xgrid
static void set_xcd_shares(fdtd_ctx* c, int k_first, int planes) {
  DevParams& p = c->p;
  const auto key = std::make_pair(k_first, planes);
  auto it = c->xcd_cache.find(key);
  if (it == c->xcd_cache.end()) {
    fdtd_ctx::XcdShare sh{};
    const unsigned n = (unsigned)p.nstrips * (unsigned)planes * (unsigned)p.nbs;
    const bool full = k_first == 0 && planes == p.nk && c->xcd_adapt_done > 0;   // measured fractions: the whole-slab launches of k_step
    if (!c->xcd_balance || (!c->have_cpml && !full)) equal_shares(n, sh.xs, sh.grid);
    else {
      std::vector<double> act, ylay, cost((size_t)n);
      strip_block_costs(c, act, ylay);
      for (int s = 0; s < p.nstrips; ++s)
        for (int kk = 0; kk < planes; ++kk) {
          const int k = k_first + kk;
          const bool zl = k < p.pml_lo[2] || k >= p.pml_hi[2];
          for (int pb = 0; pb < p.nbs; ++pb) {
            const double a = act[(size_t)s * p.nbs + pb], y = ylay[(size_t)s * p.nbs + pb];
            cost[((size_t)s * planes + kk) * p.nbs + pb] =
                XCD_BLOCK_CONST * FDTD_BLOCK + a + c->xw_y * y + (zl ? c->xw_z * a + c->xw_yz * y : 0.0);
          }
        }
      static const double eighth[8] = {0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125};
      cut_shares(cost, full ? c->xfrac : eighth, sh.xs, sh.grid);
    }
    it = c->xcd_cache.emplace(key, sh).first;
  }
  for (int x = 0; x <= 8; ++x) p.xs[x] = it->second.xs[x];
  p.xgrid = it->second.grid;
}
// shares of the blocks of ONE plane (k_step with H a few planes behind E: every XCD group takes its share of each plane) -> c->p.ps / pm
static void set_plane_shares(fdtd_ctx* c) {
  DevParams& p = c->p;
  const unsigned nbp = (unsigned)p.nstrips * (unsigned)p.nbs;
  unsigned grid;
  if (!c->xcd_balance || !c->have_cpml) equal_shares(nbp, p.ps, grid);
  else {
    std::vector<double> act, ylay, cost((size_t)nbp);
    strip_block_costs(c, act, ylay);
    for (size_t q = 0; q < cost.size(); ++q) cost[q] = XCD_BLOCK_CONST * FDTD_BLOCK + act[q] + c->xw_y * ylay[q];
    cut_shares(cost, c->pfrac, p.ps, grid);
  }
  p.pm = grid / 8u;
}

// ---- measured XCD shares -------------------------------------------------------------------------------------------------
// The cost model above knows the bytes a block moves, not the chip it runs on: traced per XCD (tools/xcd_trace.py), the eight
// shares of a north-star launch finish up to 5 us apart, in a pattern that differs from box to box and has nothing to do with
// the CPML layers.  So the last k_step launch of an fdtd_run call leaves every block's end time (one 8-byte store per block,
// no atomics), and when the call has drained anyway the host takes each share's finish time T_x (its last block, measured
// from the launch's first block) and moves its cost fraction towards what would have made it finish with the others:
//   f_x <- f_x * (mean(T) / T_x)^g, renormalised, clamped to [1/16, 1/4];  g = 0.75 for the first two corrections, 0.4 after
// (one launch is a noisy sample: +-1.5 % per share on the north-star grid, +-5 % on 200x200x40).
// Measured (profiles/r03/xcd_adaptive_shares_ab.txt): first sample of a north-star context 0.92 .. 1.06 of the mean, after
// three corrections 0.99 .. 1.01; NS 72.2 -> 70.7 us per timestep, 200x200x40 27.5 -> 26.7.  Only for the order "all E
// blocks, then all H blocks": in plane groups the XCD groups advance in step (flags) and finish within the noise anyway.
// Results do not depend on the shares (a block computes the same cells wherever it runs); only the time does.
int xcd_stamp_arm(fdtd_ctx* c, hipStream_t s) {
  if (!c->xcd_adapt || !c->xcd_balance || c->p.p2p || wf_lag_for(c) < c->p.nk) return FDTD_OK;
  // room for the largest launch this slab can make (either order of k_step) + the eight start stamps
  const size_t nbp = (size_t)c->p.nstrips * c->p.nbs;
  const size_t need = 2 * (nbp + 8) * (size_t)(c->p.nk + 64) + 64;   // (plane groups: 16 * m * (nk + lag), m <= nbp / 8 + 1, lag small)
  if (need > ((size_t)1 << 24)) return FDTD_OK;                       // 128 MiB of stamps: not worth it
  if (c->xstamp_cap < need) {
    hipFree(c->xstamp); c->xstamp = nullptr; c->xstamp_cap = 0;
    if (hipMalloc(&c->xstamp, need * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); return FDTD_OK; }
    c->xstamp_cap = need;
  }
  HIPCK(c, hipMemsetAsync(c->xstamp, 0, c->xstamp_cap * sizeof(unsigned long long), s));
  c->p.xstamp = c->xstamp;
  c->xstamp_grid = 0; c->xstamp_mode = 0;
  return FDTD_OK;
}

int xcd_adapt(fdtd_ctx* c) {
  if (!c->xstamp || !c->xstamp_grid || !c->xstamp_mode) return FDTD_OK;
  const unsigned n = c->xstamp_grid;
  std::vector<unsigned long long> h((size_t)n + 8);
  HIPCK(c, hipMemcpy(h.data(), c->xstamp, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  const int mode = c->xstamp_mode;
  c->xstamp_grid = 0; c->xstamp_mode = 0;
  unsigned long long t0 = ~0ull, last[8] = {};
  for (unsigned x = 0; x < 8; ++x) if (h[n + x]) t0 = std::min(t0, h[n + x]);
  for (unsigned b = 0; b < n; ++b) last[b & 7u] = std::max(last[b & 7u], h[b]);
  double T[8], mean = 0.0;
  for (int x = 0; x < 8; ++x) {
    if (!last[x] || t0 == ~0ull || last[x] <= t0) return FDTD_OK;    // a share without blocks, or no stamps: leave everything as it is
    T[x] = (double)(last[x] - t0);
    mean += T[x] / 8.0;
  }
  if (mode != 1) return FDTD_OK;
  // one launch is one sample: a share that finished more than 25 % off the mean is an event (a pre-empted XCD, a page fault),
  // not a property of the chip — drop the sample; and no single correction moves a share by more than 6 %
  for (int x = 0; x < 8; ++x) if (T[x] < 0.8 * mean || T[x] > 1.25 * mean) return FDTD_OK;
  double (&f)[8] = c->xfrac;
  const double gain = c->xcd_adapt_done < 2 ? 0.75 : 0.4;
  double sum = 0.0;
  for (int x = 0; x < 8; ++x) {
    const double corr = std::min(1.06, std::max(0.94, std::pow(mean / T[x], gain)));
    f[x] = std::min(0.25, std::max(0.0625, f[x] * corr));
    sum += f[x];
  }
  for (int x = 0; x < 8; ++x) f[x] /= sum;
  c->xcd_adapt_done++;
  xcd_shares_reset(c);
  if (getenv("FDTD_XCD_DEBUG")) {
    fprintf(stderr, "[fdtd-hip] xcd_adapt #%d (%s): finish times / mean", c->xcd_adapt_done, mode == 1 ? "E then H" : "plane groups");
    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.3f", T[x] / mean);
    fprintf(stderr, " -> fractions");
    for (int x = 0; x < 8; ++x) fprintf(stderr, " %.4f", f[x]);
    fprintf(stderr, "\n");
  }
  return FDTD_OK;
}

// What the launchers need to know about the chip, asked of the runtime once per device (round 3 hard-coded 256 CUs and 160 KiB: wrong on a
// partitioned gfx950 — CPX / DPX logical devices of 32 / 128 CUs — and one __shared__ edit away from an over-sized launch).
struct ChipInfo { int cus = 256; unsigned lds_cu = 163840u, lds_block = 65536u; bool ok = false; };
static const ChipInfo& chip_info(int device) {
  static ChipInfo info[64];
  ChipInfo& ci = info[device & 63];
  if (!ci.ok) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
      ci.cus = std::max(1, prop.multiProcessorCount);
      ci.lds_cu = (unsigned)prop.maxSharedMemoryPerMultiProcessor;
      ci.lds_block = (unsigned)prop.sharedMemPerBlock;
      if (ci.lds_cu < ci.lds_block) ci.lds_cu = ci.lds_block;
    } else (void)hipGetLastError();
    ci.ok = true;
  }
  return ci;
}
int chip_cus(int device) { return chip_info(device).cus; }
// static LDS of a kernel as compiled (hipFuncGetAttributes), cached per kernel; ~0u: the runtime would not say
static unsigned static_lds_of(const void* fn) {
  static std::map<const void*, unsigned> cache;
  auto it = cache.find(fn);
  if (it != cache.end()) return it->second;
  hipFuncAttributes at;
  unsigned v = ~0u;
  if (hipFuncGetAttributes(&at, fn) == hipSuccess) v = (unsigned)at.sharedSizeBytes; else (void)hipGetLastError();
  cache[fn] = v;
  return v;
}
// Occupancy cap without recompiling: dynamic LDS padding so that at most `cap` blocks of THIS kernel fit a CU (0 = no cap).  Sized from the
// kernel's own static LDS and the device's limits: `cap` blocks fit, cap + 1 do not, static + dynamic never exceeds what one block may
// have — where that cannot be had (or the runtime does not tell) there is no padding: a cap is a speed knob, never a reason to fail.
static unsigned lds_pad(const fdtd_ctx* c, const void* fn, int cap, unsigned dyn_base) {
  if (cap <= 0) return 0;
  const unsigned st = static_lds_of(fn);
  if (st == ~0u) return 0;
  const ChipInfo& ci = chip_info(c->d.device);
  unsigned per = (ci.lds_cu / (unsigned)cap) & ~1023u;     // a block's LDS so that `cap` of them fill the CU
  if (per > ci.lds_block) per = ci.lds_block & ~1023u;
  const unsigned used = st + dyn_base;
  return per > used ? per - used : 0;
}

// Main-kernel launch with `dyn` bytes of dynamic LDS (+ the padding of an occupancy cap); in a profiled run (fdtd_run_profiled) the launch
// carries start / stop events that receive the dispatch's own begin and end timestamps.  A launch the runtime refuses is recorded in the
// context (the step loops return it as FDTD_E_DEVICE): the header promises error codes, never an abort.
template <typename K, typename... A>
static void launch_main(fdtd_ctx* c, K kern, dim3 grid, unsigned dyn, int cap, hipStream_t s, A... args) {
  const unsigned lds = dyn + lds_pad(c, reinterpret_cast<const void*>(kern), cap, dyn);
  const ChipInfo& ci = chip_info(c->d.device);
  const unsigned st = static_lds_of(reinterpret_cast<const void*>(kern));
  if (st != ~0u && st + lds > ci.lds_block) {
    if (!c->launch_failed) fdtd_fail(c, FDTD_E_UNSUPPORTED, "kernel launch needs %u bytes of LDS per workgroup, the device allows %u", st + lds, ci.lds_block);
    c->launch_failed = FDTD_E_UNSUPPORTED;
    return;
  }
  if (c->kev0) hipExtLaunchKernelGGL(kern, grid, dim3(FDTD_BLOCK), lds, s, c->kev0, c->kev1, 0, args...);
  else hipLaunchKernelGGL(kern, grid, dim3(FDTD_BLOCK), lds, s, args...);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess && !c->launch_failed) {
    fdtd_fail(c, FDTD_E_DEVICE, "kernel launch failed: %s (grid %u, %u bytes of dynamic LDS)", hipGetErrorString(e), grid.x, lds);
    c->launch_failed = FDTD_E_DEVICE;
  }
}

template <int COEF, bool PML>
static void launch_E2(fdtd_ctx* c, int k_begin, int nkr, long long step, bool fused, int extra, hipStream_t s) {
  const unsigned lut_bytes = (unsigned)(c->raw_op ? 0 : (c->p.lut_n + 1) / 2) * 16u;   // dynamic LDS: the coefficient table, whole 16-byte LDS-DMA pieces
  const unsigned pad = lut_bytes;
  const int cap = c->occ_e;
  if (c->p.p2p) {   // whole slab in one launch: [bottom plane (halo-dependent)] [planes 1.. in eight XCD shares] [probe block] (decode_block_p2p)
    const FastDiv fd_ps = make_fastdiv((unsigned)(nkr - 1) * (unsigned)c->p.nbs);
    const unsigned nb_main = (unsigned)c->p.nstrips * (unsigned)(nkr - 1) * (unsigned)c->p.nbs;
    set_xcd_shares(c, 1, nkr - 1);
    const dim3 grid(c->p.xgrid + (unsigned)(c->p.nstrips * c->p.nbs) + (unsigned)extra);
    launch_main(c, k_update_E<COEF, PML, true, true>, grid, pad, cap, s, c->p, 0, fd_ps, step, extra, nb_main);
    return;
  }
  set_xcd_shares(c, k_begin, nkr);
  const dim3 grid(c->p.xgrid + (unsigned)extra);
  const FastDiv fd_ps = make_fastdiv((unsigned)nkr * (unsigned)c->p.nbs);
  if (fused && c->mur_post_in_E) {
    launch_main(c, k_update_E_mur<COEF, PML>, grid, pad, cap, s, c->p, k_begin, fd_ps, step, extra, c->h_mur);
    return;
  }
  if (fused) launch_main(c, k_update_E<COEF, PML, true, false>, grid, pad, cap, s, c->p, k_begin, fd_ps, step, extra, 0u);
  else launch_main(c, k_update_E<COEF, PML, false, false>, grid, pad, cap, s, c->p, k_begin, fd_ps, step, 0, 0u);
}

void launch_update_E(fdtd_ctx* c, int k_begin, int k_end, long long step, bool fused, bool probe_block, hipStream_t s) {
  const int nkr = k_end - k_begin;
  if (nkr <= 0) return;
  const int extra = (fused && probe_block) ? c->nprobe : 0;   // one block per probe (those of the other kind leave at once)
  const int coef = c->raw_op ? 0 : (c->packed_op ? 2 : 1);
  if (c->have_cpml) {
    if (coef == 0) launch_E2<0, true>(c, k_begin, nkr, step, fused, extra, s);
    else if (coef == 1) launch_E2<1, true>(c, k_begin, nkr, step, fused, extra, s);
    else launch_E2<2, true>(c, k_begin, nkr, step, fused, extra, s);
  } else {
    if (coef == 0) launch_E2<0, false>(c, k_begin, nkr, step, fused, extra, s);
    else if (coef == 1) launch_E2<1, false>(c, k_begin, nkr, step, fused, extra, s);
    else launch_E2<2, false>(c, k_begin, nkr, step, fused, extra, s);
  }
}

template <bool RAW, bool PML>
static void launch_H2(fdtd_ctx* c, int k_begin, int nkr, long long step, int extra, hipStream_t s) {
  const unsigned pad = 0u;
  const int cap = c->occ_h;
  if (c->p.p2p) {
    const FastDiv fd_ps = make_fastdiv((unsigned)(nkr - 1) * (unsigned)c->p.nbs);
    const unsigned nb_main = (unsigned)c->p.nstrips * (unsigned)(nkr - 1) * (unsigned)c->p.nbs;
    set_xcd_shares(c, 0, nkr - 1);
    const dim3 grid(c->p.xgrid + (unsigned)(c->p.nstrips * c->p.nbs) + (unsigned)extra);
    launch_main(c, k_update_H<RAW, PML, true>, grid, pad, cap, s, c->p, 0, fd_ps, step, extra, nb_main, c->h_murh);
    return;
  }
  set_xcd_shares(c, k_begin, nkr);
  const dim3 grid(c->p.xgrid + (unsigned)extra);
  const FastDiv fd_ps = make_fastdiv((unsigned)nkr * (unsigned)c->p.nbs);
  if (c->p.mur_direct) launch_main(c, k_update_H<RAW, PML, false, true>, grid, pad, cap, s, c->p, k_begin, fd_ps, step, extra, 0u, c->h_murh);
  else launch_main(c, k_update_H<RAW, PML, false>, grid, pad, cap, s, c->p, k_begin, fd_ps, step, extra, 0u, c->h_murh);
}

void launch_update_H(fdtd_ctx* c, int k_begin, int k_end, long long step, bool probe_block, hipStream_t s, bool mur_pre) {
  const int nkr = k_end - k_begin;
  if (nkr <= 0) return;
  // extra blocks behind the main ones: [Mur pre pass of the next step (12 rows of mur_nbx blocks)] [probe block]; the Mur
  // blocks only ride along when the probe block does (the kernel tells them apart by their distance from the end)
  c->p.mur_nb = (mur_pre && probe_block && c->any_mur && c->d_mur) ? 12 * c->p.mur_nbx : 0;
  c->p.mur_direct = (c->mur_direct && c->p.mur_nb > 0) ? 1 : 0;   // (phase_E skipped the apply launch on the same condition)
  if (c->p.mur_direct) c->p.mur_nb = 0;                           // no apply pass: the main blocks store boundary voltages and st themselves
  const int extra = (probe_block ? c->nprobe : 0) + c->p.mur_nb;
  if (c->raw_op) {
    if (c->have_cpml) launch_H2<true, true>(c, k_begin, nkr, step, extra, s);
    else launch_H2<true, false>(c, k_begin, nkr, step, extra, s);
  } else {
    if (c->have_cpml) launch_H2<false, true>(c, k_begin, nkr, step, extra, s);
    else launch_H2<false, false>(c, k_begin, nkr, step, extra, s);
  }
}

// ---- one launch per timestep (k_step) ----------------------------------------------------------------------------------
// E runs `lag` plane groups ahead of H.  A plane group is 2 * 8 * m blocks; an H block should find its flags set when it
// starts, i.e. the E blocks it waits for should have LEFT the chip: lag = resident blocks / blocks per group + 2
// (measured with the streaming stand-in: smaller stalls the H blocks, larger gives the Infinity Cache away).
int wf_lag_for(const fdtd_ctx* c) {
  if (c->wf_lag > 0) return c->wf_lag;
  // cache-resident slab: nothing to gain from H following E closely (the L2s are far too small for the resident-block window),
  // something to lose (write-through V leaves L2; waiting H blocks occupy the chip): all E blocks first, then all H blocks —
  // what is left is one launch instead of two: no kernel boundary, H blocks start while the last E blocks drain
  // (300x300x60: 72.5 -> 74.8 Gcells/s, 200x200x40: 56.5 -> 60.3)
  if ((size_t)(c->d.nk + 2) * c->plane * 6 * sizeof(float) <= ((size_t)FDTD_WF_AUTO_MIB << 20)) return c->d.nk;
  const unsigned nbp = (unsigned)c->p.nstrips * (unsigned)c->p.nbs, m = (nbp + 7u) / 8u;
  const unsigned resident = (unsigned)chip_cus(c->d.device) * (unsigned)(c->occ_wf > 0 && c->occ_wf < FDTD_WF_MINBLOCKS ? c->occ_wf : FDTD_WF_MINBLOCKS);
  return (int)((resident + 16u * m - 1u) / (16u * m)) + 2;
}

template <int COEF, bool PML, bool P2P>
static void launch_step3(fdtd_ctx* c, long long step, int lag, hipStream_t s, int nsteps) {
  const unsigned nbp = (unsigned)c->p.nstrips * (unsigned)c->p.nbs;
  const unsigned lut_bytes = (unsigned)(c->raw_op ? 0 : (c->p.lut_n + 1) / 2) * 16u;
  const unsigned pad = lut_bytes;
  const int down = (!P2P && c->p.sweep_rev && (step & 1)) ? 1 : 0;
  if (lag >= c->p.nk) {   // all E blocks, then all H blocks (cache-resident slabs: wf_lag_for), each half in eight cost-weighted XCD shares
    set_xcd_shares(c, 0, c->p.nk);
    const unsigned nE = c->p.xgrid;
    if (c->p.xstamp) {
      if ((size_t)2u * nE + 8u > c->xstamp_cap) c->p.xstamp = nullptr;   // (the table was sized for other shares: skip this calibration)
      else { c->p.xstamp_n = 2u * nE; c->xstamp_grid = 2u * nE; c->xstamp_mode = 1; }
    }
    if (nsteps > 1) {   // several timesteps in one launch (never a P2P slab: launch_step_wf); every timestep walks forwards
      const unsigned per = 2u * nE + (((unsigned)c->nprobe + 7u) & ~7u);
      c->p.xstamp = nullptr;
      // Occupancy cap for SMALL grids.  With fewer blocks per half-step than the chip holds (7 x 256 = 1792), the blocks of the
      // NEXT half-steps and timesteps become resident too and sit polling flags beside the blocks that do the work:
      // 150x150x40 (960 blocks per half-step) 25.6 us per timestep with 7 blocks per CU, 18.3 with 4; 200x200x40 (1600) 24.8 ->
      // 20.9 with 6; 100x100x40 (480) 19.1 -> 13.8 with 2; from 1920 blocks on the cap costs
      // (profiles/r03/occupancy_cap_sweep_multi_timestep_launches.txt).  Resident slots ~ the blocks of one half-step, rounded up:
      int cap_m = c->occ_wf;
      if (c->occ_wf <= 0) {
        const double per_cu = (double)nbp * c->p.nk / (double)chip_cus(c->d.device);     // blocks of one half-step per CU
        int cap = per_cu >= 7.0 ? FDTD_WF_MINBLOCKS : std::min(6, (int)per_cu + 1);
        // (the reference's multi-patch scene — 143x129x89, four lumped ports of 1 350 source edges each: the dense source image, eight probes of as many
        //  cells — without CPML: 17.7 us per timestep (PEC) / 22.9 (MUR) with 4 or 5 blocks per CU, 19.8 / 26.9 with 6; with CPML 22.8 against 20.0.  The same
        //  grid with ONE small port: 6 is best for all three.  round 4)
        if (per_cu < 7.0 && !c->have_cpml && c->src_max_per_strip_plane > SRC_SCAN_MAX) cap = std::min(cap, 5);
        if (cap < FDTD_WF_MINBLOCKS) cap_m = cap;
      }
      if constexpr (!P2P) {
        if (c->wf_mur)
          launch_main(c, k_step<COEF, PML, false, true, true>, dim3(per * (unsigned)nsteps), pad, cap_m, s, c->p, step, -1, c->wf_epoch - (unsigned)(nsteps - 1), nbp,
                      make_fastdiv((unsigned)c->p.nk * (unsigned)c->p.nbs), 0, 2u * nE, make_fastdiv(per), c->h_mur, c->h_murh);
        else
          launch_main(c, k_step<COEF, PML, false, true>, dim3(per * (unsigned)nsteps), pad, cap_m, s, c->p, step, -1, c->wf_epoch - (unsigned)(nsteps - 1), nbp,
                      make_fastdiv((unsigned)c->p.nk * (unsigned)c->p.nbs), 0, 2u * nE, make_fastdiv(per), c->h_mur, c->h_murh);
      }
      return;
    }
    if constexpr (!P2P) {
      if (c->wf_mur) {
        launch_main(c, k_step<COEF, PML, false, false, true>, dim3(2u * nE + (unsigned)c->nprobe), pad, c->occ_wf, s, c->p, step, -1, c->wf_epoch, nbp,
                    make_fastdiv((unsigned)c->p.nk * (unsigned)c->p.nbs), down, 2u * nE, make_fastdiv(1u), c->h_mur, c->h_murh);
        c->p.xstamp = nullptr;
        return;
      }
    }
    launch_main(c, k_step<COEF, PML, P2P>, dim3(2u * nE + (unsigned)c->nprobe), pad, c->occ_wf, s, c->p, step, -1, c->wf_epoch, nbp,
                make_fastdiv((unsigned)c->p.nk * (unsigned)c->p.nbs), down, 2u * nE, make_fastdiv(1u), c->h_mur, c->h_murh);
    c->p.xstamp = nullptr;
    return;
  }
  set_plane_shares(c);
  const unsigned m = c->p.pm;   // positions per role in a plane group: the largest XCD share of a plane
  const unsigned nmain = 8u * 2u * m * (unsigned)(c->p.nk + lag);
  const dim3 grid(nmain + (unsigned)c->nprobe);
  c->p.xstamp = nullptr;   // (no calibration in this order: the XCD groups advance in step)
  launch_main(c, k_step<COEF, PML, P2P>, grid, pad, c->occ_wf, s, c->p, step, lag, c->wf_epoch, nbp, make_fastdiv(2u * m), down, nmain, make_fastdiv(1u), c->h_mur, c->h_murh);
}
template <int COEF, bool PML>
static void launch_step2(fdtd_ctx* c, long long step, int lag, hipStream_t s, int nsteps) {
  if (c->p.p2p) launch_step3<COEF, PML, true>(c, step, lag, s, 1);
  else launch_step3<COEF, PML, false>(c, step, lag, s, nsteps);
}

// Probe tables of the wavefront launch: which blocks own each probe's cells (the probe block waits for their flags), and which
// strip-planes hold I-probe cells (their H blocks store write-through and publish).  Rebuilt when a probe was added.
static int build_wf_probe_tables(fdtd_ctx* c, hipStream_t s) {
  const int nsp = c->p.nk * c->p.nstrips;
  std::vector<int> sp(nsp, 0), spV(nsp, 0), blk;
  std::vector<int2> rng(FDTD_MAX_PROBES, make_int2(0, 0));
  for (int q = 0; q < c->nprobe; ++q) {
    std::vector<int> mine;
    for (int off : c->h_prb_off[q]) {
      const int k = off / c->plane, r = off - k * c->plane, j = r / c->P, i = r - j * c->P;
      const int strip = j / c->p.tys, t = (j - strip * c->p.tys) * c->p.P4 + i / 4;
      mine.push_back((k * c->p.nstrips + strip) * c->p.nbs + t / FDTD_BLOCK);
      (c->probe[q].kind == FDTD_KIND_I ? sp : spV)[k * c->p.nstrips + strip] = 1;
    }
    std::sort(mine.begin(), mine.end());
    mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
    rng[q] = make_int2((int)blk.size(), (int)(blk.size() + mine.size()));
    blk.insert(blk.end(), mine.begin(), mine.end());
  }
  if (blk.empty()) blk.push_back(0);
  HIPCK(c, hipStreamSynchronize(s));     // nothing in flight reads the old tables
  hipFree(c->wf_prb_sp); hipFree(c->wf_prb_blk); hipFree(c->wf_prb_rng); hipFree(c->wf_prbV_sp);
  c->wf_prb_sp = nullptr; c->wf_prb_blk = nullptr; c->wf_prb_rng = nullptr; c->wf_prbV_sp = nullptr;
  HIPCK(c, hipMalloc(&c->wf_prbV_sp, spV.size() * sizeof(int)));
  HIPCK(c, hipMemcpy(c->wf_prbV_sp, spV.data(), spV.size() * sizeof(int), hipMemcpyHostToDevice));
  if (!c->wf_prb_done) {
    HIPCK(c, hipMalloc(&c->wf_prb_done, FDTD_MAX_PROBES * sizeof(unsigned)));
    HIPCK(c, hipMemset(c->wf_prb_done, 0, FDTD_MAX_PROBES * sizeof(unsigned)));
  }
  HIPCK(c, hipMalloc(&c->wf_prb_sp, sp.size() * sizeof(int)));
  HIPCK(c, hipMalloc(&c->wf_prb_blk, blk.size() * sizeof(int)));
  HIPCK(c, hipMalloc(&c->wf_prb_rng, rng.size() * sizeof(int2)));
  HIPCK(c, hipMemcpy(c->wf_prb_sp, sp.data(), sp.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCK(c, hipMemcpy(c->wf_prb_blk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCK(c, hipMemcpy(c->wf_prb_rng, rng.data(), rng.size() * sizeof(int2), hipMemcpyHostToDevice));
  c->wf_prb_dirty = false;
  return FDTD_OK;
}

// Timesteps one launch may hold: the order "all E blocks, then all H blocks" on a single slab (wf_lag_for == nk), the grid
// within 2^31 blocks; $FDTD_WF_MULTI caps it (1: one launch per timestep, as in round 2).
int wf_multi_max(const fdtd_ctx* c) {
  if (c->p.p2p || c->d.world != 1 || c->wf_multi <= 1 || wf_lag_for(c) < c->p.nk) return 1;
  const unsigned long long per = 2ull * ((unsigned long long)c->p.nstrips * c->p.nk * c->p.nbs + 64ull) + 64ull;
  const unsigned long long fit = ((1ull << 31) - 1ull) / per;
  return (int)std::max(1ull, std::min<unsigned long long>((unsigned long long)c->wf_multi, fit));
}

int launch_step_wf(fdtd_ctx* c, long long step, hipStream_t s, int nsteps) {
  const size_t nflags = (size_t)c->p.nk * c->p.nstrips * c->p.nbs;
  if (!c->wf_flags || c->wf_nflags != nflags) {   // first use (or a new tiling): flags start at 0, the epoch counts the launches
    if (c->wf_flags) hipFree(c->wf_flags);
    if (c->wf_flagsH) hipFree(c->wf_flagsH);
    c->wf_flags = nullptr; c->wf_flagsH = nullptr;
    HIPCK(c, hipMalloc(&c->wf_flags, nflags * sizeof(unsigned)));
    HIPCK(c, hipMemsetAsync(c->wf_flags, 0, nflags * sizeof(unsigned), s));
    HIPCK(c, hipMalloc(&c->wf_flagsH, nflags * sizeof(unsigned)));
    HIPCK(c, hipMemsetAsync(c->wf_flagsH, 0, nflags * sizeof(unsigned), s));
    c->wf_prb_dirty = true;
    if (!c->wf_err) { HIPCK(c, hipMalloc(&c->wf_err, 8 * sizeof(int))); HIPCK(c, hipMemsetAsync(c->wf_err, 0, 8 * sizeof(int), s)); }   // error word + the record of the wait that gave up
    c->wf_nflags = nflags;
    c->wf_epoch = 0;
  }
  if (c->wf_prb_dirty) { int r = build_wf_probe_tables(c, s); if (r) return r; }
  c->p.wf_flags = c->wf_flags; c->p.wf_err = c->wf_err;
  c->p.wf_flagsH = c->wf_flagsH; c->p.wf_prb_sp = c->wf_prb_sp; c->p.wf_prb_blk = c->wf_prb_blk; c->p.wf_prb_rng = c->wf_prb_rng;
  c->p.wf_prbV_sp = c->wf_prbV_sp; c->p.wf_prb_done = c->wf_prb_done;
  c->p.wf_limit = 200000000ull;   // 2 s of the 100 MHz wall clock
  c->p.wf_wait_bias = 0u;
  if (c->wf_fault_step >= step && c->wf_fault_step < step + nsteps) { c->p.wf_wait_bias = 1u; c->p.wf_limit = 2000ull; }   // test hook: a flag value nobody publishes, 20 us
  if (2 * (1 + c->p.P4 / FDTD_BLOCK) + 3 > 64) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "wavefront schedule: rows of more than %d cells", 30 * FDTD_BLOCK * 4);
  if (nsteps > 1 && nsteps > wf_multi_max(c)) return fdtd_fail(c, FDTD_E_ARG, "%d timesteps in one launch: at most %d here", nsteps, wf_multi_max(c));
  c->wf_epoch += (unsigned)nsteps;      // the flag value of the launch's LAST timestep
  const int lag = wf_lag_for(c);
  const int coef = c->raw_op ? 0 : (c->packed_op ? 2 : 1);
  if (c->have_cpml) {
    if (coef == 0) launch_step2<0, true>(c, step, lag, s, nsteps);
    else if (coef == 1) launch_step2<1, true>(c, step, lag, s, nsteps);
    else launch_step2<2, true>(c, step, lag, s, nsteps);
  } else {
    if (coef == 0) launch_step2<0, false>(c, step, lag, s, nsteps);
    else if (coef == 1) launch_step2<1, false>(c, step, lag, s, nsteps);
    else launch_step2<2, false>(c, step, lag, s, nsteps);
  }
  return FDTD_OK;
}

void launch_p2p_prime(fdtd_ctx* c, hipStream_t s) {
  hipLaunchKernelGGL(k_p2p_prime, dim3((unsigned)((c->plane / 4 + FDTD_BLOCK - 1) / FDTD_BLOCK)), dim3(FDTD_BLOCK), 0, s, c->p);
}

void launch_probes(fdtd_ctx* c, long long step, hipStream_t s) {
  if (c->nprobe > 0) hipLaunchKernelGGL(k_probes, dim3((unsigned)c->nprobe), dim3(FDTD_BLOCK), 0, s, c->p, step);
}

// face table of the context's Mur faces -> host copy + device copy (DevParams::mur)
int build_mur_table(fdtd_ctx* c) {
  MurDev& m = c->h_mur;
  m = MurDev{};
  const int dim[3] = {c->p.nx, c->p.ny, c->p.nk};
  int maxpts = 1;
  size_t tail_next[3];
  for (int q = 0; q < 3; ++q) tail_next[q] = (size_t)(c->p.nk + 1) * c->p.plane;   // first float behind the upper ghost plane (fdtd_create: mur_tail)
  for (int f = 0; f < 6; ++f) {
    const int a = f / 2, hi = f & 1;
    MurDevFace& d = m.f[f];
    d.on = c->mur[f].on;
    d.a = a;
    if (a == 2) d.b = hi ? c->d.nz - 1 - c->d.k0 : -c->d.k0;
    else d.b = hi ? dim[a] - 1 : 0;
    d.in = hi ? d.b - 1 : d.b + 1;
    m.bnd[f] = d.b;
    const int pa = (a + 1) % 3, qa = (a + 2) % 3;
    d.ua = pa < qa ? pa : qa;
    d.va = pa < qa ? qa : pa;
    d.du = dim[d.ua]; d.dv = dim[d.va];
    d.coeff = c->mur[f].coeff;
    d.comp[0] = pa; d.comp[1] = qa;
    d.cs = a == 0 ? dim[1] : c->p.P;
    for (int t = 0; t < 2; ++t) {   // candidates: behind the voltage array of their component (mur_load_V), 16-byte aligned pieces
      const int comp = d.comp[t];
      const size_t n = a == 0 ? (size_t)dim[2] * dim[1] : (size_t)(a == 1 ? dim[2] : dim[1]) * c->p.P;
      const size_t nr = (n + 3) & ~(size_t)3;
      d.co[t] = (int)tail_next[comp];
      d.cd[t] = c->p.V[comp] + tail_next[comp];
      d.cdn = (int)nr;
      tail_next[comp] += 2 * nr;                   // two copies of cd (one-launch schedule: alternating with the timestep)
      c->h_murh.so[f][t] = (int)tail_next[comp];   // st: same layout, right behind
      d.st[t] = c->p.V[comp] + tail_next[comp];
      tail_next[comp] += nr;
    }
    if (d.on && d.du * d.dv > maxpts) maxpts = d.du * d.dv;
  }
  for (int f = 0; f < 6; ++f) { c->h_murh.b[f] = m.f[f].on ? m.f[f].b : -1; c->h_murh.co[f][0] = m.f[f].co[0]; c->h_murh.co[f][1] = m.f[f].co[1]; c->h_murh.coeff[f] = m.f[f].coeff; c->h_murh.cdn[f] = m.f[f].cdn; }
  c->p.mur_nbx = (maxpts + FDTD_BLOCK - 1) / FDTD_BLOCK;
  c->p.mur = nullptr; c->p.mur_nb = 0;
  c->mur_pre_step = -1;
  if (!c->any_mur) return FDTD_OK;
  if (!c->d_mur) HIPCK(c, hipMalloc(&c->d_mur, sizeof(MurDev)));
  HIPCK(c, hipMemcpy(c->d_mur, &m, sizeof(MurDev), hipMemcpyHostToDevice));
  c->p.mur = c->d_mur;
  return FDTD_OK;
}

void launch_mur(fdtd_ctx* c, int mode, hipStream_t s) {
  if (!c->any_mur) return;
  const dim3 grid((unsigned)c->p.mur_nbx, 12), block(FDTD_BLOCK);
  hipLaunchKernelGGL(k_mur, grid, block, 0, s, c->p, c->h_mur, mode);
}

void launch_dft(fdtd_ctx* c, int kind, long long step, hipStream_t s) {
  if (!((c->nfreq || c->recorder) && c->nbox && (step % c->every) == 0)) return;
  const long pts = kind < 0 ? std::max(c->box_maxpts[0], c->box_maxpts[1]) : c->box_maxpts[kind];
  if (pts <= 0) return;
  unsigned gx = (unsigned)((pts + FDTD_BLOCK - 1) / FDTD_BLOCK);
  if (gx > 1024) gx = 1024;
  if (c->recorder)
    hipLaunchKernelGGL(k_rec, dim3(gx, (unsigned)c->nbox), dim3(FDTD_BLOCK), 0, s, c->p, kind, c->d_box, c->every, c->nsamples, step);
  else
    hipLaunchKernelGGL(k_dft, dim3(gx, (unsigned)c->nbox), dim3(FDTD_BLOCK), 0, s, c->p, kind, c->d_box, c->nfreq,
                       c->every, c->nsamples, c->tw_v, c->tw_i, step);
}

void launch_rec_dft(const float* rec, long npts, int ns, int nfreq, const double* d_tw, double* d_out, hipStream_t s) {
  const dim3 grid((unsigned)((npts + FDTD_BLOCK - 1) / FDTD_BLOCK)), block(FDTD_BLOCK);
  for (int f0 = 0; f0 < nfreq; f0 += 8)
    hipLaunchKernelGGL((k_rec_dft<8>), grid, block, 0, s, rec, npts, ns, nfreq, f0, d_tw, d_out);
}

void launch_post(fdtd_ctx* c, int kind, long long step, bool sources, hipStream_t s) {
  hipLaunchKernelGGL(k_post, dim3(1), dim3(FDTD_BLOCK), 0, s, c->p, kind, sources ? 1 : 0, step);
}

void launch_energy(fdtd_ctx* c, hipStream_t s) {
  // d_energy: [0..1] the two sums, [2..2 + 2 * ENERGY_BLOCKS) per-block partials, then the arrival counter (zero between launches)
  // (as many blocks as the slab has 1024-cell groups, at most ENERGY_BLOCKS: on the reference's 0.15 Mcell default scene 1024 mostly idle blocks
  // and their 1024 partial sums took 25 us of every 200-timestep check interval)
  const unsigned nblk = (unsigned)std::max<size_t>(1, std::min<size_t>(ENERGY_BLOCKS, (c->nloc / 4 + FDTD_BLOCK - 1) / FDTD_BLOCK));
  hipLaunchKernelGGL(k_energy, dim3(nblk), dim3(FDTD_BLOCK), 0, s, c->p, c->d_energy, c->d_energy + 2,
                     reinterpret_cast<unsigned*>(c->d_energy + 2 + 2 * ENERGY_BLOCKS));
}

#ifdef FDTD_XCD_TRACE
// diagnostic builds only: fetch the per-block trace table of the launch of `step` (the last two launches are kept)
extern "C" int fdtd_debug_xcd_trace(long long step, unsigned long long* out) {
  if (step < 0) {   // clear both tables
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_xcd_trace)) != hipSuccess) return -3;
    return hipMemset(q, 0, sizeof(unsigned long long) * 2 * FDTD_XCD_TRACE_MAX * 4) == hipSuccess ? 0 : -3;
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xcd_trace), (size_t)FDTD_XCD_TRACE_MAX * 4 * sizeof(unsigned long long),
                             (size_t)(step & 1) * FDTD_XCD_TRACE_MAX * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif
