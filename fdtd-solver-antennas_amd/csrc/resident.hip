// resident.hip — small grids: the whole grid RESIDENT IN REGISTERS for the length of a launch (k_resident).
//
// Why.  The reference GUI's default scenes (prepare_*_patch_fixed and its siblings: 56x55x50 ... 86x71x52 nodes, MUR or PML_8 faces,
// solver_fdtd_openems_fixed.py:173, solver_fdtd_openems_microstrip.py:134-145, gui_app.py:190) are 0.15-0.3 Mcells: 4-7 MB of fields on a
// chip with 128 MB of vector registers.  Stepped as sweeps over arrays, a Mur timestep is three dependent launches (E + Mur
// post, Mur apply, H + Mur pre) of 4-7 us each — every one at its launch / memory-latency floor, 14-16 us per timestep for 11 MB
// of algorithmic bytes (profiles/r04/mur_scene/before_three_launches_kernel_gaps.txt: the device is busy with those kernels, not idle
// between them), and several timesteps per launch behind per-block flags (k_step<.., MULTI>: the CPML twins) cannot get below a block's
// life time per half-step either (load round trip + store drain + flag hop: ~6 us; 12-20 us per timestep on these grids).
//
// Here a workgroup OWNS a tile of the grid — ZT planes x R rows x all of x, one thread = 4 x-cells of one row of one plane — and
// keeps its six field vectors AND its update coefficients in registers from the first timestep of a launch to the last.  Per
// half-step a thread needs its neighbours' fresh values along x, y, z:
//   * inside the tile: through LDS (every thread leaves its new vectors there; one barrier per half-step);
//   * across tiles: as data-tagged GRANULES {value, tag} through a device-scope exchange buffer (the mailbox protocol of the
//     multi-GPU halo transport, kernel_common.hpp, between workgroups of one launch): the producer stores its boundary rows /
//     planes write-through once and goes on, the consumer loads them and looks at the tags — ONE hop per half-step (measured 1.7 us between
//     loaded CUs, profiles/r04/resident_phase_trace.txt; MI355X_MICROARCH.md, handoff-1to1: 0.8-2.9 us) instead of a kernel boundary + a
//     cold memory round trip.  No flags, no atomics.
// Tiles pair the planes {0, 1} and {nk-2, nk-1} (and whole rows), so every first-order Mur update — boundary point and its inner
// neighbour — is inside ONE tile: candidates from LDS snapshots of the old and the freshly updated voltages, later face wins on
// shared edges, exactly k_mur's pre / post / apply.  CPML: the psi values of a thread's layer cells live in registers as well (res_cpml).
// Sources are applied by the threads that own their edges; the NF2FF faces are recorded by the threads that own their cells; probe cells are
// staged per timestep and reduced after the launch by probe_block's tree (identical sums).  All workgroups must be resident at
// once (they wait for each other): the launcher checks the grid against the occupancy query, every wait is bounded
// (error word -> FDTD_E_DEVICE, simulation.Simulation.run repeats the run under the two-launch schedule).
//
// ONE exchange per timestep instead of two (ghost currents updated redundantly by the neighbour tile, so that E needs nothing from outside)
// was built as well: bit-exact, and slower — 4.8 against 4.5 us on the default scene, the one hop of 2.5x the bytes costs 2.2-2.9 us and the
// redundant work sits on the critical cycle (profiles/r04/resident_one_hop_negative.txt; the kernel is in the history).
//
// Float32 operation order is that of body_E / body_H (kernels.hip) and of oracle/fdtd_oracle.c: compared bit for bit.
#include <hip/hip_ext.h>
#include <algorithm>
#include <vector>

#include "kernel_common.hpp"

namespace {

constexpr int RES_MAX_SRC = 128;     // soft-source edges one tile can hold (the reference's single-patch ports have 4-16)
constexpr int RES_MAX_PRB = 256;     // probe cells one tile can hold (a lumped port: 4 + 8-16)
#ifndef RES_POLL_SLEEP
#define RES_POLL_SLEEP 0      // s_sleep units (64 clocks) between two polling rounds of a halo that has not arrived (0 / 2 / 8 measured alike)
#endif

struct ResDev {
  int nzt, nstrips;          // z tiles, strips (a workgroup = one z tile x one strip, all of x)
  const int* kt;             // [nzt + 1] first plane of every z tile (tiles of 1 or 2 planes)
  const int* jt;             // [nstrips + 1] first row of every strip
  float* gx;                 // exchange buffer: [2 parities][blocks][4 kinds][2 comps][256 groups][8 floats] granules {value, tag}
  int* err; unsigned long long limit;
  const int2* src_rng; const int* src_ids;      // soft sources per tile (CSR)
  const int2* prb_rng; const int4* prb_cells;   // probe cells per tile (CSR): (thread, comp * 4 + element, stage slot, kind)
  float* stage; int nslots;                     // [timesteps of the launch][nslots] probe cell values
  const DevBox* boxes; int nbox, every, nsamples;   // NF2FF faces recorded in the time domain BY the kernel (nbox = 0: none / running DFT)
  int mur_on[6]; float mur_c[6];
  int nsteps; long long step0;
  unsigned tag0;             // tags of this launch: tag0 + 1 = the initial I halos, then + 2 per half-step pair
  unsigned pull_bias;        // added to the tags the pulls expect: 0, except under the fault-injection test hook ($FDTD_WF_FAULT_STEP)
};

enum { GX_KUP = 0, GX_KDOWN = 1, GX_JUP = 2, GX_JDOWN = 3 };
__device__ __forceinline__ unsigned gx_slot(const unsigned nblocks, const unsigned par, const unsigned blk, const unsigned kind, const unsigned comp) {
  return ((((par * nblocks + blk) * 4u + kind) * 2u + comp) * 256u) * 8u;   // float offset of a (kind, comp) slot
}
// one float4 of a thread as four granules
__device__ __forceinline__ void gx_push(float* gx, const unsigned slot, const unsigned idx, const unsigned tag, const float4& v) {
  const float t = __uint_as_float(tag);
  sto4_dev(gx, slot + idx * 8u, make_float4(v.x, t, v.y, t));
  sto4_dev(gx, slot + idx * 8u + 4u, make_float4(v.z, t, v.w, t));
}
__device__ __forceinline__ bool gx_load(const DevRsrc rs, const unsigned sa, const unsigned sb, const unsigned idx, const unsigned tag, float4& a, float4& b) {
  const unsigned oa = (sa + idx * 8u) << 2, ob = (sb + idx * 8u) << 2;
  const v4u_dev a0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oa, 0, 16), a1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oa + 16, 0, 16);
  const v4u_dev b0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)ob, 0, 16), b1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)ob + 16, 0, 16);
  const bool ok = a0.y == tag && a0.w == tag && a1.y == tag && a1.w == tag && b0.y == tag && b0.w == tag && b1.y == tag && b1.w == tag;
  a = make_float4(__uint_as_float(a0.x), __uint_as_float(a0.z), __uint_as_float(a1.x), __uint_as_float(a1.z));
  b = make_float4(__uint_as_float(b0.x), __uint_as_float(b0.z), __uint_as_float(b1.x), __uint_as_float(b1.z));
  return ok;
}
// Pull up to two halo pairs of this thread (k direction: ka, kb; j direction: ja, jb) — all loads of a round in flight together,
// a pair whose tags have not arrived is loaded again (bounded: `limit` ticks, then the error word; once it is set nobody spins).
// The error word and the clock are looked at every 32nd round only: each is a memory round trip of its own, and a round that finds its tags
// costs nothing but its loads.
__device__ __forceinline__ void gx_pull(const ResDev& r, const DevRsrc rs, const bool needK, const unsigned ska, const unsigned skb, const unsigned idxK,
                                        const bool needJ, const unsigned sja, const unsigned sjb, const unsigned idxJ, const unsigned tag,
                                        float4& ka, float4& kb, float4& ja, float4& jb) {
  bool pk = needK, pj = needJ;
  unsigned long long t0 = 0ull;
  for (int round = 0;; ++round) {
    if (pk) pk = !gx_load(rs, ska, skb, idxK, tag, ka, kb);
    if (pj) pj = !gx_load(rs, sja, sjb, idxJ, tag, ja, jb);
    if (__ballot(pk || pj) == 0ull) break;
    // (Polling harder or softer changes nothing: pausing between rounds, or probing one 16-byte piece per pair until its tags are there,
    // left the timestep where it was — profiles/r04/resident_phase_trace.txt.  What a waiting workgroup waits for is the hop itself.)
#if RES_POLL_SLEEP > 0
    __builtin_amdgcn_s_sleep(RES_POLL_SLEEP);
#endif
    if ((round & 31) != 31) continue;
    if (t0 == 0ull) t0 = wall_clock64();
    if (__hip_atomic_load(r.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    if ((unsigned long long)wall_clock64() - t0 > r.limit) { __hip_atomic_store(r.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
  }
}
// sample `smp` of every recorded box of `kind` that holds cells of this thread (k_rec, kernels.hip: rec[smp][pt], pt = (kk * nj + jj) * ni + ii)
__device__ __forceinline__ void res_record(const ResDev& r, const int kind, const long long smp, const int k, const int j, const int i0,
                                           const float4& f0, const float4& f1, const float4& f2) {
  for (int bq = 0; bq < r.nbox; ++bq) {
    const DevBox bx = r.boxes[bq];
    if (bx.kind != kind || bx.npts == 0) continue;
    const int kk = k - bx.lo[2], jj = j - bx.lo[1];
    if (kk < 0 || kk >= bx.nkk || jj < 0 || jj >= bx.nj) continue;
    const float4 f = bx.comp == 0 ? f0 : (bx.comp == 1 ? f1 : f2);
    float* dst = bx.rec + (size_t)smp * bx.npts + ((size_t)kk * bx.nj + jj) * bx.ni;
    const int ii = i0 - bx.lo[0];
    if (ii >= 0 && ii < bx.ni) dst[ii] = f.x;
    if (ii + 1 >= 0 && ii + 1 < bx.ni) dst[ii + 1] = f.y;
    if (ii + 2 >= 0 && ii + 2 < bx.ni) dst[ii + 2] = f.z;
    if (ii + 3 >= 0 && ii + 3 < bx.ni) dst[ii + 3] = f.w;
  }
}

// Element e (run-time) of a vector, read / replaced with bit masks: a chain of selects becomes an extractelement / insertelement with a
// variable index, which the backend serves from SCRATCH — and then keeps all six field vectors of the thread there (112 bytes per lane).
// Workgroup barrier that orders LDS traffic ONLY: __syncthreads() also drains the wave's outstanding vector-memory operations (vmcnt(0)) —
// here the write-through halo granules just published, whose acknowledgement takes microseconds and which nobody in this workgroup waits
// for (a granule validates itself by its tag; the exchange needs no store ordering).  LDS writes before it are visible to LDS reads after it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float f4_elem(const float4& v, const int e) {
  const unsigned m0 = 0u - (unsigned)(e == 0), m1 = 0u - (unsigned)(e == 1), m2 = 0u - (unsigned)(e == 2), m3 = 0u - (unsigned)(e == 3);
  return __uint_as_float((__float_as_uint(v.x) & m0) | (__float_as_uint(v.y) & m1) | (__float_as_uint(v.z) & m2) | (__float_as_uint(v.w) & m3));
}
__device__ __forceinline__ float f4_pick(const float old, const float a, const bool take) {
  const unsigned m = 0u - (unsigned)take;
  return __uint_as_float(__float_as_uint(old) ^ ((__float_as_uint(old) ^ __float_as_uint(a)) & m));
}
__device__ __forceinline__ float4 f4_with(const float4& v, const int e, const float a) {
  return make_float4(f4_pick(v.x, a, e == 0), f4_pick(v.y, a, e == 1), f4_pick(v.z, a, e == 2), f4_pick(v.w, a, e == 3));
}
// first-order Mur: S = V_in_old - c * V_b_old (pre), S += c * V_in_new (post), V_b = S (apply)  — k_mur, kernels.hip
__device__ __forceinline__ float mur_cand(const float co, const float in_new, const float b_old, const float in_old) {
  return __builtin_fmaf(co, in_new, __builtin_fmaf(-co, b_old, in_old));
}
__device__ __forceinline__ float4 mur_cand4(const float co, const float4& in_new, const float4& b_old, const float4& in_old) {
  return make_float4(mur_cand(co, in_new.x, b_old.x, in_old.x), mur_cand(co, in_new.y, b_old.y, in_old.y),
                     mur_cand(co, in_new.z, b_old.z, in_old.z), mur_cand(co, in_new.w, b_old.w, in_old.w));
}

#ifdef FDTD_RES_TRACE   // diagnostic builds only (tools/res_trace.py): per workgroup, wall-clock ticks summed over the timesteps of a launch
#define RES_TRACE_MAX 1024
__device__ unsigned long long g_res_trace[RES_TRACE_MAX * 8];   // [block][E wait, E compute, H wait, H compute, launch total, set-up, -, -]
#define RES_T(var) const unsigned long long var = wall_clock64()
#else
#define RES_T(var)
#endif

// COEF: 0 raw arrays, 1 class byte per edge, 2 one packed class byte per cell (as body_E)
// CPML inside the resident kernel: the psi values of a thread's cells — two per component and half-step, only where the cell lies in a layer —
// live in registers next to the fields (48 VGPRs); the 1-D coefficients (b, c, 1/kappa) of the thread's row / plane / four x-cells are fetched
// at the top of every half-step (read-only tables, L1 hits, in flight while the halos arrive).  Same operations as psi_stage_apply
// (kernel_common.hpp): psi <- b psi + c d;  d <- d / kappa + psi, on the difference taken along the layer's axis.
struct ResPml {
  bool inx, iny, inz;
  int ox, oy, oz;        // element offsets of the thread's psi values in the x / y / z psi arrays (psi_off_*)
};
template <bool EH>   // E-located (false) / H-located (true) coefficient tables
__device__ __forceinline__ void res_cpml(const DevParams& p, const ResPml& m, const int k, const int j, const int i0, float4 (&ps)[6],
                                         float4& dzA, float4& dzB, float4& dxA, float4& dxB, float4& dyA, float4& dyB) {
  constexpr int eh = EH ? 1 : 0;
  // ps[0], ps[1]: the z pair (psi[0][1], psi[1][0]); ps[2], ps[3]: the y pair (psi[0][0], psi[2][1]); ps[4], ps[5]: the x pair (psi[1][1], psi[2][0])
  if (m.inz) {
    const float b = p.cp[2][eh][0][k], c = p.cp[2][eh][1][k], ik = p.cp[2][eh][2][k];
    cpml_row4_reg(dzA, ps[0], b, c, ik);
    cpml_row4_reg(dzB, ps[1], b, c, ik);
  }
  if (m.iny) {
    const float b = p.cp[1][eh][0][j], c = p.cp[1][eh][1][j], ik = p.cp[1][eh][2][j];
    cpml_row4_reg(dyA, ps[2], b, c, ik);
    cpml_row4_reg(dyB, ps[3], b, c, ik);
  }
  if (m.inx) {
    const float4 b = ld4(p.cp[0][eh][0] + i0), c = ld4(p.cp[0][eh][1] + i0), ik = ld4(p.cp[0][eh][2] + i0);
    cpml_x4_apply(dxA, ps[4], b, c, ik);
    cpml_x4_apply(dxB, ps[5], b, c, ik);
  }
}
__device__ __forceinline__ void res_psi_io(const ResPml& m, float* const (&psi)[3][2], float4 (&ps)[6], const bool store) {
  float* const arr[6] = {psi[0][1], psi[1][0], psi[0][0], psi[2][1], psi[1][1], psi[2][0]};
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const bool in = q < 2 ? m.inz : (q < 4 ? m.iny : m.inx);
    const int o = q < 2 ? m.oz : (q < 4 ? m.oy : m.ox);
    if (!in) { if (!store) ps[q] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
    if (store) st4(arr[q] + o, ps[q]); else ps[q] = ld4(arr[q] + o);
  }
}

template <int COEF, bool MUR, bool PML>
__global__ __launch_bounds__(FDTD_BLOCK, 2) void k_resident(const DevParams p, const ResDev r) {
  __shared__ float4 sV[3][FDTD_BLOCK], sI[3][FDTD_BLOCK];
  __shared__ float4 sO[MUR ? 3 : 1][MUR ? FDTD_BLOCK : 1];
  __shared__ int s_sthr[RES_MAX_SRC], s_sdel[RES_MAX_SRC];
  __shared__ float s_samp[RES_MAX_SRC], s_sval[RES_MAX_SRC];
  __shared__ signed char s_ssel[RES_MAX_SRC];
  __shared__ int s_pthr[RES_MAX_PRB], s_pslot[RES_MAX_PRB];   // probe cells of this tile: V-probe cells first, then I-probe cells
  __shared__ signed char s_psel[RES_MAX_PRB];

  const unsigned b = blockIdx.x, nblocks = gridDim.x;
#ifdef FDTD_RES_TRACE
  const unsigned long long tr_begin = wall_clock64();
  unsigned long long tr_ew = 0, tr_ec = 0, tr_hw = 0, tr_hc = 0, tr_setup = 0;
#endif
  const int zt = (int)(b / (unsigned)r.nstrips), strip = (int)(b - (unsigned)zt * (unsigned)r.nstrips);
  const int k0 = r.kt[zt], ZT = r.kt[zt + 1] - k0, j0 = r.jt[strip], R = r.jt[strip + 1] - j0;
  const int P4 = p.P4, RP = R * P4, nth = ZT * RP;
  const int t = (int)threadIdx.x;
  const bool valid = t < nth;
  const int tt = valid ? t : 0;
  const int kk = tt / RP, rem = tt - kk * RP, rr = rem / P4, g = rem - rr * P4;
  const int k = k0 + kk, j = j0 + rr, i0 = 4 * g;
  const int off = k * p.plane + j * p.P + i0;
  const DevRsrc rs = dev_buf(r.gx);

  // ---- state: fields and coefficients of this thread's four cells, in registers for the whole launch -------------------------
  float4 vx = ld4(p.V[0] + off), vy = ld4(p.V[1] + off), vz = ld4(p.V[2] + off);
  float4 ix = ld4(p.I[0] + off), iy = ld4(p.I[1] + off), iz = ld4(p.I[2] + off);
  float4 ea[3], eb[3], ha[3], hb[3];
  {
    uchar4 cc = make_uchar4(0, 0, 0, 0);
    if (COEF == 2) cc = *reinterpret_cast<const uchar4*>(p.ecls + off);
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      if (COEF == 0) {
        ea[comp] = ld4(p.vv + comp * p.nloc + off); eb[comp] = ld4(p.vi + comp * p.nloc + off);
        ha[comp] = ld4(p.ii + comp * p.nloc + off); hb[comp] = ld4(p.iv + comp * p.nloc + off);
      } else {
        const float4 ex = ld4(p.emet[comp][0] + i0);
        const float m = p.emet[comp][1][j] * p.emet[comp][2][k];
        float2 l0, l1, l2, l3;
        if (COEF == 1) {
          const uchar4 c1 = *reinterpret_cast<const uchar4*>(p.ecls + comp * p.nloc + off);
          l0 = p.lut[c1.x]; l1 = p.lut[c1.y]; l2 = p.lut[c1.z]; l3 = p.lut[c1.w];
        } else {
          l0 = p.lut[3 * cc.x + comp]; l1 = p.lut[3 * cc.y + comp]; l2 = p.lut[3 * cc.z + comp]; l3 = p.lut[3 * cc.w + comp];
        }
        ea[comp] = make_float4(l0.x, l1.x, l2.x, l3.x);
        eb[comp] = make_float4(l0.y * (ex.x * m), l1.y * (ex.y * m), l2.y * (ex.z * m), l3.y * (ex.w * m));
        const float4 hx = ld4(p.hmet[comp][0] + i0);
        const float mh = p.hmet[comp][1][j] * p.hmet[comp][2][k];
        hb[comp] = make_float4(hx.x * mh, hx.y * mh, hx.z * mh, hx.w * mh);
        ha[comp] = make_float4(1.f, 1.f, 1.f, 1.f);   // (unused: ii = 1, I + t == fmaf(1, I, t) exactly)
      }
    }
  }
  ResPml pm{};
  float4 psE[6], psH[6];   // (dead without PML)
  if constexpr (PML) {
    const int ox = psi_off_x(p, k, j, i0), oy = psi_off_y(p, k, j, i0), oz = psi_off_z(p, k, j, i0);
    pm.inx = valid && ox >= 0; pm.iny = valid && oy >= 0; pm.inz = valid && oz >= 0;
    pm.ox = ox; pm.oy = oy; pm.oz = oz;
    res_psi_io(pm, p.psiE, psE, false);
    res_psi_io(pm, p.psiH, psH, false);
  }
  // soft sources of this tile -> LDS (thread, component * 4 + element, amplitude, delay), in list order
  const int2 srng = r.src_rng[b];
  const int nsrc_t = min(srng.y - srng.x, RES_MAX_SRC);
  if (t < nsrc_t) {
    const int e = r.src_ids[srng.x + t];
    const int so = p.src_off[e];
    const int sk = so / p.plane, s2 = so - sk * p.plane, sj = s2 / p.P, si = s2 - sj * p.P;
    s_sthr[t] = ((sk - k0) * R + (sj - j0)) * P4 + (si >> 2);
    s_ssel[t] = (signed char)(p.src_comp[e] * 4 + (si & 3));
    s_samp[t] = p.src_amp[e];
    s_sdel[t] = p.src_delay[e];
  }
  // probe cells of this tile -> LDS, V-probe cells first (the host lists them in that order; npv of them)
  const int2 prng = r.prb_rng[b];
  const int nprb_t = min(prng.y - prng.x, RES_MAX_PRB);
  int npv = 0;
  for (int q = t; q < nprb_t; q += FDTD_BLOCK) {
    const int4 pc = r.prb_cells[prng.x + q];
    s_pthr[q] = pc.x; s_psel[q] = (signed char)pc.y; s_pslot[q] = pc.z;
  }
  for (int q = 0; q < nprb_t; ++q) npv += r.prb_cells[prng.x + q].w == FDTD_KIND_V ? 1 : 0;   // (block-uniform, once per launch)
  const bool rec_on = r.nbox > 0;

  // where this thread's halos come from / go to
  const bool upK = valid && kk == ZT - 1 && zt < r.nzt - 1;        // my I (x, y) is the k-1 neighbour of the tile above
  const bool downK = valid && kk == 0 && zt > 0;                   // my V (x, y) is the k+1 neighbour of the tile below; I pull I from there
  const bool upJ = valid && rr == R - 1 && strip < r.nstrips - 1;  // my I (z, x) is the j-1 neighbour of the next strip
  const bool downJ = valid && rr == 0 && strip > 0;
  const unsigned idxK = (unsigned)(rr * P4 + g), idxJ = (unsigned)(kk * P4 + g);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // the I halos of "timestep step0 - 1": the fields as the launch found them
  sI[0][t] = ix; sI[1][t] = iy; sI[2][t] = iz;
  {
    const unsigned par = 0u, tag = r.tag0 + 1u;
    if (upK) { gx_push(r.gx, gx_slot(nblocks, par, b, GX_KUP, 0), idxK, tag, ix); gx_push(r.gx, gx_slot(nblocks, par, b, GX_KUP, 1), idxK, tag, iy); }
    if (upJ) { gx_push(r.gx, gx_slot(nblocks, par, b, GX_JUP, 0), idxJ, tag, iz); gx_push(r.gx, gx_slot(nblocks, par, b, GX_JUP, 1), idxJ, tag, ix); }
  }

#ifdef FDTD_RES_TRACE
  tr_setup = wall_clock64() - tr_begin;
#endif
  for (int n = 0; n < r.nsteps; ++n) {
    RES_T(tr0);
    const long long step = r.step0 + n;
    const unsigned par = (unsigned)n & 1u;
    const unsigned tagI_prev = r.tag0 + 2u * (unsigned)n + 1u;   // I of the previous timestep (or the initial fields)
    const unsigned tagV = r.tag0 + 2u * (unsigned)n + 2u;        // V of this timestep
    const unsigned tagI = r.tag0 + 2u * (unsigned)n + 3u;        // I of this timestep

    // ================= E half-step: V <- vv V + vi curl I ====================================================================
    if (t < nsrc_t) {   // this timestep's source values (stage_sources, kernel_common.hpp): amp * sig[step - delay] or 0
      const long long ts = step - s_sdel[t];
      s_sval[t] = (ts >= 0 && ts < p.nsig) ? s_samp[t] * p.sig[ts] : 0.f;
    }
    lds_barrier();   // every thread's I of the previous half-step is in sI
    {
      float4 iy_km = zero4, ix_km = zero4, iz_jm = zero4, ix_jm = zero4;
      float iz_im = 0.f, iy_im = 0.f;
      if (valid) {
        if (kk > 0) { iy_km = sI[1][t - RP]; ix_km = sI[0][t - RP]; }
        if (rr > 0) { iz_jm = sI[2][t - P4]; ix_jm = sI[0][t - P4]; }
        if (t > 0) { iz_im = sI[2][t - 1].w; iy_im = sI[1][t - 1].w; }
      }
      // (the slot parity of I of timestep s is (s + 1) & 1 counted from the launch's first timestep: the initial halos went to 0)
      gx_pull(r, rs, downK, gx_slot(nblocks, par, b - (unsigned)r.nstrips, GX_KUP, 0), gx_slot(nblocks, par, b - (unsigned)r.nstrips, GX_KUP, 1), idxK,
              downJ, gx_slot(nblocks, par, b - 1u, GX_JUP, 0), gx_slot(nblocks, par, b - 1u, GX_JUP, 1), idxJ, tagI_prev + r.pull_bias,
              ix_km, iy_km, iz_jm, ix_jm);
#ifdef FDTD_RES_TRACE
      tr_ew += wall_clock64() - tr0;
#endif
      // component x: d1 along y (of Iz), d2 along z (of Iy); y: d1 along z (Ix), d2 along x (Iz); z: d1 along x (Iy), d2 along y (Ix)
      float4 dx1 = sub4(iz, iz_jm), dx2 = sub4(iy, iy_km);
      float4 dy1 = sub4(ix, ix_km);
      float4 dy2 = make_float4(iz.x - iz_im, iz.y - iz.x, iz.z - iz.y, iz.w - iz.z);
      float4 dz1 = make_float4(iy.x - iy_im, iy.y - iy.x, iy.z - iy.y, iy.w - iy.z);
      float4 dz2 = sub4(ix, ix_jm);
      if constexpr (PML) res_cpml<false>(p, pm, k, j, i0, psE, dx2, dy1, dy2, dz1, dx1, dz2);
      if (MUR) { sO[0][t] = vx; sO[1][t] = vy; sO[2][t] = vz; }   // the voltages the Mur "pre" pass sees
      vx = upd4(ea[0], vx, eb[0], dx1, dx2);
      vy = upd4(ea[1], vy, eb[1], dy1, dy2);
      vz = upd4(ea[2], vz, eb[2], dz1, dz2);
    }
    // soft sources: V += amp * sig[step - delay] on the edges of this tile, in list order (apply_staged, kernel_common.hpp)
    for (int q = 0; q < nsrc_t; ++q) {
      if (s_sthr[q] != t) continue;
      const float a = s_sval[q];
      if (a == 0.f) continue;
      const int sel = s_ssel[q];   // (three static cases: picking the vector by a computed index sends all three through scratch)
      const int se = sel & 3;
      if ((sel >> 2) == 0) vx = f4_with(vx, se, f4_elem(vx, se) + a);
      else if ((sel >> 2) == 1) vy = f4_with(vy, se, f4_elem(vy, se) + a);
      else vz = f4_with(vz, se, f4_elem(vz, se) + a);
    }
    if (MUR) {
      // snapshots of the freshly updated voltages (what the "post" pass sees: before ANY face is applied), then the candidates of
      // every face this thread's cells lie on, faces in order 0..5: the later face wins on shared edges (k_mur's apply rule)
      sV[0][t] = vx; sV[1][t] = vy; sV[2][t] = vz;
      lds_barrier();
      if (valid) {
        if (r.mur_on[0] && g == 0) {                       // x = 0: components y, z of cell 0; inner cell 1 (same thread)
          const float co = r.mur_c[0];
          const float4 ny_ = sV[1][t], oy = sO[1][t], nz_ = sV[2][t], oz = sO[2][t];
          vy.x = mur_cand(co, ny_.y, oy.x, oy.y);
          vz.x = mur_cand(co, nz_.y, oz.x, oz.y);
        }
        if (r.mur_on[1] && g == ((p.nx - 1) >> 2)) {      // x = nx - 1: inner cell nx - 2 (this thread's, or the last of the thread before)
          const float co = r.mur_c[1];
          const int e = (p.nx - 1) & 3;
          const float4 ny_ = sV[1][t], oy = sO[1][t], nz_ = sV[2][t], oz = sO[2][t];
          float iny, ioy, inz, ioz;
          if (e > 0) { iny = f4_elem(ny_, e - 1); ioy = f4_elem(oy, e - 1); inz = f4_elem(nz_, e - 1); ioz = f4_elem(oz, e - 1); }
          else { iny = sV[1][t - 1].w; ioy = sO[1][t - 1].w; inz = sV[2][t - 1].w; ioz = sO[2][t - 1].w; }
          vy = f4_with(vy, e, mur_cand(co, iny, f4_elem(oy, e), ioy));
          vz = f4_with(vz, e, mur_cand(co, inz, f4_elem(oz, e), ioz));
        }
        if (r.mur_on[2] && j == 0) {                       // y = 0: components z, x; inner row 1
          const float co = r.mur_c[2];
          vz = mur_cand4(co, sV[2][t + P4], sO[2][t], sO[2][t + P4]);
          vx = mur_cand4(co, sV[0][t + P4], sO[0][t], sO[0][t + P4]);
        }
        if (r.mur_on[3] && j == p.ny - 1) {
          const float co = r.mur_c[3];
          vz = mur_cand4(co, sV[2][t - P4], sO[2][t], sO[2][t - P4]);
          vx = mur_cand4(co, sV[0][t - P4], sO[0][t], sO[0][t - P4]);
        }
        if (r.mur_on[4] && k == 0) {                       // z = 0: components x, y; inner plane 1 (same tile: tiles pair the end planes)
          const float co = r.mur_c[4];
          vx = mur_cand4(co, sV[0][t + RP], sO[0][t], sO[0][t + RP]);
          vy = mur_cand4(co, sV[1][t + RP], sO[1][t], sO[1][t + RP]);
        }
        if (r.mur_on[5] && k == p.nk - 1) {
          const float co = r.mur_c[5];
          vx = mur_cand4(co, sV[0][t - RP], sO[0][t], sO[0][t - RP]);
          vy = mur_cand4(co, sV[1][t - RP], sO[1][t], sO[1][t - RP]);
        }
      }
      lds_barrier();   // nobody reads the snapshots any more
    }
    sV[0][t] = vx; sV[1][t] = vy; sV[2][t] = vz;
    if (downK) { gx_push(r.gx, gx_slot(nblocks, par, b, GX_KDOWN, 0), idxK, tagV, vx); gx_push(r.gx, gx_slot(nblocks, par, b, GX_KDOWN, 1), idxK, tagV, vy); }
    if (downJ) { gx_push(r.gx, gx_slot(nblocks, par, b, GX_JDOWN, 0), idxJ, tagV, vz); gx_push(r.gx, gx_slot(nblocks, par, b, GX_JDOWN, 1), idxJ, tagV, vx); }
    for (int q = 0; q < npv; ++q) {   // V-probe cells of this tile
      if (s_pthr[q] != t) continue;
      const int sel = s_psel[q], e = sel & 3;
      r.stage[(size_t)n * r.nslots + s_pslot[q]] = (sel >> 2) == 0 ? f4_elem(vx, e) : ((sel >> 2) == 1 ? f4_elem(vy, e) : f4_elem(vz, e));
    }
    if (rec_on && step % r.every == 0 && step / r.every < r.nsamples && valid) res_record(r, FDTD_KIND_V, step / r.every, k, j, i0, vx, vy, vz);

    // ================= H half-step: I <- ii I + iv curl V ====================================================================
    RES_T(tr2);
#ifdef FDTD_RES_TRACE
    tr_ec += tr2 - tr0;
#endif
    lds_barrier();   // every thread's V of this timestep is in sV
    {
      float4 vy_kp = zero4, vx_kp = zero4, vz_jp = zero4, vx_jp = zero4;
      float vz_ip = 0.f, vy_ip = 0.f;
      if (valid) {
        if (kk < ZT - 1) { vy_kp = sV[1][t + RP]; vx_kp = sV[0][t + RP]; }
        if (rr < R - 1) { vz_jp = sV[2][t + P4]; vx_jp = sV[0][t + P4]; }
        if (t + 1 < nth) { vz_ip = sV[2][t + 1].x; vy_ip = sV[1][t + 1].x; }
      }
      gx_pull(r, rs, upK, gx_slot(nblocks, par, b + (unsigned)r.nstrips, GX_KDOWN, 0), gx_slot(nblocks, par, b + (unsigned)r.nstrips, GX_KDOWN, 1), idxK,
              upJ, gx_slot(nblocks, par, b + 1u, GX_JDOWN, 0), gx_slot(nblocks, par, b + 1u, GX_JDOWN, 1), idxJ, tagV + r.pull_bias,
              vx_kp, vy_kp, vz_jp, vx_jp);
#ifdef FDTD_RES_TRACE
      tr_hw += wall_clock64() - tr2;
#endif
      float4 dx1 = sub4(vz, vz_jp), dx2 = sub4(vy, vy_kp);
      float4 dy1 = sub4(vx, vx_kp);
      float4 dy2 = make_float4(vz.x - vz.y, vz.y - vz.z, vz.z - vz.w, vz.w - vz_ip);
      float4 dz1 = make_float4(vy.x - vy.y, vy.y - vy.z, vy.z - vy.w, vy.w - vy_ip);
      float4 dz2 = sub4(vx, vx_jp);
      if constexpr (PML) res_cpml<true>(p, pm, k, j, i0, psH, dx2, dy1, dy2, dz1, dx1, dz2);
      if (COEF == 0) {
        ix = upd4(ha[0], ix, hb[0], dx1, dx2);
        iy = upd4(ha[1], iy, hb[1], dy1, dy2);
        iz = upd4(ha[2], iz, hb[2], dz1, dz2);
      } else {   // ii = 1
        ix = make_float4(ix.x + hb[0].x * (dx1.x - dx2.x), ix.y + hb[0].y * (dx1.y - dx2.y), ix.z + hb[0].z * (dx1.z - dx2.z), ix.w + hb[0].w * (dx1.w - dx2.w));
        iy = make_float4(iy.x + hb[1].x * (dy1.x - dy2.x), iy.y + hb[1].y * (dy1.y - dy2.y), iy.z + hb[1].z * (dy1.z - dy2.z), iy.w + hb[1].w * (dy1.w - dy2.w));
        iz = make_float4(iz.x + hb[2].x * (dz1.x - dz2.x), iz.y + hb[2].y * (dz1.y - dz2.y), iz.z + hb[2].z * (dz1.z - dz2.z), iz.w + hb[2].w * (dz1.w - dz2.w));
      }
    }
    sI[0][t] = ix; sI[1][t] = iy; sI[2][t] = iz;
    {
      const unsigned parI = par ^ 1u;
      if (upK) { gx_push(r.gx, gx_slot(nblocks, parI, b, GX_KUP, 0), idxK, tagI, ix); gx_push(r.gx, gx_slot(nblocks, parI, b, GX_KUP, 1), idxK, tagI, iy); }
      if (upJ) { gx_push(r.gx, gx_slot(nblocks, parI, b, GX_JUP, 0), idxJ, tagI, iz); gx_push(r.gx, gx_slot(nblocks, parI, b, GX_JUP, 1), idxJ, tagI, ix); }
    }
    for (int q = npv; q < nprb_t; ++q) {   // I-probe cells of this tile
      if (s_pthr[q] != t) continue;
      const int sel = s_psel[q], e = sel & 3;
      r.stage[(size_t)n * r.nslots + s_pslot[q]] = (sel >> 2) == 0 ? f4_elem(ix, e) : ((sel >> 2) == 1 ? f4_elem(iy, e) : f4_elem(iz, e));
    }
    if (rec_on && step % r.every == 0 && step / r.every < r.nsamples && valid) res_record(r, FDTD_KIND_I, step / r.every, k, j, i0, ix, iy, iz);
#ifdef FDTD_RES_TRACE
    tr_hc += wall_clock64() - tr2;
#endif
  }
#ifdef FDTD_RES_TRACE
  if (t == 0 && b < RES_TRACE_MAX) {   // (E compute / H compute include their waits: subtract on the host)
    unsigned long long* q = g_res_trace + 8 * b;
    q[0] = tr_ew; q[1] = tr_ec; q[2] = tr_hw; q[3] = tr_hc; q[4] = wall_clock64() - tr_begin; q[5] = tr_setup; q[6] = (unsigned long long)r.nsteps;
  }
#endif
  if constexpr (PML) { res_psi_io(pm, p.psiE, psE, true); res_psi_io(pm, p.psiH, psH, true); }
  if (valid) {
    st4(p.V[0] + off, vx); st4(p.V[1] + off, vy); st4(p.V[2] + off, vz);
    st4(p.I[0] + off, ix); st4(p.I[1] + off, iy); st4(p.I[2] + off, iz);
  }
}

// series[step] = sum_e w[e] * cell[e] for probe blockIdx.x and timestep step0 + blockIdx.y of the launch, from the staged cell values:
// probe_block's strided partial sums and tree (kernel_common.hpp) — identical sums to every other schedule
__global__ __launch_bounds__(FDTD_BLOCK) void k_res_probes(const DevParams p, const float* __restrict__ stage, const int nslots,
                                                           const int* __restrict__ slot0, const long long step0) {
  __shared__ double red[FDTD_BLOCK];
  const long long step = step0 + blockIdx.y;
  if (step < 0 || step >= p.max_steps) return;
  const DevProbe pr = p.probes[blockIdx.x];
  const float* cells = stage + (size_t)blockIdx.y * nslots + slot0[blockIdx.x];
  double s = 0.0;
  for (int e = threadIdx.x; e < pr.n; e += FDTD_BLOCK) s = fma((double)pr.w[e], (double)cells[e], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = FDTD_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) pr.series[step] = red[0];
}

template <typename T>
hipError_t res_upload(T** dst, const std::vector<T>& v) {
  hipFree(*dst); *dst = nullptr;
  hipError_t e = hipMalloc(dst, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e == hipSuccess && !v.empty()) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

template <int COEF, bool MUR, bool PML>
const void* res_kernel() { return reinterpret_cast<const void*>(&k_resident<COEF, MUR, PML>); }
template <int COEF>
const void* res_kernel_c(bool mur, bool pml) {
  return mur ? (pml ? res_kernel<COEF, true, true>() : res_kernel<COEF, true, false>()) : (pml ? res_kernel<COEF, false, true>() : res_kernel<COEF, false, false>());
}
const void* res_kernel_of(const fdtd_ctx* c) {
  const int coef = c->raw_op ? 0 : (c->packed_op ? 2 : 1);
  return coef == 0 ? res_kernel_c<0>(c->any_mur, c->have_cpml) : coef == 1 ? res_kernel_c<1>(c->any_mur, c->have_cpml) : res_kernel_c<2>(c->any_mur, c->have_cpml);
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------
void res_free(fdtd_ctx* c) {
  ResHost& h = c->res;
  hipFree(h.d_kt); hipFree(h.d_jt); hipFree(h.gx); hipFree(h.d_src_rng); hipFree(h.d_src_ids); hipFree(h.d_prb_rng); hipFree(h.d_prb_cells);
  hipFree(h.d_slot0); hipFree(h.stage); hipFree(h.err);
  h = ResHost{};
}

// Tiling: a workgroup = ZT planes x R rows x P4 four-cell groups <= 256 threads, whole rows.  Pairs of planes whenever two rows of
// two planes fit (then {0, 1} and {nk-2, nk-1} are always pairs: an odd plane count puts its single plane second), rows spread
// evenly over the strips.  Returns false when the grid cannot be tiled (rows of more than 1024 cells).
static bool res_tiling(const fdtd_ctx* c, std::vector<int>& kt, std::vector<int>& jt) {
  const int P4 = c->p.P4, ny = c->p.ny, nk = c->p.nk;
  if (P4 > FDTD_BLOCK) return false;
  const int ZT = (4 * P4 <= FDTD_BLOCK && nk >= 4) ? 2 : 1;
  const int Rmax = std::min(ny, FDTD_BLOCK / (ZT * P4));
  const int nstrips = (ny + Rmax - 1) / Rmax;
  jt.assign(1, 0);
  for (int s = 0; s < nstrips; ++s) jt.push_back(jt.back() + ny / nstrips + (s < ny % nstrips ? 1 : 0));
  kt.assign(1, 0);
  if (ZT == 1) for (int k = 1; k <= nk; ++k) kt.push_back(k);
  else {
    kt.push_back(2);
    if (nk & 1) kt.push_back(3);
    while (kt.back() < nk) kt.push_back(kt.back() + 2);
  }
  return true;
}

// Can this context step with the grid resident in registers?  (why: a message for FDTD_E_UNSUPPORTED when it was asked for by name)
bool res_possible(fdtd_ctx* c, const char** why) {
  static const char* w_ok = "";
  const char* dummy; if (!why) why = &dummy;
  *why = w_ok;
  if (c->d.world != 1 || c->p.p2p) { *why = "single slab only"; return false; }
  if (c->d.nx < 6 || c->d.ny < 5 || c->d.nk < 5) { *why = "at least 6 x 5 x 5 nodes"; return false; }
  std::vector<int> kt, jt;
  if (!res_tiling(c, kt, jt)) { *why = "rows of at most 1024 cells"; return false; }
  const int nzt = (int)kt.size() - 1, nstrips = (int)jt.size() - 1;
  if (c->any_mur) {
    if ((c->mur[2].on && jt[1] - jt[0] < 2) || (c->mur[3].on && jt[nstrips] - jt[nstrips - 1] < 2)) { *why = "Mur y faces need two rows in the end strips"; return false; }
    if ((c->mur[4].on && kt[1] - kt[0] < 2) || (c->mur[5].on && kt[nzt] - kt[nzt - 1] < 2)) { *why = "Mur z faces need plane pairs (rows of at most 256 cells)"; return false; }
  }
  if (!c->h_src_off.empty() || c->nprobe > 0) {   // what a tile's LDS tables hold
    std::vector<int> nsrc_t((size_t)nzt * nstrips, 0), nprb_t((size_t)nzt * nstrips, 0);
    auto tile = [&](int off) {
      const int k = off / c->plane, j = (off - k * c->plane) / c->P;
      return ((int)(std::upper_bound(kt.begin(), kt.end(), k) - kt.begin()) - 1) * nstrips + (int)(std::upper_bound(jt.begin(), jt.end(), j) - jt.begin()) - 1;
    };
    for (int off : c->h_src_off) if (++nsrc_t[(size_t)tile(off)] > RES_MAX_SRC) { *why = "at most 128 source edges per tile"; return false; }
    for (int q = 0; q < c->nprobe; ++q)
      for (int off : c->h_prb_off[q]) if (++nprb_t[(size_t)tile(off)] > RES_MAX_PRB) { *why = "at most 256 probe cells per tile"; return false; }
  }
  // every workgroup must be resident at once
  const int variant = ((c->raw_op ? 0 : (c->packed_op ? 2 : 1)) * 2 + (c->any_mur ? 1 : 0)) * 2 + (c->have_cpml ? 1 : 0);
  if (c->res.capacity < 0 || c->res.capacity_variant != variant) {
    c->res.capacity_variant = variant;
    int per_cu = 0, dev = c->d.device;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, res_kernel_of(c), FDTD_BLOCK, 0) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); c->res.capacity = 0; }
    else c->res.capacity = per_cu * prop.multiProcessorCount;
  }
  if ((long)nzt * nstrips > c->res.capacity) { *why = "more tiles than the chip holds resident workgroups"; return false; }
  return true;
}

// Tables of a resident launch (tiling, sources and probe cells per tile, exchange buffer): built on first use, rebuilt when a
// source or a probe was added.
int res_prepare(fdtd_ctx* c, int max_chunk) {
  ResHost& h = c->res;
  const int nsrc = (int)c->h_src_off.size();
  if (h.built && h.nsrc_seen == nsrc && h.nprobe_seen == c->nprobe && h.chunk_cap >= max_chunk) return FDTD_OK;
  HIPCK(c, hipStreamSynchronize(c->stream));   // nothing in flight reads the old tables
  std::vector<int> kt, jt;
  if (!res_tiling(c, kt, jt)) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "resident schedule: rows of more than 1024 cells");
  h.nzt = (int)kt.size() - 1; h.nstrips = (int)jt.size() - 1; h.nblocks = h.nzt * h.nstrips;
  HIPCK(c, res_upload(&h.d_kt, kt));
  HIPCK(c, res_upload(&h.d_jt, jt));
  auto tile_of = [&](int off, int& thread, int& elem) {
    const int k = off / c->plane, r2 = off - k * c->plane, j = r2 / c->P, i = r2 - j * c->P;
    const int zt = (int)(std::upper_bound(kt.begin(), kt.end(), k) - kt.begin()) - 1;
    const int st = (int)(std::upper_bound(jt.begin(), jt.end(), j) - jt.begin()) - 1;
    const int R = jt[st + 1] - jt[st];
    thread = ((k - kt[zt]) * R + (j - jt[st])) * c->p.P4 + (i >> 2);
    elem = i & 3;
    return zt * h.nstrips + st;
  };
  // sources per tile
  {
    std::vector<std::vector<int>> lists((size_t)h.nblocks);
    for (int e = 0; e < nsrc; ++e) { int th, el; lists[(size_t)tile_of(c->h_src_off[e], th, el)].push_back(e); }
    std::vector<int2> rng((size_t)h.nblocks);
    std::vector<int> ids;
    for (int q = 0; q < h.nblocks; ++q) {
      if ((int)lists[q].size() > RES_MAX_SRC) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "resident schedule: more than %d source edges in one tile", RES_MAX_SRC);
      rng[q].x = (int)ids.size(); ids.insert(ids.end(), lists[q].begin(), lists[q].end()); rng[q].y = (int)ids.size();
    }
    HIPCK(c, res_upload(&h.d_src_rng, rng));
    HIPCK(c, res_upload(&h.d_src_ids, ids));
  }
  // probe cells per tile; stage slots in (probe, cell) order
  {
    std::vector<std::vector<int4>> lists((size_t)h.nblocks);
    std::vector<int> slot0((size_t)std::max(c->nprobe, 1), 0);
    std::vector<int8_t> comp;
    int slot = 0;
    for (int q = 0; q < c->nprobe; ++q) {
      slot0[q] = slot;
      comp.resize(c->h_prb_off[q].size());
      if (!comp.empty()) HIPCK(c, hipMemcpy(comp.data(), c->probe[q].comp, comp.size(), hipMemcpyDeviceToHost));
      for (size_t e = 0; e < c->h_prb_off[q].size(); ++e, ++slot) {
        int th, el;
        const int tile = tile_of(c->h_prb_off[q][e], th, el);
        lists[(size_t)tile].push_back(make_int4(th, comp[e] * 4 + el, slot, c->probe[q].kind));
      }
    }
    h.nslots = std::max(slot, 1);
    std::vector<int2> rng((size_t)h.nblocks);
    std::vector<int4> cells;
    for (int q = 0; q < h.nblocks; ++q) {
      if ((int)lists[q].size() > RES_MAX_PRB) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "resident schedule: more than %d probe cells in one tile", RES_MAX_PRB);
      std::stable_sort(lists[q].begin(), lists[q].end(), [](const int4& a, const int4& b2) { return a.w < b2.w; });   // V-probe cells (kind 0) first
      rng[q].x = (int)cells.size(); cells.insert(cells.end(), lists[q].begin(), lists[q].end()); rng[q].y = (int)cells.size();
    }
    HIPCK(c, res_upload(&h.d_prb_rng, rng));
    HIPCK(c, res_upload(&h.d_prb_cells, cells));
    HIPCK(c, res_upload(&h.d_slot0, slot0));
    hipFree(h.stage); h.stage = nullptr;
    HIPCK(c, hipMalloc(&h.stage, (size_t)max_chunk * h.nslots * sizeof(float)));
    HIPCK(c, hipMemset(h.stage, 0, (size_t)max_chunk * h.nslots * sizeof(float)));
    h.chunk_cap = max_chunk;
  }
  if (!h.gx) {   // granules start with tag 0 = never valid; tags only grow over the life of a context
    const size_t bytes = (size_t)2 * h.nblocks * 4 * 2 * 256 * 8 * sizeof(float);
    HIPCK(c, hipMalloc(&h.gx, bytes));
    HIPCK(c, hipMemset(h.gx, 0, bytes));
    h.tag = 0;
  }
  if (!h.err) { HIPCK(c, hipMalloc(&h.err, sizeof(int))); HIPCK(c, hipMemset(h.err, 0, sizeof(int))); }
  h.built = true; h.nsrc_seen = nsrc; h.nprobe_seen = c->nprobe;
  return FDTD_OK;
}

// `nsteps` timesteps from c->step in ONE launch (+ the probe reduction)
int launch_resident(fdtd_ctx* c, long long step, int nsteps, hipStream_t s) {
  ResHost& h = c->res;
  ResDev r{};
  r.nzt = h.nzt; r.nstrips = h.nstrips; r.kt = h.d_kt; r.jt = h.d_jt; r.gx = h.gx; r.err = h.err;
  r.limit = 200000000ull;   // 2 s of the 100 MHz wall clock
  if (c->wf_fault_step >= step && c->wf_fault_step < step + nsteps) { r.pull_bias = 0x40000000u; r.limit = 2000ull; }   // test hook: tags nobody publishes, 20 us
  r.src_rng = h.d_src_rng; r.src_ids = h.d_src_ids; r.prb_rng = h.d_prb_rng; r.prb_cells = h.d_prb_cells;
  r.stage = h.stage; r.nslots = h.nslots;
  if (c->recorder && c->nbox > 0) { r.boxes = c->d_box; r.nbox = c->nbox; r.every = c->every; r.nsamples = c->nsamples; }   // time-domain NF2FF record: inside the kernel
  for (int f = 0; f < 6; ++f) { r.mur_on[f] = c->mur[f].on; r.mur_c[f] = c->mur[f].coeff; }
  r.nsteps = nsteps; r.step0 = step;
  r.tag0 = h.tag;
  h.tag += 2u * (unsigned)nsteps + 4u;
  const dim3 grid((unsigned)h.nblocks), block(FDTD_BLOCK);
  const int coef = c->raw_op ? 0 : (c->packed_op ? 2 : 1);
#define RES_LAUNCH(CO, MU, PM)                                                                                          \
  do {                                                                                                                  \
    if (c->kev0) hipExtLaunchKernelGGL((k_resident<CO, MU, PM>), grid, block, 0, s, c->kev0, c->kev1, 0, c->p, r);      \
    else hipLaunchKernelGGL((k_resident<CO, MU, PM>), grid, block, 0, s, c->p, r);                                      \
  } while (0)
#define RES_LAUNCH_C(CO)                                                                                                \
  do {                                                                                                                  \
    if (c->any_mur) { if (c->have_cpml) RES_LAUNCH(CO, true, true); else RES_LAUNCH(CO, true, false); }                 \
    else { if (c->have_cpml) RES_LAUNCH(CO, false, true); else RES_LAUNCH(CO, false, false); }                          \
  } while (0)
  if (coef == 0) RES_LAUNCH_C(0); else if (coef == 1) RES_LAUNCH_C(1); else RES_LAUNCH_C(2);
#undef RES_LAUNCH_C
#undef RES_LAUNCH
  HIPCK(c, hipGetLastError());
  if (c->nprobe > 0)
    hipLaunchKernelGGL(k_res_probes, dim3((unsigned)c->nprobe, (unsigned)nsteps), dim3(FDTD_BLOCK), 0, s, c->p, h.stage, h.nslots, h.d_slot0, step);
  HIPCK(c, hipGetLastError());
  return FDTD_OK;
}

#ifdef FDTD_RES_TRACE
extern "C" int fdtd_debug_res_trace(unsigned long long* out) {   // diagnostic builds only: the table of the last resident launch
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_trace), sizeof(unsigned long long) * RES_TRACE_MAX * 8) == hipSuccess ? 0 : -3;
}
#endif
