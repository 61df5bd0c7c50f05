// fdtd_ctx.h — host context + device parameter block of libfdtd_hip.so (gfx950 only).
//
// HBM layout: every field component is one dense array of (nk+2) z-planes
// (one ghost plane below and above the owned slab), plane = ny rows of P floats, P = nx rounded
// up to 4 so every row starts 16-B aligned and a thread's four x-cells are one dwordx4.
// Rows follow each other without gaps, so a wave reads 1 KiB of consecutive memory whatever nx is.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fdtd_hip.h"

#define FDTD_MAX_PROBES 64
#define FDTD_MAX_BOXES 64
#define ENERGY_BLOCKS 1024      // blocks of k_energy (per-block partials, added in block order by the last block to finish)
#ifndef FDTD_BLOCK
#define FDTD_BLOCK 256
#endif
// minimum resident blocks per CU the update kernels are compiled for (register budget = occupancy target)
#ifndef FDTD_E_MINBLOCKS
#define FDTD_E_MINBLOCKS 7
#endif
#ifndef FDTD_H_MINBLOCKS
#define FDTD_H_MINBLOCKS 7
#endif
#ifndef FDTD_WF_AUTO_MIB
#define FDTD_WF_AUTO_MIB 256    // AUTO switches to one launch per timestep when the six field arrays exceed the Infinity Cache (256 MiB)
#endif
#ifndef FDTD_WF_MINBLOCKS
#define FDTD_WF_MINBLOCKS 7     // k_step (E and H bodies in one kernel)
#endif

// Division by a launch-invariant divisor as multiply-high + shift (valid for 0 <= n < 2^31): the block / thread
// decode of the update kernels otherwise spends ~90 VALU instructions per thread in three 32-bit divisions.
struct FastDiv { unsigned mul, shr, d; };
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f{0u, 0u, d};
  if (d > 1) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;              // ceil(log2 d)
    const unsigned p = 31 + l;
    f.mul = (unsigned)(((1ull << p) + d - 1) / d);
    f.shr = p - 32;
  }
  return f;
}

// First-order Mur faces as the kernels see them (k_mur, and the "pre" pass riding in the update_H launch)
struct MurDevFace {
  int on, a, b, in;      // axis, local boundary index, inner index
  int ua, va, du, dv;    // in-face axes (u fast), extents
  float coeff;
  float* st[2];          // pre pass: V_inner - coeff * V_boundary (old values); same place and layout as cd
  float* cd[2];          // post pass: st + coeff * V_inner (new values) = the boundary voltage the apply pass stores;
  int co[2], cs;         // it lives behind the voltage array of its component: cd[t] = V[comp[t]] + co[t], row stride cs (P; x faces: ny)
  int cdn;               // ... twice: the second copy cdn floats behind the first (the one-launch schedule alternates them with the timestep)
  int comp[2];
};
// what an H thread needs of the faces to load candidates instead of boundary voltages (kernels.hip mur_load_V): a KERNEL ARGUMENT, so it
// arrives with the other scalars — read through DevParams::mur it was a memory round trip in front of every wave's first field load
struct MurH { int b[6]; int co[6][2]; int so[6][2]; int cdn[6]; float coeff[6]; /* b < 0: face off; co / so: offsets of cd / st from V[comp]; cdn: to the second cd copy */ };
struct MurDev { MurDevFace f[6]; int bnd[6]; /* local boundary index per face, for the priority rule */ };

struct DevParams {
  int nx, ny, nk, P, P4;
  int plane;  // ny * P (floats)
  int nloc;   // nk * plane
  float* V[3];               // current fields (local plane 0; ghost planes at -plane and +nk*plane)
  float* I[3];
  // operator: raw arrays [3][nk*plane] or class bytes + LUT + 1-D metric tables
  const float* vv; const float* vi; const float* ii; const float* iv;
  const uint8_t* ecls;       // class mode: [3][nloc] one byte per edge; packed mode: [nloc] one byte per cell
  const float2* lut;         // class mode: [256] (vv, m); packed mode: [256][3] (vv, m) per component
  int lut_n;                 // entries actually used (only these are staged in LDS)
  const float* emet[3][3];   // [comp][axis], x tables padded to P with zeros
  const float* hmet[3][3];
  // CPML: index q along axis a is in a layer iff q < pml_lo[a] (slot q) or q >= pml_hi[a]
  // (slot q - pml_hi[a] + pml_hi_slot[a]); pml_hi[a] >= n_a disables the upper layer.
  int pml_lo[3], pml_hi[3], pml_hi_slot[3], nslot[3];
  // x-directed psi: element offset of (k, j, i0) = k * xplane + j * xrs + (i0 < pml_lo[0] ? xlo_off + i0 : xhi_off + i0 - pml_hi[0])
  int xrs, xplane, xlo_off, xhi_off;
  const float* cp[3][2][3];  // [axis][E-loc/H-loc][b, c, 1/kappa]
  const float* xc_tab;       // [E-loc/H-loc][b, c, 1/kappa][XC_MAX]: the x-layer cells' coefficients in psi-slot order (null: more than XC_MAX)
  float* psiE[3][2];
  float* psiH[3][2];
  // launch tiling: a block = 256 threads = 1024 consecutive x-cells of one strip of `tys` rows in one plane
  int tys, nbs, nstrips;
  FastDiv fd_nbs, fd_P4;     // dividers for the block / thread decode
  int sweep_rev;             // 1: update_H walks the blocks backwards (cache-friendly alternation with update_E)
  // XCD shares of THIS launch (set by the launcher): XCD x sweeps blocks [xs[x], xs[x+1]) of the strip-major order, ranges of
  // equal COST (CPML rows / planes weigh more: xcd_shares, kernels.hip); xgrid = 8 * the largest share = block positions of the
  // main part.  ps / pm: the same for the blocks of ONE plane (k_step beyond the Infinity Cache: plane groups)
  unsigned xs[9], xgrid;
  unsigned ps[9], pm;
  // calibration launches of k_step only (else null): block b leaves the wall clock at its end in xstamp[b], blocks 0..7 also
  // at their start in xstamp[xstamp_n + b] — the host turns them into per-XCD finish times and re-cuts the shares (xcd_adapt)
  unsigned long long* xstamp; unsigned xstamp_n;
  // P2P mailbox halo transport (in-kernel pushes over xGMI / peer mappings; no streams, events or RCCL in the step loop).
  // Mailbox layout (one allocation per context, zero at start): [E: 2 parities][2 comps][2 * plane] words — 8-byte granules
  // {value, tag} — then the same for H, then 64 control words.  mb_in_*: my own mailbox; mb_out_*: the neighbour's (peer
  // pointer), null without that neighbour.
  int p2p;                   // 1: the update kernels run the mailbox protocol
  int p2p_dep_first;         // 1: the halo plane's blocks come first in dispatch order ($FDTD_P2P_DEP_LAST clears)
  float* mb_in_E; float* mb_in_H;     // V x,y of the upper neighbour's plane 0 / I x,y of the lower neighbour's top plane
  float* mb_out_E; float* mb_out_H;   // lower neighbour's mb_in_E / upper neighbour's mb_in_H
  int* p2p_err;              // set when a halo wait timed out
  unsigned long long p2p_limit;   // wall-clock ticks a halo wait may last (10 s)
  unsigned p2p_tag_bias;     // added to the tag the halo waits expect: 0, except under the fault-injection test hook ($FDTD_P2P_FAULT_STEP)
  // one launch per timestep (k_step): per-block completion flags of the E blocks [nk][nstrips][nbs], error word, wait limit
  unsigned* wf_flags; int* wf_err; unsigned long long wf_limit;
  unsigned wf_wait_bias;     // added to the flag value the H blocks wait for: 0, except under the fault-injection test hook
  // ... and the probes of a step as the LAST blocks of its launch: H blocks of strip-planes that hold I-probe cells store
  // write-through and publish flags of their own (wf_flagsH, same indexing); probe q waits for the blocks wf_prb_blk[wf_prb_rng[q]]
  unsigned* wf_flagsH; const int* wf_prb_sp; const int* wf_prb_blk; const int2* wf_prb_rng;
  // several timesteps per launch: strip-planes that hold V-probe cells (wf_prb_sp: I-probe cells), and per probe the flag value
  // of the last timestep whose probe block has read its cells (the next timestep's blocks wait for it before they overwrite them)
  const int* wf_prbV_sp; unsigned* wf_prb_done;
  int nt;                    // 1: non-temporal stores for the field outputs (working set beyond the Infinity Cache)
  // fused soft sources (update_E) and probes (extra block of update_E / update_H)
  const int2* src_rng;       // [nk][nstrips]: range into src_ids of the sources inside that strip-plane
  const int* src_ids;
  int nsrc; const int* src_off; const int8_t* src_comp; const float* src_amp; const int* src_delay;
  int src_dense_ok;          // 1: no two sources on one edge — strip-planes with more than SRC_SCAN_MAX sources take the dense form (body_E)
  const float* sig; int nsig;
  const struct DevProbe* probes; int nprobe; int max_steps;
  // Mur "pre" pass of the NEXT step as extra blocks of the update_H launch (it only reads V, which update_H does not write)
  const MurDev* mur;         // device copy of the face table (null without Mur faces)
  int mur_nbx;               // blocks per (face, component) row
  int mur_nb;                // Mur blocks in THIS launch (0: none) — set by the launcher
  int mur_direct;            // THIS launch: no apply pass ran — the H blocks and the pre pass take the candidates (kernels.hip mur_applied)
};

struct DevProbe { int kind, n; const int* off; const int8_t* comp; const float* w; double* series; };
struct DevBox   { int kind, comp; int lo[3]; int ni, nj, nkk; double* acc; long npts; float* rec; /* recorder: [nsamples][npts] */ };

struct MurFace { int on; float coeff; int n; };

// host side of the resident schedule (resident.hip): tiling, per-tile source / probe-cell tables, the granule exchange buffer
struct ResHost {
  bool built = false;
  int nzt = 0, nstrips = 0, nblocks = 0;
  int* d_kt = nullptr; int* d_jt = nullptr;
  float* gx = nullptr; unsigned tag = 0;
  int2* d_src_rng = nullptr; int* d_src_ids = nullptr;
  int2* d_prb_rng = nullptr; int4* d_prb_cells = nullptr; int* d_slot0 = nullptr;
  float* stage = nullptr; int nslots = 1, chunk_cap = 0;
  int* err = nullptr;
  int nsrc_seen = -1, nprobe_seen = -1;
  int capacity = -1, capacity_variant = -1;   // workgroups of k_resident the chip holds at once (occupancy query), for which kernel variant
};

struct fdtd_ctx {
  fdtd_desc d{};
  DevParams p{};
  int P = 0, plane = 0;
  size_t nloc = 0;               // nk*plane
  float* fieldbase[6] = {};      // allocations incl. ghosts
  size_t mur_tail = 0;           // floats behind each voltage array: the Mur candidates (MurDevFace::cd)
  // P2P mailbox transport
  void* mbox = nullptr; size_t mbox_bytes = 0;     // my mailbox allocation
  bool mbox_fine = false;                          // fine-grained (system-coherent) device memory
  void* peer_lo = nullptr; void* peer_hi = nullptr; // neighbours' mailboxes (IPC mappings or in-process pointers)
  bool peer_lo_ipc = false, peer_hi_ipc = false;
  int32_t link_info[2][8] = {{-1, -1, -1, -1, -1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1, -1, -1}};   // fdtd_p2p_link_info, [lower / upper]
  bool p2p_primed = false;       // the initial top-plane Ix, Iy have been pushed to the upper rank (k_p2p_prime)
  // one launch per timestep (wavefront schedule, k_step)
  int wf_mode = -1;              // -1 auto (grids beyond the Infinity Cache), 0 off, 1 on; $FDTD_WAVEFRONT
  int wf_lag = 0;                // planes the E sweep runs ahead (0: auto); $FDTD_WF_LAG
  unsigned* wf_flags = nullptr; size_t wf_nflags = 0; int* wf_err = nullptr;
  unsigned wf_epoch = 0;         // flag value of the last wavefront launch
  long long p2p_fault_step = -1; // test hook ($FDTD_P2P_FAULT_STEP): a run that covers that step waits for halo tags nobody sends (bounded wait -> error)
  long long wf_fault_step = -1;  // test hook ($FDTD_WF_FAULT_STEP): at that step the H blocks wait for a flag value nobody publishes
  unsigned* wf_flagsH = nullptr; int* wf_prb_sp = nullptr; int* wf_prb_blk = nullptr; int2* wf_prb_rng = nullptr;
  int* wf_prbV_sp = nullptr; unsigned* wf_prb_done = nullptr;
  int res_mode = -1;             // grid resident in registers (k_resident): -1 auto, 0 never, 1 whenever possible; $FDTD_RESIDENT
  int res_chunk = 256;           // timesteps per resident launch at most; $FDTD_RES_CHUNK
  ResHost res;
  int wf_multi = 64;             // timesteps per launch at most (cache-resident single slabs without Mur faces); 1 = one launch per timestep; $FDTD_WF_MULTI
  bool wf_prb_dirty = true;      // probe tables of the wavefront launch need rebuilding (a probe was added)
  std::vector<int> h_prb_off[FDTD_MAX_PROBES];   // local offsets of every probe's cells (host copy)
  int occ_wf = 0;                // cap on resident blocks per CU of k_step (0: none); $FDTD_OCC_WF
  int occ_e = 0, occ_h = 0;      // cap on resident blocks per CU of update_E / update_H (0: none); $FDTD_OCC_E / $FDTD_OCC_H
  // cost-weighted XCD shares (xcd_shares): relative extra cost of a thread in a y-layer row / in a z-layer plane / in both,
  // over a thread outside the layers; $FDTD_XCD_WY / _WZ / _WYZ, $FDTD_XCD_BALANCE=0 for equal lengths
  bool xcd_balance = true;
  double xw_y = 0.45, xw_z = 0.45, xw_yz = 0.35;
  struct XcdShare { unsigned xs[9]; unsigned grid; };
  // ... and MEASURED: the eight shares are cut at cumulative cost fractions xfrac (k_step, all E blocks then all H blocks; pfrac:
  // k_step in plane groups, never adapted), 1/8 each to begin with; after a run the finish times of the eight XCD shares in its
  // last launch move them (api.hip: xcd_adapt) — the XCDs of one chip do not run equally fast, and not the same ones
  // are slow on every chip (profiles/r03/xcd_trace*.txt).  $FDTD_XCD_ADAPT=0 keeps the model's cuts.
  bool xcd_adapt = true;
  double xfrac[8] = {0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125};
  double pfrac[8] = {0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125};
  int xcd_adapt_calls = 0, xcd_adapt_done = 0;   // fdtd_run calls seen / adaptations made
  unsigned long long* xstamp = nullptr; size_t xstamp_cap = 0;   // device table of the calibration launch
  unsigned xstamp_grid = 0; int xstamp_mode = 0;   // main blocks of the stamped launch; 1: all E then all H, 2: plane groups
  std::map<std::pair<int, int>, XcdShare> xcd_cache;   // (first plane, planes) of a launch -> its shares
  float *vv = nullptr, *vi = nullptr, *ii = nullptr, *iv = nullptr;
  uint8_t* ecls = nullptr;
  float2* lut = nullptr;
  float* met = nullptr;          // packed metric tables
  bool have_op = false, raw_op = false, packed_op = false;
  int op_nclasses = 0;           // distinct (vv, m) pairs of the class operator (0: raw)
  int2* src_rng = nullptr; int* src_ids = nullptr;
  // cpml
  bool have_cpml = false;
  float* cpcoef = nullptr;
  float* xc_tab = nullptr;       // compact x-layer coefficient table (DevParams::xc_tab)
  float* psi[12] = {};
  // mur
  MurFace mur[6] = {};
  bool any_mur = false;
  MurH h_murh{};
  MurDev h_mur{}; MurDev* d_mur = nullptr;   // face table (built by fdtd_set_mur), host and device copy
  bool mur_fuse_post = true;                 // allow it ($FDTD_MUR_UNFUSED clears)
  bool mur_post_in_E = false;                // this launch of update_E runs the Mur post pass as well (set by phase_E)
  bool mur_no_apply = true;                  // allow the schedule without an apply pass ($FDTD_MUR_APPLY_PASS=1 clears): api.hip mur_direct_possible
  bool mur_direct = false;                   // this timestep: update_H reads the candidates itself and stores them (no k_mur apply launch)
  bool wf_mur = false;                       // this one-launch run carries Mur faces (k_step<..., MUR>; api.hip step_loop_wf)
  int64_t mur_pre_step = -1;                 // step whose Mur pre pass has already run (inside the previous update_H launch)
  // excitation
  float* sig = nullptr; int nsig = 0;
  int nsrc = 0; int* src_off = nullptr; int8_t* src_comp = nullptr; float* src_amp = nullptr; int* src_delay = nullptr;
  int src_max_per_strip_plane = 0;   // most source edges in one strip-plane of the current tiling (scanning path: at most FDTD_BLOCK)
  std::vector<int> h_src_off; std::vector<int8_t> h_src_comp; std::vector<float> h_src_amp; std::vector<int> h_src_delay;
  // probes
  int nprobe = 0; DevProbe probe[FDTD_MAX_PROBES] = {}; DevProbe* d_probe = nullptr;
  // dft
  int nfreq = 0, every = 0, nsamples = 0; double *tw_v = nullptr, *tw_i = nullptr;
  bool recorder = false;         // boxes keep time-domain samples (fdtd_set_recorder) instead of running-DFT sums
  int nbox = 0; DevBox box[FDTD_MAX_BOXES] = {}; DevBox* d_box = nullptr;
  int32_t box_lo[FDTD_MAX_BOXES][3] = {}, box_hi[FDTD_MAX_BOXES][3] = {};
  long box_maxpts[2] = {0, 0};
  // fdtd_rec_transform: device scratch kept between calls (two twiddle tables recognised by content, one output buffer)
  double* rt_tw[2] = {nullptr, nullptr}; std::vector<double> rt_tw_host[2]; int rt_next = 0;
  double* rt_out = nullptr; size_t rt_out_bytes = 0;
  // stepping
  int64_t step = 0;
  double* d_energy = nullptr;
  hipStream_t stream = nullptr, comm_stream = nullptr;   // (api.hip stream_take / stream_shared: single-slab contexts share one per device)
  bool stream_shared = false;
  hipEvent_t ev_E = nullptr, ev_H = nullptr, ev_haloE = nullptr, ev_haloH = nullptr;
  bool haloE_pending = false, haloH_pending = false;
  void* comm = nullptr;          // ncclComm_t
  int rccl_inline_mode = -1;     // RCCL exchange in stream order on the compute stream: -1 auto (short sweeps), 0 never, 1 always; $FDTD_RCCL_INLINE
  fdtd_ctx* link_lo = nullptr;   // in-process neighbours (fdtd_link)
  fdtd_ctx* link_hi = nullptr;
  bool haloE_issued = false, haloH_issued = false;
  bool tables_dirty = true;
  int launch_failed = 0;         // a main-kernel launch the runtime refused (launch_main): the step loop returns this code, message in err
  hipEvent_t kev0 = nullptr, kev1 = nullptr;   // profiled run: start / stop events the next main launch carries
  std::string err;
};

int fdtd_fail(fdtd_ctx* c, int code, const char* fmt, ...);
#define HIPCK(c, expr)                                                                       \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fdtd_fail(c, FDTD_E_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// kernels.hip
// `fused`: sources injected inside update_E; `probe_block`: one extra block samples the probes
// (update_E: I-probes of step-1, update_H: V-probes of step) so a step is exactly two launches.
int upload_metric_tables(fdtd_ctx* c, const float* emet, const float* hmet);
void launch_update_E(fdtd_ctx* c, int k_begin, int k_end, long long step, bool fused, bool probe_block, hipStream_t s);
void launch_update_H(fdtd_ctx* c, int k_begin, int k_end, long long step, bool probe_block, hipStream_t s, bool mur_pre = false);
// one launch = E and H half-step of all planes (single slab, no Mur, fusable sources); probes are sampled by launch_probes
int launch_step_wf(fdtd_ctx* c, long long step, hipStream_t s, int nsteps = 1);   // nsteps > 1: that many timesteps in ONE launch
int wf_multi_max(const fdtd_ctx* c);   // timesteps one launch may hold (1: this context steps one timestep per launch)
int wf_lag_for(const fdtd_ctx* c);
void launch_p2p_prime(fdtd_ctx* c, hipStream_t s);   // p2p transport, before step 0: initial Ix, Iy of the top plane -> upper rank's mailbox
void launch_probes(fdtd_ctx* c, long long step, hipStream_t s);   // V- and I-probes of `step` in one launch (stand-alone form)
int build_mur_table(fdtd_ctx* c);   // after fdtd_set_mur: face table -> device
void launch_mur(fdtd_ctx* c, int mode, hipStream_t s);
void launch_post(fdtd_ctx* c, int kind, long long step, bool sources, hipStream_t s);   // stand-alone sources + probes
void launch_dft(fdtd_ctx* c, int kind, long long step, hipStream_t s);   // running DFT or time-domain recording of the boxes (kind < 0: both kinds)
void launch_rec_dft(const float* rec, long npts, int ns, int nfreq, const double* d_tw, double* d_out, hipStream_t s);
void launch_energy(fdtd_ctx* c, hipStream_t s);
void choose_tiling(fdtd_ctx* c);
int chip_cus(int device);        // compute units of the (logical) device, from the runtime
// resident.hip: small grids resident in registers for the length of a launch
bool res_possible(fdtd_ctx* c, const char** why);
int res_prepare(fdtd_ctx* c, int max_chunk);
int launch_resident(fdtd_ctx* c, long long step, int nsteps, hipStream_t s);
void res_free(fdtd_ctx* c);
void xcd_shares_reset(fdtd_ctx* c);   // after the CPML layers or the tiling changed
int xcd_stamp_arm(fdtd_ctx* c, hipStream_t s);   // the next k_step launch leaves its blocks' end times (calibration)
int xcd_adapt(fdtd_ctx* c);              // after that launch has finished: per-XCD finish times -> new share fractions
