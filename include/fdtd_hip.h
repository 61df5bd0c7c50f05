/*
 * fdtd_hip.h — C ABI of libfdtd_hip.so, the MI355X (gfx950) EC-FDTD engine.
 *
 * This is the drop-in boundary for the one hot path of Veeryan/FDTD-solver-antennas:
 * everything the reference delegates to the external openEMS engine behind
 *
 *     FDTD.Run(sim_path, verbose, cleanup)            antenna_sim/solver_fdtd_openems_fixed.py:280
 *                                                     antenna_sim/solver_fdtd_openems_microstrip.py:401
 *                                                     antenna_sim/solver_fdtd_openems_microstrip_3d.py:214
 *                                                     antenna_sim/solver_fdtd_openems_microstrip_multi_3d.py:610
 *                                                     antenna_sim/solver_fdtd_openems.py:289
 *     nf2ff.CalcNF2FF(sim_path, f, theta, phi, center) antenna_sim/solver_fdtd_openems_fixed.py:296 (and :433/:225/:621/:301)
 *     port.CalcPort(sim_path, f)                      antenna_sim/solver_fdtd_openems_microstrip.py:409 (dead upstream)
 *
 * The reference has no FFI of its own (it imports the openEMS Python module); the entry
 * points below are what a ctypes binding for this path binds instead.
 * Plain pointers and sizes only; no C++/torch types.  All functions return 0 on success
 * and a negative FDTD_E_* code on failure (never throw, never abort); the message is
 * available from fdtd_last_error().  A context is used by one host thread at a time.
 *
 * The same ABI is exported by oracle/libfdtd_oracle.so (plain-C CPU restatement, test
 * infrastructure only) so every parity test drives both libraries through one wrapper.
 *
 * Conventions
 * -----------
 *  - Grid: nx*ny*nz mesh NODES (openEMS "numLines"); node (i,j,k) owns the three edges
 *    leaving it in +x,+y,+z (voltages V = E*len) and the three dual faces (currents I = H*len~).
 *    "cells" in the Mcells/s metric = nx*ny*nz, like openEMS's "MC/s".
 *  - Host arrays are dense, x fastest: a[c][k][j][i] -> ((c*nk + k)*ny + j)*nx + i, c in {x,y,z}.
 *  - z-slab decomposition: a context owns planes [k0, k0+nk) of the global grid; `nk`-sized
 *    array arguments are LOCAL to the slab, index arguments (sources, probes, boxes) are GLOBAL
 *    and are clipped to the slab by the library.
 *  - Leapfrog step n (n = 0,1,...), identical operation order in the HIP and oracle builds:
 *       1. Mur pre-pass            (boundary faces with Mur enabled)
 *       2. V <- vv*V + vi*(curl I [CPML-stretched] )                       "E half-step"
 *       3. Mur post+apply; V[e] += amp[e]*signal[n-delay[e]]; V-probes sample; V-DFT boxes accumulate
 *       4. I <- ii*I + iv*(curl V [CPML-stretched] )                       "H half-step"
 *       5. I-probes sample; I-DFT boxes accumulate; n <- n+1
 *    with (x: cyclic y,z)
 *       curlI_x = Iz(i,j,k) - Iz(i,j-1,k) - Iy(i,j,k) + Iy(i,j,k-1)
 *       curlV_x = Vz(i,j,k) - Vz(i,j+1,k) - Vy(i,j,k) + Vy(i,j,k+1)
 *    Out-of-range neighbours are only ever multiplied by zero coefficients.
 */
#ifndef FDTD_HIP_H
#define FDTD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: fdtd_p2p_link_info, fdtd_schedule_info; FDTD_FLAG_NO_GRAPH (reserved, unused) removed; fdtd_profile.ms_update_e is per timestep */
/* 4: FDTD_FLAG_KERNEL_RESIDENT (+ fdtd_schedule_info info[1] = -1 for it); no entry point added or changed */
#define FDTD_ABI_VERSION 4

enum {
  FDTD_OK = 0,
  FDTD_E_ARG = -1,      /* bad argument / shape mismatch */
  FDTD_E_STATE = -2,    /* call order (e.g. run before operator set) */
  FDTD_E_DEVICE = -3,   /* HIP / RCCL runtime error */
  FDTD_E_NOMEM = -4,
  FDTD_E_UNSUPPORTED = -5
};

enum { FDTD_KIND_V = 0, FDTD_KIND_I = 1 };           /* edge voltages (E) / face currents (H) */
enum { FDTD_PHASE_E = 0, FDTD_PHASE_H = 1 };
enum { FDTD_HALO_H_UP = 0, FDTD_HALO_E_DOWN = 1 };   /* Ix,Iy top plane -> rank+1 ; Vx,Vy bottom plane -> rank-1 */

/* Kernel selection (fdtd_desc.flags). */
enum {
  FDTD_FLAG_KERNEL_AUTO   = 0,   /* the schedule measured faster (fdtd_schedule_info tells which one a context took).  One launch per
                                    timestep (WAVEFRONT below) where it is possible — at least 2 planes, rows of at most
                                    30 720 cells — AND: on a single slab, when the fields exceed the 256 MiB Infinity Cache, or the slab
                                    has CPML layers, or a sweep has >= 1700 blocks of 1024 cells (Mur faces: single slabs within the cache); on a slab of a decomposed grid (p2p
                                    mailbox transport only), when the fields exceed the Infinity Cache or a sweep has >= 1800 blocks —
                                    unless a neighbour's slab lives on the SAME device (fdtd_p2p_link_info: same_device; several slabs
                                    of one process, or several ranks, on one GPU): the resident blocks of several one-launch kernels
                                    that spin on each other's halos can starve one another on a shared chip.
                                    Else two launches (Mur faces: the H kernel — like the H blocks of the one launch — reads the candidates of the post pass instead of the boundary
                                    voltages; three only when a voltage probe or an NF2FF box holds a node of a Mur face).  Slabs of different size may therefore step under
                                    different schedules in one run; the results do not depend on it. */
  FDTD_FLAG_KERNEL_DIRECT = 1,   /* the same, named explicitly */
  /* 2..4 were one-pass (fused E+H) variants — per-thread recompute, overlapped LDS tiles, z-marching tiles (round 1), and a
     z-marching kernel on an LDS-DMA ring (round 2, git history).  All were bit-exact and all measured slower than the two
     passes (profiles/r01, profiles/r02/one_pass_*); none ships: selecting them is FDTD_E_UNSUPPORTED. */
  FDTD_FLAG_KERNEL_WAVEFRONT = 5, /* ONE launch per timestep: the E sweep runs a few planes ahead of the H sweep, coupled by per-block
                                    flags, so that H reads what E just touched from the Infinity Cache instead of HBM.  Single slab, or
                                    slabs on the p2p mailbox transport; Mur faces on single slabs whose fields fit the Infinity Cache, with no source edge, voltage probe or
                                    NF2FF box on or next to a face (else FDTD_E_UNSUPPORTED).  Below 256 MiB of fields all E blocks
                                    run first, then all H blocks — and on a single slab ONE launch then holds SEVERAL timesteps (up to 64,
                                    cut at the timesteps whose NF2FF faces are sampled; fdtd_schedule_info info[7]): no kernel boundary, the E
                                    blocks of the next timestep start while the H blocks of this one drain.  DIRECT never takes it.
                                    Results are identical to the two-pass kernels bit for bit.  fdtd_profile.fused = 1: ms_update_e is the
                                    main-launch time per timestep. */
  FDTD_FLAG_KERNEL_RESIDENT = 6, /* the grid RESIDENT IN REGISTERS for the length of a launch (csrc/resident.hip): a workgroup owns a tile
                                    (1-2 planes x a few rows x all of x), keeps its fields and coefficients in registers over up to 256
                                    timesteps (cut at the timesteps whose NF2FF faces are sampled) and exchanges tile halos as data-tagged
                                    granules through a device-scope buffer — one ~1 us hop per half-step instead of a kernel boundary.
                                    Single slab, PEC / Mur faces (no CPML layers), rows of at most 1024 cells (256 with Mur z faces), no
                                    more tiles than the chip holds resident workgroups (else FDTD_E_UNSUPPORTED).  AUTO takes it for
                                    every such slab with Mur faces: the reference GUI's default scenes.  Results identical bit for bit. */
  FDTD_FLAG_KERNEL_MASK   = 0xF,
  FDTD_FLAG_OVERLAP_ON    = 0x20, /* multi-slab: split sweeps into interior + halo-dependent plane (the default) */
  FDTD_FLAG_OVERLAP_OFF   = 0x40, /* multi-slab: one launch per sweep, after the halo has arrived */
  FDTD_FLAG_LOOPBACK      = 0x80  /* transport self-test on ONE GPU: an interior slab (0 < rank < world-1) exchanges
                                     both halos with ITSELF — fdtd_comm_init makes a communicator of one rank and every
                                     ncclSend/ncclRecv of the step loop is a self send/recv.  Same result as driving
                                     fdtd_half_step + fdtd_halo_get/put back into the same context.
                                     With the p2p mailbox transport (fdtd_p2p_attach(ctx, own blob, own blob)) ANY slab of a
                                     decomposition can be timed alone on one GPU: it pulls the halos a slab in its place pulls
                                     (none below rank 0, none above the last rank) and pushes BOTH of its halo planes into its
                                     own mailbox (an end slab thereby pushes one plane more than it would in a real run: its
                                     time is an upper bound).  Fields are meaningless; tools/slab_balance.py. */
};

typedef struct fdtd_ctx fdtd_ctx;

typedef struct fdtd_desc {
  int32_t nx, ny, nz;     /* global node counts */
  int32_t k0, nk;         /* owned z-planes [k0, k0+nk) */
  int32_t rank, world;    /* position in the z-slab chain (rank r neighbours r-1 below, r+1 above) */
  int32_t device;         /* HIP device ordinal */
  int32_t max_steps;      /* capacity of the probe time series */
  uint32_t flags;
  double dt;              /* time step [s] (informational; all coefficients arrive pre-multiplied) */
} fdtd_desc;

typedef struct fdtd_profile {
  double ms_total;        /* stream time for the profiled steps, HIP events */
  double ms_update_e;     /* average begin-to-end duration of one E half-step main kernel launch; fused = 1: of the main launches PER
                             TIMESTEP (their durations summed, over `steps`: a launch may hold several timesteps, launches_e tells) */
  double ms_update_h;     /* average begin-to-end duration of one H half-step main kernel launch */
  int32_t launches_e;     /* launches averaged */
  int32_t launches_h;
  int32_t steps;
  int32_t fused;          /* 1: one launch per timestep (FDTD_FLAG_KERNEL_WAVEFRONT schedule): ms_update_e is that launch, ms_update_h = 0 */
  double ms_event_overhead; /* always 0 since ABI v2: the durations are the dispatches' own begin / end timestamps
                               (start / stop events carried by each launch), nothing is subtracted */
} fdtd_profile;

/* ---- lifecycle -------------------------------------------------------------------------- */
int  fdtd_version(void);
int  fdtd_device_count(void);
const char* fdtd_backend(void);                       /* "hip:gfx950" or "oracle:cpu" */
int  fdtd_create(const fdtd_desc* desc, fdtd_ctx** out);
void fdtd_destroy(fdtd_ctx* ctx);
const char* fdtd_last_error(const fdtd_ctx* ctx);     /* ctx may be NULL: last error of a failed create */

/* ---- operator (replaces openEMS's operator set-up inside FDTD.Run) ---------------------- */
/* Raw EC coefficients, each [3][nk][ny][nx]. */
int fdtd_set_operator_raw(fdtd_ctx* ctx, const float* vv, const float* vi,
                          const float* ii, const float* iv);
/* Compressed form: one class byte per edge + separable 1-D metric tables.
 *   vv   = cls_vv[c]
 *   vi_x = cls_m[c] * (ex[i] * (ey[j] * ez[k]))      with (ex,ey,ez) = emet rows of component x
 *   ii   = 1
 *   iv_x = hx[i] * (hy[j] * hz[k])                   (float32, exactly this association)
 * ecls: [3][nk][ny][nx];  emet/hmet: [3 comps][nx + ny + nk] (x table, then y, then LOCAL z). */
int fdtd_set_operator_classes(fdtd_ctx* ctx, const uint8_t* ecls, int ncls,
                              const float* cls_vv, const float* cls_m,
                              const float* emet, const float* hmet);

/* Operator set-up ON THE DEVICE (the operator-build phase the reference triggers inside FDTD.Run,
 * solver_fdtd_openems_fixed.py:280): per-cell materials, PEC edge flags and primal cell sizes in; the
 * class-compressed operator out (raw if the slab has more than 256 distinct (vv, m) pairs or prefer_classes == 0).
 * Bit for bit what the host formulation gives: for a live edge of component c (a1, a2 the other two axes),
 *   eps_e = EPS0 * sum(eps_r*w) / sum(w),  kap_e = sum(kappa*w) / sum(w)  over the 4 cells around the edge,
 *   w = d[a1]*d[a2] of the cell, summed with the a1 offset outer (-1, 0) and the a2 offset inner (-1, 0), float64;
 *   x = ((0.5*dt)*kap_e)/eps_e;  vv = (float)((1-x)/(1+x));  m = (float)(dt/(eps_e*(1+x)));
 * dead edges (pec != 0, last index along c, first/last index along a1 or a2) get vv = m = 0; then the overrides.
 *   dx,dy,dz   primal edge lengths of the GLOBAL grid (nx, ny, nz entries)
 *   eps_r,kappa  per cell of the GLOBAL grid [nz-1][ny-1][nx-1];  pec  [3][nz][ny][nx] bytes of the GLOBAL grid
 *   over_*     edges whose (vv, m) the host fixes itself (lumped elements): global edge index (k*ny + j)*nx + i
 *   emet,hmet  as in fdtd_set_operator_classes */
int fdtd_build_operator(fdtd_ctx* ctx, const double* dx, const double* dy, const double* dz,
                        const double* eps_r, const double* kappa, const uint8_t* pec, double eps0,
                        int n_over, const int64_t* over_edge, const int8_t* over_comp,
                        const float* over_vv, const float* over_m,
                        const float* emet, const float* hmet, int prefer_classes);
/* form: 0 none, 1 classes (one byte per edge), 2 classes packed (one byte per cell), 3 raw. */
int fdtd_operator_form(fdtd_ctx* ctx, int* form, int* nclasses);
/* The operator that is set, expanded to the four raw arrays, each [3][nk][ny][nx] (parity / debugging). */
int fdtd_get_operator(fdtd_ctx* ctx, float* vv, float* vi, float* ii, float* iv);

/* ---- absorbing boundaries --------------------------------------------------------------- */
/* CPML.  slot_a[idx] >= 0 gives the psi storage slot of index idx along axis a (-1: no psi).
 * coef: [3 axes][2 (0: E-located = node, 1: H-located = half node)][3 (b, c, 1/kappa)][n_a]
 * stored axis after axis with n_x = nx, n_y = ny, n_z = nk (LOCAL).  Outside the layers b=c=0, 1/kappa=1.
 *   psi = b*psi + c*d ;  term = (1/kappa)*d + psi     for every difference d along that axis. */
int fdtd_set_cpml(fdtd_ctx* ctx, const int32_t* slot_x, const int32_t* slot_y, const int32_t* slot_z,
                  int nslot_x, int nslot_y, int nslot_z, const float* coef);
/* First-order Mur on faces {x-,x+,y-,y+,z-,z+}; coeff = (c*dt - d)/(c*dt + d). z faces apply on the
 * ranks that own plane 0 / nz-1. */
int fdtd_set_mur(fdtd_ctx* ctx, const int32_t enable[6], const float coeff[6]);

/* ---- excitation, probes, frequency-domain recording ------------------------------------- */
int fdtd_set_signal(fdtd_ctx* ctx, const float* sig, int n);
/* V[comp[e]][idx[e]] += amp[e] * sig[step - delay[e]];  idx = (k*ny + j)*nx + i, GLOBAL k. */
int fdtd_add_source(fdtd_ctx* ctx, int n, const int64_t* idx, const int8_t* comp,
                    const float* amp, const int32_t* delay);
/* value[step] = sum_e w[e] * field[comp[e]][idx[e]] over the edges this slab owns. */
int fdtd_add_probe(fdtd_ctx* ctx, int kind, int n, const int64_t* idx, const int8_t* comp,
                   const float* w, int* id_out);
int fdtd_get_probe(fdtd_ctx* ctx, int id, double* out, int cap, int* n_out);
/* Running DFT: every `every` steps (steps every*s, s = 0..), for each registered box,
 *   acc[f] += field * tw[s][f]   (complex double; tw_v for V samples, tw_i for I samples).
 * tw_*: [nsamples][nfreq][2]. */
int fdtd_set_dft(fdtd_ctx* ctx, int nfreq, int every, int nsamples,
                 const double* tw_v, const double* tw_i);
/* lo/hi: inclusive GLOBAL node index box. */
int fdtd_add_dft_box(fdtd_ctx* ctx, int kind, int comp, const int32_t lo[3], const int32_t hi[3],
                     int* id_out);
/* Returns the part of the box inside this slab: lo_own/hi_own (GLOBAL indices; hi<lo if empty) and
 * out[nfreq][kk][jj][ii][2].  out may be NULL to query the extent only. */
int fdtd_get_dft_box(fdtd_ctx* ctx, int id, double* out, int32_t lo_own[3], int32_t hi_own[3]);

/* Time-domain recording of the same boxes — the alternative to fdtd_set_dft (call ONE of the two before the first
 * fdtd_add_dft_box).  This is what the reference's engine does with an NF2FF box: it dumps the surface fields in the time
 * domain and nf2ff.CalcNF2FF(sim_path, f, ...) transforms them to whatever frequency the caller asks for AFTER the run
 * (solver_fdtd_openems_fixed.py:220,296) — e.g. the S11 resonance, which is only known then
 * (solver_fdtd_openems_microstrip.py:407-433).  Every `every` steps the raw float32 samples of each box go to device
 * memory, [nsamples][npts] per box (sized for 288 GB of HBM; FDTD_E_NOMEM from fdtd_add_dft_box = use fdtd_set_dft). */
int fdtd_set_recorder(fdtd_ctx* ctx, int every, int nsamples);
/* out[nfreq][kk][jj][ii][2] = sum over the samples s recorded so far, in order, of sample_s * tw[s][f] (on the device,
 * float64 fma chain: bit for bit what the running DFT of fdtd_set_dft accumulates for the same tw).
 * tw: [nsamples][nfreq][2] for THIS box's field kind.  out may be NULL to query the owned extent only. */
int fdtd_rec_transform(fdtd_ctx* ctx, int id, int nfreq, const double* tw, double* out,
                       int32_t lo_own[3], int32_t hi_own[3]);

/* ---- time stepping ------------------------------------------------------------------------ */
/* Run nsteps full leapfrog steps (halo exchange inside when world > 1 and a communicator is set).
 * Blocks until the device is idle. */
int fdtd_run(fdtd_ctx* ctx, int nsteps);
/* Same (1..4096 steps), with start / stop events on every main-kernel launch: their begin / end timestamps. */
int fdtd_run_profiled(fdtd_ctx* ctx, int nsteps, fdtd_profile* out);
int fdtd_get_step(fdtd_ctx* ctx, int64_t* step);
/* The step schedule this context runs under its current flags, boundaries and transport:
 *   info[0] main-kernel launches per timestep (1: k_step, 2: update_E + update_H, 3: with Mur faces; 0: driven by
 *           fdtd_half_step / not steppable yet)
 *   info[1] one launch per timestep only: planes the E sweep runs ahead of the H sweep (== nk: all E blocks, then all H blocks);
 *           -1: the resident schedule (FDTD_FLAG_KERNEL_RESIDENT: info[0] = 1, info[2] = strips, info[3] = tiles = workgroups,
 *           info[7] = timesteps one launch may hold)
 *   info[2] rows per strip, info[3] blocks (of 1024 cells) per sweep
 *   info[4] halo transport: 0 none (single slab), 1 p2p mailbox, 2 RCCL, 3 linked contexts, 4 external (fdtd_half_step)
 *   info[5] 1 if the XCD shares are cost-weighted (CPML layers present)
 *   info[6] how often the shares have been re-cut from MEASURED per-XCD finish times so far (the last launch of an
 *           fdtd_run call of >= 16 timesteps on a single slab is a calibration launch: every block leaves its end time)
 *   info[7] one launch per timestep only: how many timesteps ONE launch may hold (cache-resident single slabs: up to 64, cut
 *           at the timesteps whose NF2FF faces are sampled; 1: a launch per timestep) */
int fdtd_schedule_info(fdtd_ctx* ctx, int32_t info[8]);
/* sums[0] = sum V^2, sums[1] = sum I^2 over the owned planes. */
int fdtd_energy(fdtd_ctx* ctx, double sums[2]);

/* ---- halo transport ----------------------------------------------------------------------- */
/* (a) RCCL over xGMI, owned by the library.  Rank 0 creates the id, the host program ships the
 *     128 bytes to every rank (e.g. torch.distributed broadcast), every rank calls comm_init. */
int fdtd_comm_unique_id(void* out128);
int fdtd_comm_init(fdtd_ctx* ctx, const void* uid128);
/* Ranks of the RCCL communicator this context steps through (ncclCommCount); 0 when none is attached. */
int fdtd_comm_nranks(fdtd_ctx* ctx, int* nranks);
/* (a0) P2P mailbox transport — the default for one process per GPU: the update kernels push the outgoing halo plane
 *      straight into the neighbour's mailbox (peer / IPC mapping, xGMI stores) and publish a step counter there; the
 *      neighbour's kernel waits for the counter before it touches the halo-dependent plane, which it schedules last.
 *      A timestep is then two launches on one stream: no communication stream, no events, no RCCL call.  Measured
 *      on MI355X, an 8-plane slab of the north-star grid steps in ~80 us with grouped ncclSend/ncclRecv (or peer copies)
 *      ordered by events on a second stream, however small the slab — the reason for this transport.
 *      Every rank exports a 128-byte blob, the host program ships the blobs (e.g. torch.distributed all_gather),
 *      every rank attaches its neighbours' blobs (null where there is none).  Needs >= 2 planes per slab, no Mur.
 *      Tear-down: a rank's last H half-step still writes into its upper neighbour's mailbox, so synchronise the ranks
 *      (a barrier after the last fdtd_run) before any of them calls fdtd_p2p_detach or fdtd_destroy. */
int fdtd_p2p_export(fdtd_ctx* ctx, void* out128);
int fdtd_p2p_attach(fdtd_ctx* ctx, const void* lower128, const void* upper128);
/* Hand-shake with the attached neighbours (token written into their mailboxes, theirs awaited for <= 10 s): call on
 * all ranks at about the same time, before the first step.  Error = fall back (fdtd_p2p_detach, then e.g. RCCL).
 * The test ends by zeroing THIS rank's mailbox.  Callers must therefore synchronise all ranks (a barrier) between the
 * self-test and the first fdtd_run: a neighbour that starts stepping earlier pushes its initial halo (tag 1) into a
 * mailbox this rank may still be verifying or zeroing.  (distributed.SlabComm.attach does: its agreement all-reduce.) */
int fdtd_p2p_selftest(fdtd_ctx* ctx, unsigned token);
int fdtd_p2p_detach(fdtd_ctx* ctx);
/* The link between this context's GPU and an attached neighbour's (which: 0 = lower rank, 1 = upper rank), so that a
 * multi-GPU record can tell an xGMI neighbour from a PCIe one:
 *   info[0] neighbour's device ordinal in THIS process (-1: not attached; -2: attached, but that GPU is not visible here)
 *   info[1] 1 same process (plain pointer), 2 another process (HIP IPC mapping)
 *   info[2] link type   (hipExtGetLinkTypeAndHopCount: HSA_AMD_LINK_INFO_TYPE_*: 0 HyperTransport, 1 QPI, 2 PCIe, 3 InfiniBand, 4 xGMI)
 *   info[3] hop count   (same call; 0 for the same device)
 *   info[4] hipDevP2PAttrPerformanceRank, info[5] hipDevP2PAttrAccessSupported, info[6] hipDevP2PAttrNativeAtomicSupported
 *   info[7] 1 if the neighbour is this very device (several ranks on one GPU, or a loopback slab)
 * Entries the runtime cannot tell are -1. */
int fdtd_p2p_link_info(fdtd_ctx* ctx, int which, int32_t info[8]);
/* (a') Several slabs inside ONE process (one host thread driving several GPUs, or several slabs on one GPU):
 *      link adjacent contexts, then step them together; halos move by peer copies on the communication
 *      streams with the same overlapped schedule as the RCCL path. ctxs[r] must be rank r of a world of n. */
int fdtd_link(fdtd_ctx* lower, fdtd_ctx* upper);
int fdtd_run_linked(fdtd_ctx** ctxs, int n, int nsteps);
/* (b) External transport (MPI, gloo, ...): one half-step at a time, halos through host buffers.
 *     buf: [2][ny][nx] floats = the two tangential components (x then y) of one plane. */
int fdtd_half_step(fdtd_ctx* ctx, int phase);
int fdtd_halo_get(fdtd_ctx* ctx, int which, float* buf);
int fdtd_halo_put(fdtd_ctx* ctx, int which, const float* buf);

/* ---- field access (parity / debugging) ---------------------------------------------------- */
int fdtd_get_field(fdtd_ctx* ctx, int kind, int comp, float* out);      /* [nk][ny][nx] */
int fdtd_set_field(fdtd_ctx* ctx, int kind, int comp, const float* in);

/* ---- near-to-far-field transform (replaces nf2ff.CalcNF2FF's radiation integral) ---------- */
/* Surface equivalence: npts quadrature points with position pos[p][3] (relative to the phase
 * centre, metres) and area-weighted complex equivalent currents Js = n x H * dA, Ms = -n x E * dA,
 * [p][3][2].  For every direction a: r^ = (sin th cos ph, sin th sin ph, cos th)
 *   N = sum_p Js_p exp(+j k r^.pos_p),  L = sum_p Ms_p exp(+j k r^.pos_p)
 *   Eth[a] = -j k/(4 pi) (L_ph + eta0 N_th),  Eph[a] = +j k/(4 pi) (L_th - eta0 N_ph)   (r*E, far zone)
 * Eth/Eph: [nang][2]. */
int fdtd_farfield(int device, int npts, const double* pos, const double* Js, const double* Ms,
                  double k_wave, int nang, const double* theta, const double* phi,
                  double* Eth, double* Eph);

#ifdef __cplusplus
}
#endif
#endif /* FDTD_HIP_H */
