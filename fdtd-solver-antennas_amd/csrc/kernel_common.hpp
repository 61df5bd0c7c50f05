// kernel_common.hpp — device helpers of the update kernels (kernels.hip) and of the mailbox self-test (api.hip).
#pragma once
#include "fdtd_ctx.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
// Scalar base + 32-bit unsigned BYTE offset: selects the `global_load ... v_off, s[base]` addressing form, i.e. one
// VGPR per address instead of a 64-bit pair (the update kernels are occupancy-sensitive).  e = element offset >= 0.
__device__ __forceinline__ float4 ldo4(const float* base, const unsigned e) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (e << 2));
}
__device__ __forceinline__ float ldo1(const float* base, const unsigned e) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (e << 2));
}
// streamed outputs: non-temporal once the working set exceeds the 256 MiB Infinity Cache (measured on
// MI355X: +3-4 % at 400x400x80, -15 % at 300x300x60 where the fields live in the cache)
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4s(const int nt, float* p, const float4& v) {
  if (nt) {
    v4f_nt t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_nt*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}
__device__ __forceinline__ void sto4s(const int nt, float* base, const unsigned e, const float4& v) {
  st4s(nt, reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (e << 2)), v);
}
// 8-byte load through the scalar cache from a wave-uniform address (data the host wrote before the launch); waits only for
// scalar-memory returns, never for the vector loads in flight
__device__ __forceinline__ int2 sload_int2(const int2* q) {
  unsigned long long v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q));
  return make_int2((int)(unsigned)v, (int)(unsigned)(v >> 32));
}
__device__ __forceinline__ float4 sub4(const float4& a, const float4& b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}

// XCD-aware, strip-major block decode.  Block index = 8 * position + XCD (workgroups go round-robin to the eight XCDs, each
// with its own 4 MiB L2); XCD x sweeps the contiguous range [xs[x], xs[x+1]) of the strip-major block order, so that the
// k+-1 and j+-1 neighbour rows are hits in ITS L2.  The ranges are equal in COST, not in length (xcd_shares, kernels.hip):
// blocks of CPML planes / rows move 40-50 % more bytes, and the XCDs do not help each other out — with equal lengths the XCDs
// that own the y-layer strips (the first and the last) and, on short grids, the z-layer planes finish last and everybody
// else idles.  Positions beyond an XCD's share are empty blocks at the END of the dispatch order: they return at once.
__device__ __forceinline__ unsigned fd_div(const unsigned n, const FastDiv& f) { return f.d == 1u ? n : __umulhi(n, f.mul) >> f.shr; }
// bb: block index within the main part of the launch.  rev: this XCD walks its range backwards.  update_E sweeps forwards
// and update_H backwards, so each half-step starts on the data the previous one touched last (still in this XCD's L2 / the
// Infinity Cache) instead of on the data it touched first (evicted by then on grids larger than the caches).
__device__ __forceinline__ bool xcd_position(const unsigned (&xs)[9], const unsigned bb, const int rev, unsigned& v) {
  const unsigned xcd = bb & 7u;
  unsigned pos = bb >> 3;
  const unsigned first = xs[xcd], cnt = xs[xcd + 1] - first;
  if (pos >= cnt) return false;
  if (rev) pos = cnt - 1u - pos;
  v = first + pos;
  return true;
}
// strip-major order: v = (strip * planes + kk) * nbs + pb (fd_ps divides by planes * nbs)
__device__ __forceinline__ bool decode_block_fd(const DevParams& p, const FastDiv& fd_ps, const FastDiv& fd_nbs, int rev, int& strip, int& kk, int& pb,
                                                const unsigned first = 0u /* blocks in front of the main ones: a multiple of 8 */) {
  unsigned v;
  if (!xcd_position(p.xs, blockIdx.x - first, rev, v)) return false;
  const unsigned s = fd_div(v, fd_ps);
  const unsigned rem = v - s * fd_ps.d;
  const unsigned k = fd_div(rem, fd_nbs);
  strip = (int)s; kk = (int)k; pb = (int)(rem - k * fd_nbs.d);
  return true;
}
// ---- P2P mailbox protocol -------------------------------------------------------------------------------------------
// A halo value travels as an 8-byte GRANULE {value, tag}: tag = number of the timestep whose half-step produced it (+1 for E, +2 for
// H: the H halo of step s is what the E sweep of step s+1 expects under tag s+2; 0 is never a valid tag), in
// the same naturally aligned 8 bytes as the value, so whoever sees the tag sees the value.  No flag, no arrival counter, no
// wait for store acknowledgements on the producer; ONE round trip on the consumer (it loads the granules together with its
// field loads and looks at the tags; only a late neighbour makes it load again).  The earlier protocol — payload stores,
// vmcnt(0), wave counter, step flag; flag poll, then payload loads — cost 13 us per timestep on a thin slab (two more
// dependent round trips in each kernel's halo plane: profiles/r02/thin_slab_p2p_protocol.txt).
// A thread's float4 goes out as two 16-byte stores {x, t, y, t}, {z, t, w, t} (a 16-byte store may tear into its 8-byte
// halves: each half is a granule of its own).  Mailbox words are only ever touched with system-scope (sc0 sc1) accesses,
// which go past this XCD's L2 to memory / the fabric (the mailbox is fine-grained memory: another GPU writes it while this
// one reads it).  Slots alternate with the parity of the step, so a value is overwritten two steps later — by then the
// writer has consumed the reader's own halo of the step in between, which the reader produced after reading this one.
typedef float v4f_sys __attribute__((ext_vector_type(4)));
// The s_nop is part of the instruction's contract here: a vector-memory store of more than 64 bits reads its data
// registers AFTER issue, and a VALU write to them in the next wait states corrupts what the later lanes store (hipcc pads
// its own stores; it cannot see into an asm statement).  Found by the data-path self-test of the mailbox transport.
__device__ __forceinline__ void st4_sys(float* q, const float4& v) {
  const v4f_sys t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(q), "v"(t) : "memory");
}
// words of one (parity, component) slot: 2 per cell
__device__ __forceinline__ size_t mb_slot_words(const DevParams& p) { return (size_t)2 * p.plane; }
// push the four values of cell group `o` (element offset inside the plane, a multiple of 4) into slot `slot`
__device__ __forceinline__ void mb_push(float* slot, const unsigned o, const unsigned tag, const float4& v) {
  float* q = slot + 2u * o;
  const float t = __uint_as_float(tag);
  st4_sys(q, make_float4(v.x, t, v.y, t));
  st4_sys(q + 4, make_float4(v.z, t, v.w, t));
}
// pull the two halo components of cell group `o`: load both groups' granules, accept when all eight tags are `tag`; a wave
// whose neighbour is late loads again (bounded: `limit` ticks of the wall clock, then the error word — never a hang; once
// the error word is set nobody spins).  Every lane of the wave takes part (lanes beyond the strip pull group 0).
// A wait that gives up says what it was waiting for: err[0] = 1 (the error word every waiter looks at), and — written by the ONE lane that
// set it — err[1] = `who` (the caller's code for launch kind / plane / strip / block), err[2] = the tag awaited, err[3] = the tag found
// instead, err[4] = the group offset.  The host turns the record into the message of fdtd_last_error (api.hip: p2p_check).
__device__ __forceinline__ void mb_pull2(const float* slot_a, const float* slot_b, const unsigned o, const unsigned tag,
                                         float4& a, float4& b, int* err, const unsigned long long limit, const unsigned who = 0u) {
  const float* qa = slot_a + 2u * o;
  const float* qb = slot_b + 2u * o;
  v4f_sys a0, a1, b0, b1;
  unsigned long long t0 = 0ull;
  for (int round = 0;; ++round) {
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc0 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(qa), "v"(qb) : "memory");
    const bool ok = __float_as_uint(a0.y) == tag && __float_as_uint(a0.w) == tag && __float_as_uint(a1.y) == tag && __float_as_uint(a1.w) == tag &&
                    __float_as_uint(b0.y) == tag && __float_as_uint(b0.w) == tag && __float_as_uint(b1.y) == tag && __float_as_uint(b1.w) == tag;
    if (__ballot(!ok) == 0ull) break;
    if (round == 0) t0 = wall_clock64();
    __builtin_amdgcn_s_sleep(4);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    if ((unsigned long long)wall_clock64() - t0 > limit) {
      if (!ok && atomicCAS(err, 0, 1) == 0) {   // first to give up: leave the record
        const unsigned seen = __float_as_uint(a0.y) != tag ? __float_as_uint(a0.y) : __float_as_uint(a0.w) != tag ? __float_as_uint(a0.w) :
                              __float_as_uint(a1.y) != tag ? __float_as_uint(a1.y) : __float_as_uint(a1.w) != tag ? __float_as_uint(a1.w) :
                              __float_as_uint(b0.y) != tag ? __float_as_uint(b0.y) : __float_as_uint(b0.w) != tag ? __float_as_uint(b0.w) :
                              __float_as_uint(b1.y) != tag ? __float_as_uint(b1.y) : __float_as_uint(b1.w);
        err[1] = (int)who; err[2] = (int)tag; err[3] = (int)seen; err[4] = (int)o;
      }
      break;
    }
  }
  a = make_float4(a0.x, a0.z, a1.x, a1.z);
  b = make_float4(b0.x, b0.z, b1.x, b1.z);
}
// Block decode of the P2P launches.  dep_plane: 0 for update_E (the main part covers planes 1..), nk-1 for update_H (main
// part planes 0..nk-2).  dep_first: the halo plane's blocks are the FIRST blocks in dispatch order — its pull from and push to
// the neighbour's mailbox (fine-grained memory: a round trip of several microseconds) then run beside the other planes
// instead of at the kernel's tail; a neighbour that is late makes these blocks load again while the rest proceeds.  With
// dep_first = 0 they are the last ones (what the flag protocol of round 1 needed, when a wait could last a kernel).
__device__ __forceinline__ bool decode_block_p2p(const DevParams& p, const FastDiv& fd_ps, const FastDiv& fd_nbs, unsigned nb_main, unsigned nb_dep, int dep_plane,
                                                 int main_first, int dep_first, int& strip, int& k, int& pb) {
  const unsigned b = blockIdx.x;
  unsigned v;
  if (dep_first ? b < nb_dep : b >= p.xgrid) {           // a block of the halo plane
    v = nb_main + (dep_first ? b : b - p.xgrid);
  } else if (!xcd_position(p.xs, dep_first ? b - nb_dep : b, 0, v)) return false;
  if (v < nb_main) {
    const unsigned s = fd_div(v, fd_ps);
    const unsigned rem = v - s * fd_ps.d;
    const unsigned kk = fd_div(rem, fd_nbs);
    strip = (int)s; k = main_first + (int)kk; pb = (int)(rem - kk * fd_nbs.d);
  } else {
    const unsigned w = v - nb_main;
    const unsigned s = fd_div(w, fd_nbs);
    strip = (int)s; k = dep_plane; pb = (int)(w - s * fd_nbs.d);
  }
  return true;
}

// ---- one launch per timestep (k_step): device-scope accesses and the per-block flags ------------------------------------
// V written by an E block is read by H blocks on other CUs (possibly other XCDs) inside the same launch.  A CU's L1 is
// never refreshed by another CU's stores and the per-XCD L2s are not coherent with each other, so: write-through (sc1)
// stores on the producer, sc1 loads to registers on the consumer (MI355X_MICROARCH.md, inter-workgroup visibility).
// Buffer instructions because the compiler has builtins for them WITH the cache-policy operand (aux 16 = sc1 on gfx950),
// i.e. it tracks them in its own s_waitcnt bookkeeping and pads the wide store's data hazard itself.
typedef unsigned v4u_dev __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t DevRsrc;
__device__ __forceinline__ DevRsrc dev_buf(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);   // raw buffer, no stride, no range limit
}
__device__ __forceinline__ float4 ldb4_dev(const DevRsrc b, const unsigned byte_off, const unsigned sdisp) {
  const v4u_dev r = __builtin_amdgcn_raw_buffer_load_b128(b, (int)byte_off, (int)sdisp, 16);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
__device__ __forceinline__ float ldb1_dev(const DevRsrc b, const unsigned byte_off, const unsigned sdisp) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b, (int)byte_off, (int)sdisp, 16));
}
__device__ __forceinline__ void sto4_dev(float* base, const unsigned e, const float4& v) {
  const v4u_dev t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(t, dev_buf(base), (int)(e << 2), 0, 16);
}
__device__ __forceinline__ void sto1_dev(float* base, const unsigned e, const float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dev_buf(base), (int)(e << 2), 0, 16);
}
// Block (k, strip, pb) is done: every wave has its stores acknowledged, the block meets, one lane publishes.
__device__ __forceinline__ void wf_publish(const DevParams& p, unsigned* flags, const int k, const int strip, const int pb, const unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(flags + ((size_t)k * p.nstrips + strip) * p.nbs + pb, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int sload_int(const int* q) {
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q));
  return v;
}
// bounded poll of one flag (any lane).  The first wait to give up leaves a record behind the error word: wf_err[1] = index of the flag (which
// table: the pointer tells, wf_err[5]), [2] = the value awaited, [3] = the value found, [4] = the waiting block (blockIdx.x)
__device__ __forceinline__ void wf_poll(const DevParams& p, const unsigned* f, const unsigned target) {
  const unsigned long long t0 = wall_clock64();
  unsigned seen;
  while ((seen = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (__hip_atomic_load(p.wf_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    if ((unsigned long long)wall_clock64() - t0 > p.wf_limit) {
      if (atomicCAS(p.wf_err, 0, 1) == 0) {
        const long de = f - p.wf_flags, dh = f - p.wf_flagsH;
        const bool is_e = de >= 0 && de < (long)p.nk * p.nstrips * p.nbs;
        p.wf_err[1] = (int)(is_e ? de : dh); p.wf_err[2] = (int)target; p.wf_err[3] = (int)seen; p.wf_err[4] = (int)blockIdx.x;
        p.wf_err[5] = is_e ? 0 : (dh >= 0 && dh < (long)p.nk * p.nstrips * p.nbs ? 1 : 2);   // 0: an E block's flag, 1: an H block's, 2: a probe block's
      }
      break;
    }
  }
}
// Probe q of step `step` as one of the LAST blocks of the step's launch: wait for the blocks that own its cells (E flags for
// a V-probe, H flags for an I-probe: those H blocks stored write-through), read the cells with device-scope loads, reduce
// with probe_block's tree (identical sums).
// MULTI (several timesteps per launch): when the cells have been read the block says so (wf_prb_done[q] = target) — the blocks
// that overwrite probe cells in the NEXT timestep of the launch wait for that.
template <bool MULTI = false>
__device__ __forceinline__ void wf_probe_tail(const DevParams& p, const int q, const long long step, const unsigned target, double* red) {
  if (step < 0 || step >= p.max_steps) {
    if (MULTI && threadIdx.x == 0) __hip_atomic_store(p.wf_prb_done + q, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const DevProbe pr = p.probes[q];
  const int2 rng = p.wf_prb_rng[q];
  const unsigned* flags = pr.kind == FDTD_KIND_V ? p.wf_flags : p.wf_flagsH;
  for (int e = rng.x + (int)threadIdx.x; e < rng.y; e += FDTD_BLOCK) wf_poll(p, flags + p.wf_prb_blk[e], target);
  __syncthreads();
  double s = 0.0;
  for (int e = threadIdx.x; e < pr.n; e += FDTD_BLOCK) {
    const float* F = (pr.kind == FDTD_KIND_V ? p.V[pr.comp[e]] : p.I[pr.comp[e]]);
    const unsigned bits = __hip_atomic_load(reinterpret_cast<const unsigned*>(F + pr.off[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = fma((double)pr.w[e], (double)__uint_as_float(bits), s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = FDTD_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pr.series[step] = red[0];
    if (MULTI) __hip_atomic_store(p.wf_prb_done + q, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (every thread's loads were consumed before the first barrier above)
  }
}
// A block that is about to overwrite cells a probe of `kind` samples: wait until the probe blocks of the previous timestep
// have read them (one lane per probe; FDTD_MAX_PROBES <= 64)
__device__ __forceinline__ void wf_wait_probes(const DevParams& p, const int kind, const unsigned target) {
  if ((int)threadIdx.x < p.nprobe && p.probes[threadIdx.x].kind == kind) wf_poll(p, p.wf_prb_done + threadIdx.x, target);
  __syncthreads();
}
// E block (k, strip, pb) of a LATER timestep of a multi-step launch: wait for the H blocks of the previous timestep that wrote
// the I values it reads and read the V values it overwrites (the same set): threads t - P4 - 1 .. t + 255 of the strip-plane
// (its own cells, the row below, the element left of a group) — running into the previous strip's last rows at a strip's
// start — and the same block of plane k - 1.  The mirror image of wf_wait below.
__device__ __forceinline__ void wf_wait_back(const DevParams& p, const int k, const int strip, const int pb, const unsigned target) {
  if (threadIdx.x < 64u) {
    const int hop = 1 + p.P4 / FDTD_BLOCK;
    const int first = pb * FDTD_BLOCK - p.P4 - 1;                 // first strip-linear thread whose data is read (may be negative)
    const int t = (int)threadIdx.x;
    const unsigned* f = nullptr;
    const size_t row = ((size_t)k * p.nstrips + strip) * p.nbs;
    if (t <= hop) {                                               // same strip-plane: this block and the ones before it
      if (pb - t >= 0 && (pb - t + 1) * FDTD_BLOCK - 1 >= first) f = p.wf_flagsH + row + pb - t;
    } else if (t <= 2 * hop + 1) {                                // previous strip's last blocks (always a full strip)
      const int q = t - hop - 1, Tp = p.tys * p.P4, lastb = (Tp - 1) / FDTD_BLOCK;
      if (first < 0 && strip > 0 && lastb - q >= 0 && min((lastb - q + 1) * FDTD_BLOCK, Tp) - 1 >= Tp + first) f = p.wf_flagsH + row - p.nbs + lastb - q;
    } else if (t == 2 * hop + 2) {                                // plane below
      if (k > 0) f = p.wf_flagsH + row - (size_t)p.nstrips * p.nbs + pb;
    }
    if (f) wf_poll(p, f, target);
  }
  __syncthreads();
}
// H block (k, strip, pb): wait for the E blocks whose output it reads / whose input it overwrites.  Threads t .. t+255 of
// the strip-plane read rows j and j+1 (thread t + P4) and the element right of their group (thread t + 1): strip-linear
// threads [pb*256, pb*256 + 256 + P4] at plane k — running into the next strip's first rows at a strip's end — and the
// same block at plane k + 1.  One lane per flag, wave 0 polls, the block meets.
__device__ __forceinline__ void wf_wait(const DevParams& p, const int k, const int strip, const int pb, const unsigned target) {
  if (threadIdx.x < 64u) {
    const int hop = 1 + p.P4 / FDTD_BLOCK;
    const int rows = min(p.tys, p.ny - strip * p.tys);
    const int T = rows * p.P4;                                   // threads of this strip-plane
    const int last = pb * FDTD_BLOCK + FDTD_BLOCK + p.P4;        // last strip-linear thread whose data is read
    const int t = (int)threadIdx.x;
    const unsigned* f = nullptr;
    const size_t row = ((size_t)k * p.nstrips + strip) * p.nbs;
    if (t <= hop) {                                              // same strip, this block and the ones behind it
      if ((pb + t) * FDTD_BLOCK <= min(last, T - 1)) f = p.wf_flags + row + pb + t;
    } else if (t <= 2 * hop + 1) {                               // next strip's first blocks
      const int q = t - hop - 1;
      if (last >= T && strip + 1 < p.nstrips && q < p.nbs && q * FDTD_BLOCK <= last - T) f = p.wf_flags + row + p.nbs + q;
    } else if (t == 2 * hop + 2) {                               // plane above
      if (k + 1 < p.nk) f = p.wf_flags + row + (size_t)p.nstrips * p.nbs + pb;
    }
    if (f) wf_poll(p, f, target);
  }
  __syncthreads();
}

// Mur faces inside the one-launch schedule: an H block also reads CANDIDATES that E blocks wrote — those of the inner nodes, one row / plane away
// from the boundary nodes it loads (kernels.hip mur_load_V): the row behind it, the whole reach at plane k + 1, plane k - 1 under the upper z face.
// Waited for generously: every block of strips s - 1 .. s + 1 at planes k - 1 .. k + 1 (9 nbs flags, one thread each; the host takes this schedule only
// while they fit the workgroup).  All of them are E blocks of the SAME timestep: earlier in dispatch order (all E blocks, then all H blocks).
__device__ __forceinline__ void wf_wait_mur(const DevParams& p, const int k, const int strip, const unsigned target) {
  {
    const int t = (int)threadIdx.x, nbs = p.nbs;
    const int dk = t / (3 * nbs) - 1, r = t - (dk + 1) * 3 * nbs, ds = r / nbs - 1, q = r - (ds + 1) * nbs;
    const int kk = k + dk, ss = strip + ds;
    if (t < 9 * nbs && kk >= 0 && kk < p.nk && ss >= 0 && ss < p.nstrips)
      wf_poll(p, p.wf_flags + ((size_t)kk * p.nstrips + ss) * nbs + q, target);
  }
  __syncthreads();
}

// ... and the per-thread part.  Returns false for threads beyond the strip.
__device__ __forceinline__ bool decode_thread(const DevParams& p, int strip, int pb, int& j, int& i0) {
  const int t = pb * FDTD_BLOCK + (int)threadIdx.x;
  const int rows = min(p.tys, p.ny - strip * p.tys);
  if (t >= rows * p.P4) return false;
  const int jj = (int)fd_div((unsigned)t, p.fd_P4);
  j = strip * p.tys + jj;
  i0 = (t - jj * p.P4) * 4;
  return true;
}

// Soft sources of one strip-plane, staged once per block: (flat offset | comp << 29 is avoided: two arrays).  Up to SRC_SCAN_MAX of them every
// thread scans (apply_staged); a strip-plane with more takes the dense LDS image of body_E.
constexpr int SRC_SCAN_MAX = 32;
struct SrcStage { int off[FDTD_BLOCK]; float val[FDTD_BLOCK]; signed char comp[FDTD_BLOCK]; };
// Fill the stage with sources [begin, begin+n) of the id list (n <= FDTD_BLOCK); value = amp*sig[step-delay] or 0.
__device__ __forceinline__ void stage_sources(const DevParams& p, const int* ids, int begin, int n, long long step, SrcStage& st) {
  const int q = threadIdx.x;
  if (q < n) {
    const int e = ids[begin + q];
    const long long t = step - p.src_delay[e];
    st.off[q] = p.src_off[e];
    st.comp[q] = p.src_comp[e];
    st.val[q] = (t >= 0 && t < p.nsig) ? p.src_amp[e] * p.sig[t] : 0.f;
  }
}
// V += amp*sig for staged sources that fall on component `comp` of the four cells at flat offset o
__device__ __forceinline__ void apply_staged(const SrcStage& st, int n, int comp, int o, float4& v) {
  for (int q = 0; q < n; ++q) {
    const unsigned rel = (unsigned)(st.off[q] - o);
    if (rel < 4u && st.comp[q] == comp && st.val[q] != 0.f) {
      const float a = st.val[q];
      if (rel == 0) v.x = v.x + a; else if (rel == 1) v.y = v.y + a; else if (rel == 2) v.z = v.z + a; else v.w = v.w + a;
    }
  }
}

// value[step] = sum_e w[e]*field[e] for every probe of `kind` (one block, fixed reduction tree)
// only >= 0: that probe alone (the update kernels carry one such block per probe: a multi-port scene has several probes of a thousand
// cells each, and ONE block walking through them all outlasted the main blocks — 143x129x89 with four ports: 10 us of a 38 us timestep)
__device__ __forceinline__ void probe_block(const DevParams& p, const int kind, const long long step, double* red, const int only = -1) {
  if (step < 0 || step >= p.max_steps) return;
  for (int q = only >= 0 ? only : 0; q < (only >= 0 ? only + 1 : p.nprobe); ++q) {
    const DevProbe pr = p.probes[q];
    if (pr.kind != kind) continue;
    double s = 0.0;
    for (int e = threadIdx.x; e < pr.n; e += FDTD_BLOCK) {
      const float* F = (kind == FDTD_KIND_V ? p.V[pr.comp[e]] : p.I[pr.comp[e]]);
      s = fma((double)pr.w[e], (double)F[pr.off[e]], s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = FDTD_BLOCK / 2; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) pr.series[step] = red[0];
    __syncthreads();
  }
}

__device__ __forceinline__ void add_elem(float4& v, int e, float a) {
  if (e == 0) v.x = v.x + a; else if (e == 1) v.y = v.y + a; else if (e == 2) v.z = v.z + a; else v.w = v.w + a;
}

__device__ __forceinline__ int pml_slot(const DevParams& p, int a, int q) {
  return q < p.pml_lo[a] ? q : (q >= p.pml_hi[a] ? q - p.pml_hi[a] + p.pml_hi_slot[a] : -1);
}

// psi <- b*psi + c*d ; d <- d/kappa + psi   for the four cells of a thread (row-uniform coefficients)
__device__ __forceinline__ void cpml_row4(float4& d, float* base, const unsigned o, float b, float c, float ik) {
  float* psi = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (o << 2));   // scalar base + 32-bit offset
  float4 ps = ld4(psi);
  ps.x = __builtin_fmaf(b, ps.x, c * d.x);
  ps.y = __builtin_fmaf(b, ps.y, c * d.y);
  ps.z = __builtin_fmaf(b, ps.z, c * d.z);
  ps.w = __builtin_fmaf(b, ps.w, c * d.w);
  st4(psi, ps);
  d.x = __builtin_fmaf(ik, d.x, ps.x);
  d.y = __builtin_fmaf(ik, d.y, ps.y);
  d.z = __builtin_fmaf(ik, d.z, ps.z);
  d.w = __builtin_fmaf(ik, d.w, ps.w);
}

// ---- CPML psi through LDS-DMA ------------------------------------------------------------------------------------------
// The psi loads of a layer cell depend on nothing but the cell's indices, yet they are only USED after the curl
// differences exist.  Issued there (round 1) they are a second, exposed memory round trip for every wave that touches a
// layer — measured on MI355X: the x layers (6 % of the cells, but some lanes of EVERY wave) cost 39 us of a 242 us step at
// 400x400x80, the z layers 30 us.  Issued early into registers they cost 8 VGPRs per axis and a wave of occupancy.  So they
// are issued early into LDS instead (global_load_lds_dwordx4: no register destination), right behind the field loads,
// and read back from LDS when the differences are ready.  Each wave owns PSI_SLOTS KiB of staging, lane-linear per slot
// (the LDS destination of an LDS-DMA load is wave-uniform base + lane * 16): slots 0,1 the x-directed pair, 2,3 the z pair.
// The x-neighbour scalars of the stencils — I(i0 - 1) for update_E, V(i0 + 4) for update_H — are the last / first element of the float4 the
// NEIGHBOUR LANE has just loaded (threads are linear in memory within a strip-plane, rows included): one DPP wave shift instead of a second
// vector-memory instruction whose 64 lanes touch sixteen 64-byte segments for 4 bytes each.  Only the lane at the wave's edge (and, for H, the
// last thread of a strip-plane, whose neighbour thread does not exist) still loads from memory.  0: the scalar loads of rounds 1-2.
#ifndef FDTD_LANE_SHIFT
#define FDTD_LANE_SHIFT 1
#endif
// ... where the kernel has the register it costs: the launches of several timesteps (MULTI) keep their 7 waves per SIMD; the one-launch kernels of
// the grids beyond the Infinity Cache sit exactly on 72 VGPRs and would spill (12-60 bytes per lane), several two-launch kernels would drop from 7
// to 6 waves per SIMD (tools/kernel_resources.py)
#define LANE_SHIFT_FITS(WF, MULTI) (MULTI)
__device__ __forceinline__ float lane_prev(const float v) {   // lane l <- lane l - 1 (wave_shr:1); lane 0 gets 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(const float v) {   // lane l <- lane l + 1 (wave_shl:1); lane 63 gets 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

#ifndef FDTD_PSI_STAGE
#define FDTD_PSI_STAGE 1
#endif
constexpr int PSI_SLOTS = 4;
constexpr int XC_MAX = 128;   // x-layer cells (both sides, 4-aligned) whose (b, c, 1/kappa) fit the LDS coefficient table
__device__ __forceinline__ unsigned lds_off(const void* q) { return (unsigned)(size_t)q; }   // LDS byte address of a __shared__ object
// one 16-byte LDS-DMA load per active lane: global gsrc -> LDS lds_dst + lane*16 (lds_dst wave-uniform).  M0 is the
// destination base and compiler-reserved: saved and restored inside the statement.  Not counted by the compiler's
// s_waitcnt bookkeeping: loads issued BEFORE it only get waited for more conservatively (vmcnt retires in order);
// the reader waits with an explicit vmcnt(0).
__device__ __forceinline__ void glds16(const float* gsrc, const unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same with scalar base + 32-bit unsigned byte offset per lane (one address VGPR instead of a pair)
__device__ __forceinline__ void glds16o(const float* base, const unsigned e, const unsigned lds_dst) {
  unsigned keep;
  const unsigned boff = e << 2;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(boff), "s"(lds_dst), "s"(base) : "memory");
}
__device__ __forceinline__ void sto4(float* base, const unsigned e, const float4& v) {
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + (e << 2)) = v;
}
// Device-scope (sc1) forms for launches of several timesteps, where a psi value written in one timestep is read in the next by
// another workgroup — the same cells, but wherever the dispatcher put it: write-through stores, loads that no CU's L1 serves
// (tools/streams/xcd_coherence_probe.hip: sc1 stores + sc1 loads read fresh across XCDs also when the reader's L2 holds the
// line's previous contents; plain loads do not).
__device__ __forceinline__ void glds16o_dev(const float* base, const unsigned e, const unsigned lds_dst) {
  unsigned keep;
  const unsigned boff = e << 2;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(boff), "s"(lds_dst), "s"(base) : "memory");
}
template <bool DEV>
__device__ __forceinline__ void glds16o_t(const float* base, const unsigned e, const unsigned lds_dst) {
  if (DEV) glds16o_dev(base, e, lds_dst); else glds16o(base, e, lds_dst);
}
template <bool DEV>
__device__ __forceinline__ void sto4_t(float* base, const unsigned e, const float4& v) {
  if (DEV) sto4_dev(base, e, v); else sto4(base, e, v);
}
__device__ __forceinline__ void cpml_row4_reg(float4& d, float4& ps, float b, float c, float ik);
template <bool DEV>
__device__ __forceinline__ void cpml_row4_t(float4& d, float* base, const unsigned o, float b, float c, float ik) {
  if (!DEV) { cpml_row4(d, base, o, b, c, ik); return; }
  float4 ps = ldb4_dev(dev_buf(base), o << 2, 0u);
  cpml_row4_reg(d, ps, b, c, ik);
  sto4_dev(base, o, ps);
}
__device__ __forceinline__ void cpml_row4_reg(float4& d, float4& ps, float b, float c, float ik) {
  ps.x = __builtin_fmaf(b, ps.x, c * d.x);
  ps.y = __builtin_fmaf(b, ps.y, c * d.y);
  ps.z = __builtin_fmaf(b, ps.z, c * d.z);
  ps.w = __builtin_fmaf(b, ps.w, c * d.w);
  d.x = __builtin_fmaf(ik, d.x, ps.x);
  d.y = __builtin_fmaf(ik, d.y, ps.y);
  d.z = __builtin_fmaf(ik, d.z, ps.z);
  d.w = __builtin_fmaf(ik, d.w, ps.w);
}

// x-directed layers: per-cell coefficients, psi stored in rows of xrs floats per (k, j) (psi_off_x).  Both x ranges of the internal layout start on a
// 4-cell boundary (fdtd_set_cpml), so a thread's four cells are ONE aligned float4 of psi and of each coefficient table
// (cells drawn into the aligned range from outside the real layer, and pad cells, carry identity coefficients).
__device__ __forceinline__ void cpml_x4_apply(float4& d, float4& ps, const float4& b, const float4& c, const float4& ik) {
  ps.x = __builtin_fmaf(b.x, ps.x, c.x * d.x);
  ps.y = __builtin_fmaf(b.y, ps.y, c.y * d.y);
  ps.z = __builtin_fmaf(b.z, ps.z, c.z * d.z);
  ps.w = __builtin_fmaf(b.w, ps.w, c.w * d.w);
  d.x = __builtin_fmaf(ik.x, d.x, ps.x);
  d.y = __builtin_fmaf(ik.y, d.y, ps.y);
  d.z = __builtin_fmaf(ik.z, d.z, ps.z);
  d.w = __builtin_fmaf(ik.w, d.w, ps.w);
}
__device__ __forceinline__ void cpml_x4(const DevParams& p, int eh, int i0, int rowslot, float4& da, float* psia,
                                        float4& db, float* psib) {
  const int sx = i0 < p.pml_lo[0] ? p.xlo_off + i0 : p.xhi_off + i0 - p.pml_hi[0];
  const float4 b = ld4(p.cp[0][eh][0] + i0), c = ld4(p.cp[0][eh][1] + i0), ik = ld4(p.cp[0][eh][2] + i0);
  float* qa = psia + rowslot + sx;
  float* qb = psib + rowslot + sx;
  float4 pa = ld4(qa), pb = ld4(qb);
  cpml_x4_apply(da, pa, b, c, ik);
  cpml_x4_apply(db, pb, b, c, ik);
  st4(qa, pa);
  st4(qb, pb);
}

// Issue the LDS-DMA loads of the x- and z-directed psi pairs of this thread's cell group (E: eh = 0, H: eh = 1).
// ox / oz: element offsets into the psi arrays, -1 when the group is outside that layer.
__device__ __forceinline__ int psi_off_x(const DevParams& p, const int k, const int j, const int i0) {   // -1: outside the x layers
  if (!(i0 < p.pml_lo[0] || i0 >= p.pml_hi[0])) return -1;   // both bounds are multiples of 4: all four cells or none
#ifdef FDTD_XPSI_EMULATE_2D   // TIMING EXPERIMENT ONLY (wrong fields): the x-psi work in 2 of every 7 waves, wave-uniformly — what a 2-D wave footprint
  // (16 groups x 4 rows) would leave of it on 400 x 400 x 80, without changing the memory access pattern: an upper bound of that idea's gain
  if (((blockIdx.x * (FDTD_BLOCK / 64) + (threadIdx.x >> 6)) % 7u) >= 2u) return -1;
#endif
  // product and sum kept apart (the empty asm): fused, hipcc emits v_mad_u64_u32 with an UNDEFINED high half of the 64-bit
  // addend, picks a register a field load is still in flight to for it, and guards that false dependency with
  // s_waitcnt vmcnt(0) — in front of the staged psi loads, i.e. every wave with an x-layer lane (all of them) waited for its
  // field loads before it issued its psi loads
  // layout (fdtd_set_cpml): one 128-byte-aligned row of xrs floats per (k, j), holding the HIGH part of row j - 1 and then the
  // LOW part of row j — the lanes of a wave that straddles a row boundary (end of row j - 1, start of row j) read and write
  // ONE cache line of each psi array, and no other wave touches it
  int row = k * p.xplane + j * p.xrs;
  asm volatile("" : "+v"(row));
  return row + (i0 < p.pml_lo[0] ? p.xlo_off + i0 : p.xhi_off + i0 - p.pml_hi[0]);
}
__device__ __forceinline__ int psi_off_z(const DevParams& p, const int k, const int j, const int i0) {
  const int sz = pml_slot(p, 2, k);
  return sz < 0 ? -1 : (sz * p.ny + j) * p.P + i0;
}
__device__ __forceinline__ int psi_off_y(const DevParams& p, const int k, const int j, const int i0) {
  const int sy = pml_slot(p, 1, j);
  return sy < 0 ? -1 : (k * p.nslot[1] + sy) * p.P + i0;
}
// Slots 2,3 take the z-directed pair in the z-layer planes (block-uniform: a block lies in one plane) and the y-directed
// pair elsewhere; where both layers meet (edges, corners) the y pair is loaded directly, after the differences, as before.
template <bool DEV = false>
__device__ __forceinline__ void psi_stage_issue(const DevParams& p, float* const (&psi)[3][2], const unsigned stage, const bool valid,
                                                const int k, const int j, const int i0) {
  if (!valid) return;
  const int ox = psi_off_x(p, k, j, i0);
  if (ox >= 0) {
    glds16o_t<DEV>(psi[1][1], (unsigned)ox, stage);
    glds16o_t<DEV>(psi[2][0], (unsigned)ox, stage + 1024u);
  }
  const int oz = psi_off_z(p, k, j, i0);   // >= 0 for the whole block or for none of it (a block lies in one plane)
  if (oz >= 0) {
    glds16o_t<DEV>(psi[0][1], (unsigned)oz, stage + 2048u);
    glds16o_t<DEV>(psi[1][0], (unsigned)oz, stage + 3072u);
  } else {                                  // outside the z layers the same two slots take the y-directed pair
    const int oy = psi_off_y(p, k, j, i0);
    if (oy >= 0) {
      glds16o_t<DEV>(psi[0][0], (unsigned)oy, stage + 2048u);
      glds16o_t<DEV>(psi[2][1], (unsigned)oy, stage + 3072u);
    }
  }
}
// ... and use them: z pair on (dzA, dzB) = the two differences taken along z, x pair on (dxA, dxB) = along x.
template <bool DEV = false>
__device__ __forceinline__ void psi_stage_apply(const DevParams& p, float* const (&psi)[3][2], const int eh, const float4* s_psi, const float* s_xc,
                                                const bool xc_lds, const int k, const int j, const int i0,
                                                float4& dzA, float4& dzB, float4& dxA, float4& dxB, float4& dyA, float4& dyB) {
  const int ox = psi_off_x(p, k, j, i0), oz = psi_off_z(p, k, j, i0);   // recomputed, not carried: registers
  const float4* mine = s_psi + (threadIdx.x >> 6) * (PSI_SLOTS * 64) + (threadIdx.x & 63u);
  // one psi array at a time, fenced: the scheduler would otherwise keep all four staged values and the coefficient vectors
  // live at once, and this kernel has no registers to spare
  if (oz >= 0) {   // z-layer plane (uniform over the block): z pair from the stage, y pair (edges) by direct loads
    {
      const float b = p.cp[2][eh][0][k], c = p.cp[2][eh][1][k], ik = p.cp[2][eh][2][k];
      float4 ps = mine[2 * 64];
      cpml_row4_reg(dzA, ps, b, c, ik);
      sto4_t<DEV>(psi[0][1], (unsigned)oz, ps);
      __builtin_amdgcn_sched_barrier(0);
      ps = mine[3 * 64];
      cpml_row4_reg(dzB, ps, b, c, ik);
      sto4_t<DEV>(psi[1][0], (unsigned)oz, ps);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int oy = psi_off_y(p, k, j, i0);
    if (oy >= 0) {
      const float b = p.cp[1][eh][0][j], c = p.cp[1][eh][1][j], ik = p.cp[1][eh][2][j];
      cpml_row4_t<DEV>(dyA, psi[0][0], (unsigned)oy, b, c, ik);
      __builtin_amdgcn_sched_barrier(0);
      cpml_row4_t<DEV>(dyB, psi[2][1], (unsigned)oy, b, c, ik);
    }
  } else {         // elsewhere the stage holds the y pair
    const int oy = psi_off_y(p, k, j, i0);
    if (oy >= 0) {
      const float b = p.cp[1][eh][0][j], c = p.cp[1][eh][1][j], ik = p.cp[1][eh][2][j];
      float4 ps = mine[2 * 64];
      cpml_row4_reg(dyA, ps, b, c, ik);
      sto4_t<DEV>(psi[0][0], (unsigned)oy, ps);
      __builtin_amdgcn_sched_barrier(0);
      ps = mine[3 * 64];
      cpml_row4_reg(dyB, ps, b, c, ik);
      sto4_t<DEV>(psi[2][1], (unsigned)oy, ps);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (ox >= 0) {
    float4 b, c, ik;
    if (xc_lds) {
      const int sx = i0 < p.pml_lo[0] ? i0 : i0 - p.pml_hi[0] + p.pml_hi_slot[0];
      b = *reinterpret_cast<const float4*>(s_xc + sx);
      c = *reinterpret_cast<const float4*>(s_xc + XC_MAX + sx);
      ik = *reinterpret_cast<const float4*>(s_xc + 2 * XC_MAX + sx);
    } else {
      b = ld4(p.cp[0][eh][0] + i0); c = ld4(p.cp[0][eh][1] + i0); ik = ld4(p.cp[0][eh][2] + i0);
    }
    float4 ps = mine[0];
    cpml_x4_apply(dxA, ps, b, c, ik);
    sto4_t<DEV>(psi[1][1], (unsigned)ox, ps);
    __builtin_amdgcn_sched_barrier(0);
    ps = mine[64];
    cpml_x4_apply(dxB, ps, b, c, ik);
    sto4_t<DEV>(psi[2][0], (unsigned)ox, ps);
  }
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ float4 upd4(const float4& ca, const float4& f, const float4& cb, const float4& d1,
                                       const float4& d2) {
  return make_float4(__builtin_fmaf(ca.x, f.x, cb.x * (d1.x - d2.x)), __builtin_fmaf(ca.y, f.y, cb.y * (d1.y - d2.y)),
                     __builtin_fmaf(ca.z, f.z, cb.z * (d1.z - d2.z)), __builtin_fmaf(ca.w, f.w, cb.w * (d1.w - d2.w)));
}


}  // namespace
