// Coherence probe: can a workgroup on XCD A read, INSIDE one launch, what a workgroup on XCD B has just written into a line
// that A's L2 already holds from an earlier read?  (The question behind "several timesteps per launch": in a second step
// every L2 holds the first step's copies of lines other XCDs have rewritten since — DESIGN.md §7.)
//
// Pairs of one-wave workgroups on DIFFERENT XCDs (checked with HW_REG_XCC_ID) ping-pong through the same `region` bytes for
// `rounds` rounds.  Round r: the producer fills the region with the value r (whole 128-byte lines, 16-byte stores), drains
// its stores, publishes flag = r; the consumer polls the flag, reads the region, counts every word != r, acknowledges.  The
// consumer has read the same lines in round r - 1 (and before round 1 with plain AND sc1 loads), so its XCD's L2 holds
// the previous contents when the producer overwrites them.  Variants:
//   store: 0 plain + agent release fence before the flag   1 sc1 (write-through)   2 sc0 sc1 (system scope)
//   load : 0 plain   1 sc1   2 sc0 sc1   3 plain behind an agent ACQUIRE fence (buffer_inv sc1)
// Prints stale words per variant (0 = every round read fresh).  Memory: hipMalloc (coarse-grained) or, with argv[1] = fine,
// hipExtMallocWithFlags(finegrained).
// hipcc -O3 --offload-arch=gfx950 -o xcd_probe xcd_coherence_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int ST>
__device__ __forceinline__ void st16(unsigned* q, unsigned v) {
  const v4u t = {v, v, v, v};
  if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(q), "v"(t) : "memory");
  if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(q), "v"(t) : "memory");
  if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(q), "v"(t) : "memory");
}
template <int LD>
__device__ __forceinline__ v4u ld16(const unsigned* q) {
  v4u r;
  if (LD == 0 || LD == 3) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(q) : "memory");
  if (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(q) : "memory");
  if (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(q) : "memory");
  return r;
}
__device__ __forceinline__ unsigned poll(const unsigned* f, unsigned want, unsigned long long limit, int* timeout) {
  const unsigned long long t0 = wall_clock64();
  unsigned v;
  while ((v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want) {
    __builtin_amdgcn_s_sleep(2);
    if ((unsigned long long)wall_clock64() - t0 > limit || __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      __hip_atomic_store(timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  return v;
}

// blocks 2p (producer) and 2p + 1 (consumer): consecutive block indices land on different XCDs (round-robin)
template <int ST, int LD>
__global__ __launch_bounds__(64) void k_pingpong(unsigned* data, unsigned* flags, unsigned* acks, unsigned* xcc, unsigned* stale,
                                                 int* timeout, const int words, const int rounds) {
  const int pair = blockIdx.x >> 1, role = blockIdx.x & 1, lane = threadIdx.x;
  unsigned* region = data + (size_t)pair * words;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (lane == 0) xcc[blockIdx.x] = id & 0xFu;
  const unsigned long long limit = 200000000ull;   // 2 s
  unsigned bad = 0;
  if (role == 1) {   // consumer: put the region into this XCD's L2 (and this CU's L1) before anything is written
    for (int w = lane * 4; w < words; w += 256) { v4u a = ld16<0>(region + w); v4u b = ld16<1>(region + w); bad += (a.x | b.x) & 0u; }
  }
  for (int r = 1; r <= rounds; ++r) {
    if (__hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // somebody gave up: everybody leaves
    if (role == 0) {
      if (lane == 0 && r > 1) poll(acks + pair * 32, (unsigned)(r - 1), limit, timeout);   // the consumer has read round r - 1
      __builtin_amdgcn_s_barrier();
      for (int w = lane * 4; w < words; w += 256) st16<ST>(region + w, (unsigned)r);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (ST == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      if (lane == 0) __hip_atomic_store(flags + pair * 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) poll(flags + pair * 32, (unsigned)r, limit, timeout);
      __builtin_amdgcn_s_barrier();
      if (LD == 3) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      for (int w = lane * 4; w < words; w += 256) {
        const v4u v = ld16<LD>(region + w);
        bad += (v.x != (unsigned)r) + (v.y != (unsigned)r) + (v.z != (unsigned)r) + (v.w != (unsigned)r);
      }
      __builtin_amdgcn_s_barrier();
      if (lane == 0) __hip_atomic_store(acks + pair * 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (bad) atomicAdd(stale, bad);
}

template <int ST, int LD>
static void run(const char* what, unsigned* data, unsigned* ctl, int pairs, int words, int rounds) {
  unsigned *flags = ctl, *acks = ctl + pairs * 32, *xcc = acks + pairs * 32, *stale = xcc + 2 * pairs;
  int* timeout = (int*)(stale + 1);
  hipMemset(data, 0, (size_t)pairs * words * sizeof(unsigned));
  hipMemset(ctl, 0, (size_t)(pairs * 64 + 2 * pairs + 2) * sizeof(unsigned));
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_pingpong<ST, LD>), dim3(2 * pairs), dim3(64), 0, 0, data, flags, acks, xcc, stale, timeout, words, rounds);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h(2 * pairs + 2);
  hipMemcpy(h.data(), xcc, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  int cross = 0;
  for (int p = 0; p < pairs; ++p) cross += h[2 * p] != h[2 * p + 1];
  printf("%-34s stale words %10u of %llu  (timeouts %u, %d of %d pairs on two XCDs, %.2f us per round trip)\n", what, h[2 * pairs],
         (unsigned long long)pairs * words * rounds, h[2 * pairs + 1], cross, pairs, ms * 1e3 / rounds);
}

int main(int argc, char** argv) {
  const bool fine = argc > 1 && !strcmp(argv[1], "fine");
  const int pairs = 64, words = 4096 /* 16 KiB = 128 lines per pair */, rounds = 2000;
  unsigned *data = nullptr, *ctl = nullptr;
  const size_t bytes = (size_t)pairs * words * sizeof(unsigned);
  if (fine) { if (hipExtMallocWithFlags((void**)&data, bytes, hipDeviceMallocFinegrained) != hipSuccess) { printf("no fine-grained memory\n"); return 1; } }
  else hipMalloc(&data, bytes);
  hipMalloc(&ctl, (size_t)(pairs * 64 + 2 * pairs + 2) * sizeof(unsigned));
  printf("# xcd_coherence_probe: %d producer/consumer pairs, %d bytes per pair, %d rounds, %s memory\n", pairs, words * 4, rounds,
         fine ? "fine-grained" : "coarse-grained (hipMalloc)");
  run<1, 1>("sc1 stores, sc1 loads", data, ctl, pairs, words, rounds);
  run<1, 0>("sc1 stores, plain loads", data, ctl, pairs, words, rounds);
  run<1, 3>("sc1 stores, acquire + plain loads", data, ctl, pairs, words, rounds);
  run<0, 1>("plain stores + release, sc1 loads", data, ctl, pairs, words, rounds);
  run<0, 3>("plain + release, acquire + plain", data, ctl, pairs, words, rounds);
  run<0, 0>("plain + release, plain loads", data, ctl, pairs, words, rounds);
  run<2, 2>("sc0 sc1 stores, sc0 sc1 loads", data, ctl, pairs, words, rounds);
  run<2, 1>("sc0 sc1 stores, sc1 loads", data, ctl, pairs, words, rounds);
  hipFree(data); hipFree(ctl);
  return 0;
}
