// Data-movement probe: what would ONE launch per timestep buy, with the E sweep running LAG planes ahead of the H sweep so
// that H finds E's output (and the I planes E just read) in the Infinity Cache instead of in HBM?
//
// Streaming stand-ins of the two update kernels on six in-place arrays V[3], I[3] ((nz+2) planes of ny*P floats):
//   E-like block: reads V[0..2], I[0..2] at its cells + Ix, Iy one plane below, writes V[0..2]
//   H-like block: reads V[0..2], I[0..2] at its cells + Vx, Vy one plane above, writes I[0..2]
// Every XCD owns a y-slab (1/8 of the rows) and walks it plane by plane (blockIdx % 8 = XCD group, as the product does).
//   mode 0: two launches per step (E forwards, H backwards) — the product's schedule
//   mode 1: one launch; per XCD the dispatch order is  E(plane kk), H(plane kk - LAG)  for kk = 0 .. nz+LAG-1
//   dep 1 : H blocks of plane k wait (one lane polls, bounded) until every E block of planes k and k+1 has counted itself in
//   sc  1 : E stores V write-through (sc1), H loads V with sc1 (the cross-XCD visibility form of MI355X_MICROARCH.md)
// hipcc -O3 --offload-arch=gfx950 -o eh_probe eh_interleave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Arr { float4* V[3]; float4* I[3]; };
typedef float v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld_sc1(const float4* q) {
  v4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r) : "v"(q) : "memory");
  return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ void st_sc1(float4* q, const float4& v) {
  const v4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(q), "v"(t) : "memory");
}
__device__ __forceinline__ float4 add4(float4 a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; return a; }

template <bool SC>
__device__ __forceinline__ void body_E(const Arr& a, size_t idx, size_t plane4) {
  float4 v0 = a.V[0][idx], v1 = a.V[1][idx], v2 = a.V[2][idx];
  const float4 i0 = a.I[0][idx], i1 = a.I[1][idx], i2 = a.I[2][idx];
  const float4 i0m = a.I[0][idx - plane4], i1m = a.I[1][idx - plane4];
  v0 = add4(v0, add4(i1, i1m)); v1 = add4(v1, add4(i0, i0m)); v2 = add4(v2, i2);
  if (SC) { st_sc1(a.V[0] + idx, v0); st_sc1(a.V[1] + idx, v1); st_sc1(a.V[2] + idx, v2); }
  else { a.V[0][idx] = v0; a.V[1][idx] = v1; a.V[2][idx] = v2; }
}
template <bool SC>
__device__ __forceinline__ void body_H(const Arr& a, size_t idx, size_t plane4) {
  float4 v0, v1, v2, v0p, v1p;
  if (SC) {
    v0 = ld_sc1(a.V[0] + idx); v1 = ld_sc1(a.V[1] + idx); v2 = ld_sc1(a.V[2] + idx);
    v0p = ld_sc1(a.V[0] + idx + plane4); v1p = ld_sc1(a.V[1] + idx + plane4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    v0 = a.V[0][idx]; v1 = a.V[1][idx]; v2 = a.V[2][idx]; v0p = a.V[0][idx + plane4]; v1p = a.V[1][idx + plane4];
  }
  float4 i0 = a.I[0][idx], i1 = a.I[1][idx], i2 = a.I[2][idx];
  i0 = add4(i0, add4(v1, v1p)); i1 = add4(i1, add4(v0, v0p)); i2 = add4(i2, v2);
  a.I[0][idx] = i0; a.I[1][idx] = i1; a.I[2][idx] = i2;
}

// two-launch schedule: XCD x walks its slab plane by plane, forwards (E) or backwards (H)
template <int WHICH>
__global__ __launch_bounds__(256) void k_sweep(const Arr a, const int m, const int nz, const size_t plane4, const int rev) {
  const unsigned b = blockIdx.x, x = b & 7u;
  unsigned pos = b >> 3;
  if (rev) pos = (unsigned)(m * nz) - 1u - pos;
  const unsigned k = pos / (unsigned)m, r = pos - k * (unsigned)m;
  const size_t in_plane = (size_t)(x * m + r) * 256 + threadIdx.x;
  if (in_plane >= plane4) return;
  const size_t idx = (size_t)(k + 1) * plane4 + in_plane;
  if (WHICH == 0) body_E<false>(a, idx, plane4); else body_H<false>(a, idx, plane4);
}

// one launch: per XCD  E(kk), H(kk - lag)
template <bool DEP, bool SC>
__global__ __launch_bounds__(256) void k_fused(const Arr a, const int m, const int nz, const size_t plane4, const int lag,
                                               unsigned* cnt, const unsigned target, int* err) {
  const unsigned b = blockIdx.x, x = b & 7u, pos = b >> 3;
  const unsigned grp = pos / (unsigned)(2 * m), w = pos - grp * (unsigned)(2 * m);
  const bool isH = w >= (unsigned)m;
  const unsigned r = isH ? w - (unsigned)m : w;
  const int k = isH ? (int)grp - lag : (int)grp;
  if (k < 0 || k >= nz) return;
  const size_t in_plane = (size_t)(x * m + r) * 256 + threadIdx.x;
  const size_t idx = (size_t)(k + 1) * plane4 + in_plane;
  if (!isH) {
    if (in_plane < plane4) body_E<SC>(a, idx, plane4);
    if (DEP) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    if (DEP) {
      if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        for (int q = 0; q < 2; ++q) {
          const int kq = k + q;
          if (kq >= nz) break;
          while (__hip_atomic_load(cnt + kq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if ((unsigned long long)wall_clock64() - t0 > 200000000ull) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // 2 s at 100 MHz
          }
        }
      }
      __syncthreads();
    }
    if (in_plane < plane4) body_H<SC>(a, idx, plane4);
  }
}

// one launch, per-block flags: an E block publishes flag[plane][y-block] = launch number after its write-through stores have been
// acknowledged; an H block loads its I values first (they do not depend on this launch), then three lanes poll the flags of
// the three E blocks it reads from (own cells, next y-block, plane above), then the V loads (sc1) follow.
template <bool SC>
__global__ __launch_bounds__(256) void k_fused_flags(const Arr a, const int m, const int nz, const size_t plane4, const int lag,
                                                     unsigned* flag, const unsigned target, int* err) {
  extern __shared__ float ballast[];   // (occupancy ballast of the EH_PROBE_OCC sweep; touched so that it is not optimised away)
  if (threadIdx.x == 1023) ballast[0] = 0.f;
  const unsigned b = blockIdx.x, x = b & 7u, pos = b >> 3;
  const unsigned grp = pos / (unsigned)(2 * m), w = pos - grp * (unsigned)(2 * m);
  const bool isH = w >= (unsigned)m;
  const unsigned r = isH ? w - (unsigned)m : w;
  const int k = isH ? (int)grp - lag : (int)grp;
  if (k < 0 || k >= nz) return;
  const unsigned yb = x * m + r, nyb = 8u * m;
  const size_t in_plane = (size_t)yb * 256 + threadIdx.x;
  const size_t idx = (size_t)(k + 1) * plane4 + in_plane;
  const bool valid = in_plane < plane4;
  if (!isH) {
    if (valid) body_E<SC>(a, idx, plane4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag + (size_t)k * nyb + yb, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    float4 i0 = make_float4(0, 0, 0, 0), i1 = i0, i2 = i0;
    if (valid) { i0 = a.I[0][idx]; i1 = a.I[1][idx]; i2 = a.I[2][idx]; }
    if (threadIdx.x < 3) {
      const unsigned t = threadIdx.x;
      const int kq = k + (t == 2 ? 1 : 0);
      const unsigned yq = yb + (t == 1 ? 1u : 0u);
      if (kq < nz && yq < nyb) {
        const unsigned* f = flag + (size_t)kq * nyb + yq;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
          if ((unsigned long long)wall_clock64() - t0 > 200000000ull) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
      }
    }
    __syncthreads();
    if (valid) {
      float4 v0, v1, v2, v0p, v1p;
      if (SC) {
        v0 = ld_sc1(a.V[0] + idx); v1 = ld_sc1(a.V[1] + idx); v2 = ld_sc1(a.V[2] + idx);
        v0p = ld_sc1(a.V[0] + idx + plane4); v1p = ld_sc1(a.V[1] + idx + plane4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        v0 = a.V[0][idx]; v1 = a.V[1][idx]; v2 = a.V[2][idx]; v0p = a.V[0][idx + plane4]; v1p = a.V[1][idx + plane4];
      }
      i0 = add4(i0, add4(v1, v1p)); i1 = add4(i1, add4(v0, v0p)); i2 = add4(i2, v2);
      a.I[0][idx] = i0; a.I[1][idx] = i1; a.I[2][idx] = i2;
    }
  }
}

// T timesteps in ONE launch as a space-time wavefront along z (round 3 question: would grids beyond the Infinity Cache gain if
// SEVERAL timesteps shared one pass over HBM?): per XCD and plane group g the dispatch order is
//   E_0(g), H_0(g - lag), E_1(g - 2 lag), H_1(g - 3 lag), ... , H_{T-1}(g - (2T-1) lag)
// — no dependencies (data movement only), every access sc1.  Odd launches walk the planes downwards.
__global__ __launch_bounds__(256) void k_fused_T(const Arr a, const int m, const int nz, const size_t plane4, const int lag, const int T, const int down) {
  const unsigned b = blockIdx.x, x = b & 7u, pos = b >> 3;
  const unsigned per = (unsigned)(2 * T * m);
  const unsigned grp = pos / per, w = pos - grp * per;
  const unsigned role = w / (unsigned)m, r = w - role * (unsigned)m;
  int k = (int)grp - (int)role * lag;
  if (k < 0 || k >= nz) return;
  if (down) k = nz - 1 - k;
  const size_t in_plane = (size_t)(x * m + r) * 256 + threadIdx.x;
  if (in_plane >= plane4) return;
  const size_t idx = (size_t)(k + 1) * plane4 + in_plane;
  if (!(role & 1u)) body_E<true>(a, idx, plane4); else body_H<true>(a, idx, plane4);
}

int main(int argc, char** argv) {
  struct { const char* name; int nx, ny, nz; } grids[] = {{"NS", 300, 300, 60}, {"C3", 400, 400, 80}, {"C4", 512, 512, 128}, {"C5", 800, 800, 120}};
  const int reps = 30;
  for (auto& g : grids) {
    const size_t plane4 = (size_t)g.nx * g.ny / 4;
    const int m = (int)((plane4 + 256 * 8 - 1) / (256 * 8));
    const size_t n4 = plane4 * (g.nz + 2);
    Arr a;
    for (int c = 0; c < 3; ++c) {
      hipMalloc(&a.V[c], n4 * 16); hipMemset(a.V[c], 0, n4 * 16);
      hipMalloc(&a.I[c], n4 * 16); hipMemset(a.I[c], 0, n4 * 16);
    }
    unsigned* cnt; hipMalloc(&cnt, 4096 * 4); hipMemset(cnt, 0, 4096 * 4);
    unsigned* flag; hipMalloc(&flag, (size_t)8 * m * (g.nz + 1) * 4); hipMemset(flag, 0, (size_t)8 * m * (g.nz + 1) * 4);
    int* err; hipMalloc(&err, 4); hipMemset(err, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double cells = (double)g.nx * g.ny * g.nz;
    auto report = [&](const char* tag, float ms) {
      int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      printf("%s %dx%dx%d %-34s %8.1f us/step  %6.1f Gcells/s  %5.2f TB/s (72 B/cell)%s  [%s]\n", g.name, g.nx, g.ny, g.nz, tag,
             ms / reps * 1e3, cells / (ms / reps * 1e-3) / 1e9, cells * 72.0 / (ms / reps * 1e-3) / 1e12, herr ? "  TIMEOUT" : "",
             hipGetErrorString(hipGetLastError()));
      fflush(stdout);
    };
    {   // mode 0
      const unsigned grid = 8u * m * g.nz;
      for (int it = -3; it < reps; ++it) {
        if (it == 0) hipEventRecord(e0);
        hipLaunchKernelGGL(k_sweep<0>, dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, 0);
        hipLaunchKernelGGL(k_sweep<1>, dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, 1);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); report("two launches (E fwd, H back)", ms);
    }
    unsigned launches = 0;
    for (int variant : {0, 3, 4, 5}) {      // 0: no deps, 1: plane counters, 2: plane counters + sc1, 3: no deps + sc1, 4: block flags + sc1, 5: block flags, plain
      for (int lag : {1, 2, 3, 4, 6, 8, 12, 16}) {
        if (lag >= g.nz) continue;
        const unsigned grid = 8u * 2u * m * (g.nz + lag);
        for (int it = -3; it < reps; ++it) {
          if (it == 0) hipEventRecord(e0);
          if (variant != 0 && variant != 3) ++launches;
          const unsigned target = launches * 8u * m;
          if (variant == 0) hipLaunchKernelGGL((k_fused<false, false>), dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, lag, cnt, target, err);
          if (variant == 1) hipLaunchKernelGGL((k_fused<true, false>), dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, lag, cnt, target, err);
          if (variant == 2) hipLaunchKernelGGL((k_fused<true, true>), dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, lag, cnt, target, err);
          if (variant == 3) hipLaunchKernelGGL((k_fused<false, true>), dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, lag, cnt, target, err);
          if (variant == 4) hipLaunchKernelGGL((k_fused_flags<true>), dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, lag, flag, launches, err);
          if (variant == 5) hipLaunchKernelGGL((k_fused_flags<false>), dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, lag, flag, launches, err);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        char tag[64]; snprintf(tag, sizeof tag, "one launch lag %2d %s", lag, variant == 0 ? "no deps" : variant == 1 ? "plane cnt" : variant == 2 ? "plane cnt + sc1" : variant == 3 ? "no deps + sc1" : variant == 4 ? "block flags + sc1" : "block flags, plain (racy)");
        report(tag, ms);
      }
    }
    for (int T : {1, 2, 3, 4}) {      // space-time wavefront, no dependencies
      for (int lag : {2, 4, 6, 8, 12}) {
        if ((2 * T - 1) * lag >= 2 * g.nz) continue;
        const unsigned grid = 8u * 2u * T * m * (g.nz + (2 * T - 1) * lag);
        const int nl = (reps + T - 1) / T;
        for (int it = -2; it < nl; ++it) {
          if (it == 0) hipEventRecord(e0);
          hipLaunchKernelGGL(k_fused_T, dim3(grid), dim3(256), 0, 0, a, m, g.nz, plane4, lag, T, it & 1);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        char tag[64]; snprintf(tag, sizeof tag, "wavefront T=%d lag %2d no deps + sc1", T, lag);
        report(tag, ms * reps / (float)(nl * T));
      }
    }
    // round 3, second question: the block-flag variants again with FEWER workgroups per CU (dynamic LDS as ballast) — a block of the
    // full-occupancy launch lives ~10 us = ~9 plane groups of dispatch, so an H block that follows E by fewer groups waits; with 1 ... 4
    // workgroups per CU a block lives shorter, a small lag may do, and at a small lag H finds E's output in the L2 of its XCD
    if (getenv("EH_PROBE_OCC")) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_flags<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused_flags<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      for (int sc : {0, 1}) for (int occ : {1, 2, 3, 4, 6}) for (int lag : {1, 2, 3, 4, 6, 8}) {
        const unsigned grid = 8u * 2u * m * (g.nz + lag);
        const size_t lds = (size_t)(160 * 1024 / occ - 1024) & ~(size_t)1023;
        for (int it = -3; it < reps; ++it) {
          if (it == 0) hipEventRecord(e0);
          ++launches;
          if (sc) hipLaunchKernelGGL((k_fused_flags<true>), dim3(grid), dim3(256), lds, 0, a, m, g.nz, plane4, lag, flag, launches, err);
          else hipLaunchKernelGGL((k_fused_flags<false>), dim3(grid), dim3(256), lds, 0, a, m, g.nz, plane4, lag, flag, launches, err);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        char tag[64]; snprintf(tag, sizeof tag, "flags %s occ %d lag %2d", sc ? "sc1  " : "plain", occ, lag);
        report(tag, ms);
      }
    }
    for (int c = 0; c < 3; ++c) { hipFree(a.V[c]); hipFree(a.I[c]); }
    hipFree(cnt); hipFree(err); hipFree(flag);
  }
  return 0;
}
