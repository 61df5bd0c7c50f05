// Feasibility probe for a one-pass (fused E+H) z-marching kernel fed by an LDS-DMA ring: only the DATA
// MOVEMENT of that structure, with token arithmetic — what fraction of the chip's streaming rate does it reach?
//
// A block of 16 x 16 threads owns a 15 x 15 patch of 4-cell groups in the xy plane (one feeder row / column, as the real
// kernel needs for E'(i+1), E'(j+1)) and marches through a chunk of z planes.  Per plane: all six field components of
// the 16 x 16 tile arrive in an LDS ring slot by global_load_lds_dwordx4 issued DEPTH planes ahead (no register
// destination: loads in flight cost no VGPRs), one raw s_barrier, every thread reads its own six float4 and two
// neighbours' from LDS, adds them up, and the owners store six float4 to the second buffer set.  Counted vmcnt waits
// keep the prefetched planes in flight across the barrier.
//
//   hipcc -O3 --offload-arch=gfx950 tools/streams/march_ring_probe.hip -o march_ring_probe && ./march_ring_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int MX = 16, MY = 16, MT = MX * MY;

__device__ __forceinline__ void glds16o(const float* base, const unsigned boff, const unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(boff), "s"(lds_dst), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier_raw() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
struct Arrays { const float* in[6]; float* out[6]; };

// RING slots, prefetch DEPTH = RING - 2 planes ahead
template <int RING>
__global__ __launch_bounds__(MT) void k_march(const Arrays a, const int P, const int ny, const int nz, const int kc, const int ntx, const int nty) {
  extern __shared__ float4 ring[];   // [RING][6][MT]
  constexpr int DEPTH = RING - 2;
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned q8 = nb >> 3, r8 = nb & 7u, xcd = b & 7u;
  const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int bx = (int)(v % (unsigned)ntx), by = (int)((v / (unsigned)ntx) % (unsigned)nty), bz = (int)(v / ((unsigned)ntx * (unsigned)nty));
  const int tx = threadIdx.x & (MX - 1), ty = threadIdx.x >> 4;
  const int i0 = min((bx * (MX - 1) + tx) * 4, P - 4), j = min(by * (MY - 1) + ty, ny - 1);
  const bool owner = tx < MX - 1 && ty < MY - 1 && (bx * (MX - 1) + tx) * 4 < P && by * (MY - 1) + ty < ny;
  const int kb = bz * kc, ke = min(kb + kc, nz);
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ring + (threadIdx.x >> 6) * 1024u);
  const unsigned plane = (unsigned)P * ny;
  const unsigned rowoff = ((unsigned)j * P + i0) * 4u;   // byte offset inside a plane
  auto issue = [&](const int kslot, const int k) {   // plane k -> ring slot kslot % RING
    const unsigned slot = wave_lds + (unsigned)(kslot % RING) * (6u * MT * 16u);
    const unsigned boff = (unsigned)k * plane * 4u + rowoff;
#pragma unroll
    for (int c = 0; c < 6; ++c) glds16o(a.in[c], boff, slot + c * (MT * 16u));
  };
  for (int d = 0; d < DEPTH; ++d) issue(kb + d, min(kb + d, ke - 1));   // prologue (always DEPTH x 6 loads: the counts below rely on it)
  for (int k = kb; k < ke; ++k) {
    // loads of plane k are the oldest outstanding: everything younger = (DEPTH-1) x 6 loads + min(k-kb, DEPTH-1... ) stores
    const int it = k - kb;
    if (DEPTH == 1) { if (it == 0) wait_vm<0>(); else wait_vm<6>(); }
    else { if (it == 0) wait_vm<6>(); else if (it == 1) wait_vm<12>(); else wait_vm<18>(); }
    barrier_raw();
    issue(k + DEPTH, min(k + DEPTH, ke - 1));
    const float4* s = ring + (size_t)(k % RING) * (6 * MT);
    const float4* sp = ring + (size_t)((k + RING - 1) % RING) * (6 * MT);   // previous plane (k-1 neighbour)
    float4 o[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float4 own = s[c * MT + threadIdx.x];
      const float4 jm = s[c * MT + (ty > 0 ? threadIdx.x - MX : threadIdx.x)];
      const float4 km = (it > 0 ? sp : s)[((c + 1) % 6) * MT + threadIdx.x];
      const float im = s[c * MT + (tx > 0 ? threadIdx.x - 1 : threadIdx.x)].w;
      o[c] = make_float4(own.x + jm.x - km.x + im, own.y + jm.y - km.y + own.x, own.z + jm.z - km.z + own.y, own.w + jm.w - km.w + own.z);
    }
    const unsigned eo = (unsigned)k * plane + (unsigned)j * P + i0;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      float4* q = reinterpret_cast<float4*>(a.out[c] + eo);
      if (owner) *q = o[c];
    }
    // keep the store count per wave-iteration uniform: a wave whose lanes are all non-owners must still issue six stores
    // (never the case for 16 x 16 tiles: every wave holds owner rows), so the counted waits above stay exact
  }
  wait_vm<0>();
}

// The same data movement with REGISTER prefetch instead of the LDS-DMA ring: plain loads of plane k+2 issued at the top of
// iteration k into a rotating set of three register buffers (hipcc counts its own vmcnt, so the older planes' loads are
// waited for while the younger stay in flight); LDS only for the E'-like exchange between neighbours (3 float4 per
// thread, double-buffered, one barrier per plane), as a real one-pass kernel needs it.  Neighbour operands (row j-1,
// cell i-1) are loaded from global memory like everything else.
struct PlaneRegs { float4 f[6], jm0, jm1; float im0, im1; };
__device__ __forceinline__ void load_plane(const Arrays& a, const unsigned eo, const int P, PlaneRegs& r) {
#pragma unroll
  for (int c = 0; c < 6; ++c) r.f[c] = *reinterpret_cast<const float4*>(a.in[c] + eo);
  r.jm0 = *reinterpret_cast<const float4*>(a.in[0] + eo - P);
  r.jm1 = *reinterpret_cast<const float4*>(a.in[2] + eo - P);
  r.im0 = a.in[1][eo - 1];
  r.im1 = a.in[2][eo - 1];
}
__global__ __launch_bounds__(MT) void k_march_reg(const Arrays a, const int P, const int ny, const int nz, const int kc, const int ntx, const int nty) {
  __shared__ float4 s_e[2][3][MT];
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned q8 = nb >> 3, r8 = nb & 7u, xcd = b & 7u;
  const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int bx = (int)(v % (unsigned)ntx), by = (int)((v / (unsigned)ntx) % (unsigned)nty), bz = (int)(v / ((unsigned)ntx * (unsigned)nty));
  const int tx = threadIdx.x & (MX - 1), ty = threadIdx.x >> 4;
  const int i0 = min((bx * (MX - 1) + tx) * 4, P - 4), j = max(1, min(by * (MY - 1) + ty, ny - 1));
  const bool owner = tx < MX - 1 && ty < MY - 1 && (bx * (MX - 1) + tx) * 4 < P && by * (MY - 1) + ty < ny;
  const int kb = bz * kc, ke = min(kb + kc, nz);
  const unsigned plane = (unsigned)P * ny, row = (unsigned)j * P + max(i0, 4);
  PlaneRegs r[3];
  load_plane(a, (unsigned)kb * plane + row, P, r[0]);
  load_plane(a, (unsigned)min(kb + 1, ke - 1) * plane + row, P, r[1]);
  float4 prev0 = make_float4(0.f, 0.f, 0.f, 0.f), prev1 = prev0;
  for (int k0 = kb; k0 < ke; k0 += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int k = k0 + u;
      if (k < ke) {
        load_plane(a, (unsigned)min(k + 2, ke - 1) * plane + row, P, r[(u + 2) % 3]);
        const PlaneRegs& c = r[u];
        float4 o[6];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          o[q] = make_float4(c.f[q].x + c.jm0.x - prev0.x + c.im0, c.f[q].y + c.jm1.y - prev1.y + c.f[q].x, c.f[q].z + c.f[q + 3].z, c.f[q].w - c.f[q + 3].w + c.im1);
        const int bsel = k & 1;
#pragma unroll
        for (int q = 0; q < 3; ++q) s_e[bsel][q][threadIdx.x] = o[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float4 jp = s_e[bsel][q][ty < MY - 1 ? threadIdx.x + MX : threadIdx.x];
          const float ip = s_e[bsel][(q + 1) % 3][tx < MX - 1 ? threadIdx.x + 1 : threadIdx.x].x;
          o[q + 3] = make_float4(c.f[q + 3].x + o[q].x - jp.x, c.f[q + 3].y + o[q].y - jp.y + ip, c.f[q + 3].z + jp.z, c.f[q + 3].w + jp.w);
        }
        prev0 = c.f[0]; prev1 = c.f[1];
        const unsigned eo = (unsigned)k * plane + row;
        if (owner) {
#pragma unroll
          for (int q = 0; q < 6; ++q) *reinterpret_cast<float4*>(a.out[q] + eo) = o[q];
        }
      }
    }
  }
}

static void run_reg(int nx, int ny, int nz, int kc, const char* tag) {
  const int P = (nx + 3) / 4 * 4, P4 = P / 4;
  const size_t n = (size_t)P * ny * nz;
  Arrays a;
  std::vector<float*> all;
  for (int c = 0; c < 6; ++c) {
    float *i, *o;
    hipMalloc(&i, n * 4 + 4096); hipMemset(i, 0, n * 4 + 4096);
    hipMalloc(&o, n * 4 + 4096); hipMemset(o, 0, n * 4 + 4096);
    a.in[c] = i; a.out[c] = o; all.push_back(i); all.push_back(o);
  }
  const int ntx = (P4 + MX - 2) / (MX - 1), nty = (ny + MY - 2) / (MY - 1), ntz = (nz + kc - 1) / kc;
  const unsigned grid = (unsigned)(ntx * nty * ntz);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(k_march_reg, dim3(grid), dim3(MT), 0, 0, a, P, ny, nz, kc, ntx, nty); for (int c = 0; c < 6; ++c) { const float* t = a.in[c]; a.in[c] = a.out[c]; a.out[c] = (float*)t; } }
  hipEventRecord(e0);
  const int reps = 30;
  for (int w = 0; w < reps; ++w) { hipLaunchKernelGGL(k_march_reg, dim3(grid), dim3(MT), 0, 0, a, P, ny, nz, kc, ntx, nty); for (int c = 0; c < 6; ++c) { const float* t = a.in[c]; a.in[c] = a.out[c]; a.out[c] = (float*)t; } }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / reps * 1e3, cells = (double)nx * ny * nz;
  printf("%s %dx%dx%d REGISTER prefetch kc %2d grid %5u: %8.1f us/step  %6.1f Gcells/s  useful %.2f TB/s (48 B/cell)  [%s]\n",
         tag, nx, ny, nz, kc, grid, us, cells / us / 1e3, cells * 48.0 / us / 1e6, hipGetErrorString(hipGetLastError()));
  for (auto p : all) hipFree(p);
}

template <int RING>
static void run(int nx, int ny, int nz, int kc, const char* tag) {
  const int P = (nx + 3) / 4 * 4, P4 = P / 4;
  const size_t n = (size_t)P * ny * nz;
  Arrays a;
  std::vector<float*> all;
  for (int c = 0; c < 6; ++c) {
    float *i, *o;
    hipMalloc(&i, n * 4 + 4096); hipMemset(i, 0, n * 4 + 4096);
    hipMalloc(&o, n * 4 + 4096); hipMemset(o, 0, n * 4 + 4096);
    a.in[c] = i; a.out[c] = o; all.push_back(i); all.push_back(o);
  }
  const int ntx = (P4 + MX - 2) / (MX - 1), nty = (ny + MY - 2) / (MY - 1), ntz = (nz + kc - 1) / kc;
  const unsigned grid = (unsigned)(ntx * nty * ntz);
  const size_t lds = (size_t)RING * 6 * MT * 16;
  hipFuncSetAttribute((const void*)k_march<RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL((k_march<RING>), dim3(grid), dim3(MT), lds, 0, a, P, ny, nz, kc, ntx, nty); for (int c = 0; c < 6; ++c) { const float* t = a.in[c]; a.in[c] = a.out[c]; a.out[c] = (float*)t; } }
  hipEventRecord(e0);
  const int reps = 30;
  for (int w = 0; w < reps; ++w) { hipLaunchKernelGGL((k_march<RING>), dim3(grid), dim3(MT), lds, 0, a, P, ny, nz, kc, ntx, nty); for (int c = 0; c < 6; ++c) { const float* t = a.in[c]; a.in[c] = a.out[c]; a.out[c] = (float*)t; } }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / reps * 1e3, cells = (double)nx * ny * nz;
  printf("%s %dx%dx%d ring %d (%zu KB LDS) kc %2d grid %5u: %8.1f us/step  %6.1f Gcells/s  useful %.2f TB/s (48 B/cell)  moved ~%.2f TB/s (with 16/15 overlap)  [%s]\n",
         tag, nx, ny, nz, RING, lds >> 10, kc, grid, us, cells / us / 1e3, cells * 48.0 / us / 1e6, cells * (24.0 * 256 / 225 + 24.0) / us / 1e6,
         hipGetErrorString(hipGetLastError()));
  for (auto p : all) hipFree(p);
}

int main() {
  for (int kc : {10, 20, 60}) run_reg(300, 300, 60, kc, "NS");
  for (int kc : {10, 20, 80}) run_reg(400, 400, 80, kc, "C3");
  for (int kc : {20, 60}) run_reg(800, 800, 120, kc, "C5");
  for (int kc : {10, 20, 60}) { run<3>(300, 300, 60, kc, "NS"); run<4>(300, 300, 60, kc, "NS"); }
  for (int kc : {10, 20, 40, 80}) { run<3>(400, 400, 80, kc, "C3"); run<4>(400, 400, 80, kc, "C3"); }
  for (int kc : {20, 60, 120}) { run<3>(800, 800, 120, kc, "C5"); run<4>(800, 800, 120, kc, "C5"); }
  return 0;
}
