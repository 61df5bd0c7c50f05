// Streaming ceiling for the access mix of the update kernels: NR read arrays + NW written arrays, float4 per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NR, int NW>
__global__ __launch_bounds__(256) void k_stream(float4* const* rd, float4* const* wr, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < NR; ++a) { const float4 v = rd[a][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
  for (int a = 0; a < NW; ++a) wr[a][i] = acc;
}
template <int NR, int NW>
void run(size_t n4, const char* tag) {
  std::vector<float4*> h(NR + NW);
  for (auto& p : h) { hipMalloc(&p, n4 * sizeof(float4)); hipMemset(p, 0, n4 * sizeof(float4)); }
  float4** d; hipMalloc(&d, (NR + NW) * sizeof(float4*)); hipMemcpy(d, h.data(), (NR + NW) * sizeof(float4*), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grid = (unsigned)((n4 + 255) / 256);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_stream<NR, NW>), dim3(grid), dim3(256), 0, 0, d, d + NR, n4);
  hipEventRecord(a);
  const int reps = 50;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_stream<NR, NW>), dim3(grid), dim3(256), 0, 0, d, d + NR, n4);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)(NR + NW) * n4 * 16.0 * reps;
  printf("%s  %zu Mfloat4/array  %dR+%dW: %.1f us/launch  %.2f TB/s\n", tag, n4 >> 20, NR, NW, ms / reps * 1e3, bytes / (ms * 1e-3) / 1e12);
  for (auto p : h) hipFree(p); hipFree(d);
}
int main() {
  for (size_t cells : {(size_t)5400000, (size_t)12800000, (size_t)76800000}) {
    const size_t n4 = cells / 4;
    printf("--- %zu cells per array ---\n", cells);
    run<1, 1>(n4, "copy   ");
    run<2, 1>(n4, "2R+1W  ");
    run<6, 3>(n4, "6R+3W  ");
    run<9, 3>(n4, "9R+3W  ");
  }
  return 0;
}
