// farfield.hip — near-to-far-field radiation integral on the GPU (K7).
// Replaces the integral inside nf2ff.CalcNF2FF(...) (antenna_sim/solver_fdtd_openems_fixed.py:296;
// per-phi loops at solver_fdtd_openems_microstrip_3d.py:224-225, _multi_3d.py:620-621): one launch
// covers the whole theta x phi grid.  One block per direction, fp64 accumulation, tree reduction.
#include <math.h>

#include "fdtd_ctx.h"

namespace {
constexpr int FF_BLOCK = 256;

__global__ __launch_bounds__(FF_BLOCK) void k_farfield(const int npts, const double* __restrict__ pos,
                                                       const double* __restrict__ Js, const double* __restrict__ Ms,
                                                       const double kw, const double* __restrict__ theta,
                                                       const double* __restrict__ phi, double* __restrict__ Eth,
                                                       double* __restrict__ Eph) {
  __shared__ double red[12][FF_BLOCK];
  const int a = blockIdx.x;
  double st, ct, sp, cp;
  sincos(theta[a], &st, &ct);
  sincos(phi[a], &sp, &cp);
  const double rx = st * cp, ry = st * sp, rz = ct;
  double acc[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) acc[q] = 0.0;
  for (int p = threadIdx.x; p < npts; p += FF_BLOCK) {
    const double ph = kw * (rx * pos[3 * p] + ry * pos[3 * p + 1] + rz * pos[3 * p + 2]);
    double ci, cr;
    sincos(ph, &ci, &cr);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const double jr = Js[(3 * p + n) * 2], ji = Js[(3 * p + n) * 2 + 1];
      const double mr = Ms[(3 * p + n) * 2], mi = Ms[(3 * p + n) * 2 + 1];
      acc[2 * n] += jr * cr - ji * ci;
      acc[2 * n + 1] += jr * ci + ji * cr;
      acc[6 + 2 * n] += mr * cr - mi * ci;
      acc[6 + 2 * n + 1] += mr * ci + mi * cr;
    }
  }
#pragma unroll
  for (int q = 0; q < 12; ++q) red[q][threadIdx.x] = acc[q];
  __syncthreads();
  for (int w = FF_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int q = 0; q < 12; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double eta0 = 376.730313668;
    const double fac = kw / (4.0 * M_PI);
    double Nth[2], Nph[2], Lth[2], Lph[2];
    for (int z = 0; z < 2; ++z) {
      const double Nx = red[0 + z][0], Ny = red[2 + z][0], Nz = red[4 + z][0];
      const double Lx = red[6 + z][0], Ly = red[8 + z][0], Lz = red[10 + z][0];
      Nth[z] = Nx * ct * cp + Ny * ct * sp - Nz * st;
      Nph[z] = -Nx * sp + Ny * cp;
      Lth[z] = Lx * ct * cp + Ly * ct * sp - Lz * st;
      Lph[z] = -Lx * sp + Ly * cp;
    }
    const double ar = Lph[0] + eta0 * Nth[0], ai = Lph[1] + eta0 * Nth[1];
    Eth[2 * a] = fac * ai; Eth[2 * a + 1] = -fac * ar;
    const double br = Lth[0] - eta0 * Nph[0], bi = Lth[1] - eta0 * Nph[1];
    Eph[2 * a] = -fac * bi; Eph[2 * a + 1] = fac * br;
  }
}
}  // namespace

extern "C" int fdtd_farfield(int device, int npts, const double* pos, const double* Js, const double* Ms, double kw,
                             int nang, const double* theta, const double* phi, double* Eth, double* Eph) {
  if (npts < 0 || nang < 0 || !pos || !Js || !Ms || !theta || !phi || !Eth || !Eph)
    return fdtd_fail(nullptr, FDTD_E_ARG, "bad farfield argument");
  if (nang == 0) return FDTD_OK;
  HIPCK(nullptr, hipSetDevice(device));
  double *d_pos = nullptr, *d_J = nullptr, *d_M = nullptr, *d_th = nullptr, *d_ph = nullptr, *d_eth = nullptr, *d_eph = nullptr;
  const size_t np = (size_t)(npts > 0 ? npts : 1);
  int rc = FDTD_OK;
#define FF(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess && rc == FDTD_OK) rc = fdtd_fail(nullptr, FDTD_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
  FF(hipMalloc(&d_pos, np * 3 * sizeof(double)));
  FF(hipMalloc(&d_J, np * 6 * sizeof(double)));
  FF(hipMalloc(&d_M, np * 6 * sizeof(double)));
  FF(hipMalloc(&d_th, nang * sizeof(double)));
  FF(hipMalloc(&d_ph, nang * sizeof(double)));
  FF(hipMalloc(&d_eth, nang * 2 * sizeof(double)));
  FF(hipMalloc(&d_eph, nang * 2 * sizeof(double)));
  if (rc == FDTD_OK) {
    if (npts > 0) {
      FF(hipMemcpy(d_pos, pos, (size_t)npts * 3 * sizeof(double), hipMemcpyHostToDevice));
      FF(hipMemcpy(d_J, Js, (size_t)npts * 6 * sizeof(double), hipMemcpyHostToDevice));
      FF(hipMemcpy(d_M, Ms, (size_t)npts * 6 * sizeof(double), hipMemcpyHostToDevice));
    }
    FF(hipMemcpy(d_th, theta, nang * sizeof(double), hipMemcpyHostToDevice));
    FF(hipMemcpy(d_ph, phi, nang * sizeof(double), hipMemcpyHostToDevice));
    if (rc == FDTD_OK) {
      hipLaunchKernelGGL(k_farfield, dim3(nang), dim3(FF_BLOCK), 0, 0, npts, d_pos, d_J, d_M, kw, d_th, d_ph, d_eth, d_eph);
      FF(hipGetLastError());
      FF(hipDeviceSynchronize());
      FF(hipMemcpy(Eth, d_eth, nang * 2 * sizeof(double), hipMemcpyDeviceToHost));
      FF(hipMemcpy(Eph, d_eph, nang * 2 * sizeof(double), hipMemcpyDeviceToHost));
    }
  }
#undef FF
  hipFree(d_pos); hipFree(d_J); hipFree(d_M); hipFree(d_th); hipFree(d_ph); hipFree(d_eth); hipFree(d_eph);
  return rc;
}
