// farfield.hip — near-to-far-field radiation integral on the GPU (K7).
// Replaces the integral inside nf2ff.CalcNF2FF(...) (antenna_sim/solver_fdtd_openems_fixed.py:296;
// per-phi loops at solver_fdtd_openems_microstrip_3d.py:224-225, _multi_3d.py:620-621): one launch
// covers the whole theta x phi grid.  FF_DIRS directions per block, fp64 accumulation, tree reduction.
#include <math.h>

#include "fdtd_ctx.h"

namespace {
constexpr int FF_BLOCK = 256;
// Directions per block: every direction needs every quadrature point, so a block that serves ONE direction streams the
// whole point set (120 bytes per point) for 12 complex sums — 800x800x120 with 91 x 73 directions: 6643 x 180 MB = 1.2 TB
// through L2 / the Infinity Cache, 0.16 s, memory-bound.  FF_DIRS directions per block read each point once for all of
// them.  Per direction the summation order is unchanged (thread t takes points t, t + 256, ...; then the tree), so the
// results are the same bits as with one direction per block.
constexpr int FF_DIRS = 4;

__global__ __launch_bounds__(FF_BLOCK) void k_farfield(const int npts, const double* __restrict__ pos,
                                                       const double* __restrict__ Js, const double* __restrict__ Ms,
                                                       const double kw, const int nang, const double* __restrict__ theta,
                                                       const double* __restrict__ phi, double* __restrict__ Eth,
                                                       double* __restrict__ Eph) {
  __shared__ double red[12][FF_BLOCK];
  const int a0 = blockIdx.x * FF_DIRS;
  double rx[FF_DIRS], ry[FF_DIRS], rz[FF_DIRS];
#pragma unroll
  for (int d = 0; d < FF_DIRS; ++d) {
    const int a = min(a0 + d, nang - 1);
    double st, ct, sp, cp;
    sincos(theta[a], &st, &ct);
    sincos(phi[a], &sp, &cp);
    rx[d] = st * cp; ry[d] = st * sp; rz[d] = ct;
  }
  double acc[FF_DIRS][12];
#pragma unroll
  for (int d = 0; d < FF_DIRS; ++d)
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[d][q] = 0.0;
  for (int p = threadIdx.x; p < npts; p += FF_BLOCK) {
    const double px = pos[3 * p], py = pos[3 * p + 1], pz = pos[3 * p + 2];
    double jm[12];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      jm[4 * n] = Js[(3 * p + n) * 2]; jm[4 * n + 1] = Js[(3 * p + n) * 2 + 1];
      jm[4 * n + 2] = Ms[(3 * p + n) * 2]; jm[4 * n + 3] = Ms[(3 * p + n) * 2 + 1];
    }
#pragma unroll
    for (int d = 0; d < FF_DIRS; ++d) {
      const double ph = kw * (rx[d] * px + ry[d] * py + rz[d] * pz);
      double ci, cr;
      sincos(ph, &ci, &cr);
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const double jr = jm[4 * n], ji = jm[4 * n + 1], mr = jm[4 * n + 2], mi = jm[4 * n + 3];
        acc[d][2 * n] += jr * cr - ji * ci;
        acc[d][2 * n + 1] += jr * ci + ji * cr;
        acc[d][6 + 2 * n] += mr * cr - mi * ci;
        acc[d][6 + 2 * n + 1] += mr * ci + mi * cr;
      }
    }
  }
#pragma unroll
  for (int d = 0; d < FF_DIRS; ++d) {
    const int a = a0 + d;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 12; ++q) red[q][threadIdx.x] = acc[d][q];
    __syncthreads();
    for (int w = FF_BLOCK / 2; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w)
        for (int q = 0; q < 12; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0 && a < nang) {
      double st, ct, sp, cp;
      sincos(theta[a], &st, &ct);
      sincos(phi[a], &sp, &cp);
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
      hipLaunchKernelGGL(k_farfield, dim3((nang + FF_DIRS - 1) / FF_DIRS), dim3(FF_BLOCK), 0, 0, npts, d_pos, d_J, d_M, kw, nang, d_th, d_ph, d_eth, d_eph);
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
