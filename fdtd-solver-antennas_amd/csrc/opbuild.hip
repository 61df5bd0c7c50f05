// opbuild.hip — the operator set-up phase of FDTD.Run on the device.
//
// The reference's external engine builds its update coefficients inside FDTD.Run(...) before the first
// timestep (antenna_sim/solver_fdtd_openems_fixed.py:280).  Here: per-cell materials, PEC
// edge flags and cell sizes go to the GPU once; one thread per grid node evaluates the three EC edge
// coefficients (vv, m) in float64 with exactly the operation order of the host formulation
// (ecoperator.build_operator — the spec, and what the oracle restates in C), the distinct (vv, m) pairs are
// collected in an LDS/global hash set, numbered, and the class bytes (one per cell when the scene has <= 256
// distinct triples, else one per edge) are written straight into the layout the update kernels read.  More than
// 256 pairs: the twelve raw coefficient arrays are expanded on the device instead.
#include <algorithm>
#include <vector>

#include "kernel_common.hpp"

namespace {

constexpr unsigned long long EMPTY = ~0ull;   // (vv bits = 0xFFFFFFFF is a NaN: never a coefficient)
constexpr int TAB = 1024;                     // hash slots (<= 256 live keys are accepted)
constexpr int MAXK = 256;

struct OpGeom {
  int nx, ny, nz, k0, nk, kc_lo;
  const double* d[3];
  const double* eps;       // cells of planes kc_lo.. : [..][ny-1][nx-1]
  const double* kap;
  const uint8_t* pec[3];   // [nk][ny][nx] each, local planes
  double dt, eps0;
};

// (vv, m) of the three edges at node (i, j, k0 + k) -> vm[c][(k*ny + j)*nx + i]
__global__ __launch_bounds__(256) void k_op_vm(const OpGeom g, float2* __restrict__ vm) {
  const size_t n = (size_t)g.nk * g.ny * g.nx;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int i = (int)(idx % g.nx), j = (int)((idx / g.nx) % g.ny), k = (int)(idx / ((size_t)g.nx * g.ny));
  const int pos[3] = {i, j, g.k0 + k};
  const int nn[3] = {g.nx, g.ny, g.nz};
  const size_t crow = (size_t)(g.nx - 1), cplane = (size_t)(g.nx - 1) * (g.ny - 1);
  for (int c = 0; c < 3; ++c) {
    const int a1 = (c + 1) % 3, a2 = (c + 2) % 3;
    // PEC, the non-existent last edge, tangential edges on the six outer planes (PEC backing)
    const bool dead = g.pec[c][idx] != 0 || pos[c] == nn[c] - 1 || pos[a1] == 0 || pos[a1] == nn[a1] - 1 ||
                      pos[a2] == 0 || pos[a2] == nn[a2] - 1;
    float2 out = make_float2(0.f, 0.f);
    if (!dead) {
      // area-weighted mean of eps_r / kappa over the four cells around the edge, summed in the host order:
      // a1 offset outer (-1, 0), a2 offset inner (-1, 0); weight = d[a1] * d[a2] of the cell
      double ne = 0.0, nk_ = 0.0, den = 0.0;
      for (int o1 = -1; o1 <= 0; ++o1)
        for (int o2 = -1; o2 <= 0; ++o2) {
          int ci[3];
          ci[c] = pos[c]; ci[a1] = pos[a1] + o1; ci[a2] = pos[a2] + o2;
          const double w = g.d[a1][ci[a1]] * g.d[a2][ci[a2]];
          const size_t q = (size_t)(ci[2] - g.kc_lo) * cplane + (size_t)ci[1] * crow + ci[0];
          ne = ne + g.eps[q] * w;
          nk_ = nk_ + g.kap[q] * w;
          den = den + w;
        }
      const double eps_e = (ne / den) * g.eps0;
      const double kap_e = nk_ / den;
      const double x = ((0.5 * g.dt) * kap_e) / eps_e;
      out.x = (float)((1.0 - x) / (1.0 + x));
      out.y = (float)(g.dt / (eps_e * (1.0 + x)));
    }
    vm[(size_t)c * n + idx] = out;
  }
}

__global__ void k_op_override(const int n, const long long* __restrict__ idx, const int8_t* __restrict__ comp,
                              const float* __restrict__ vv, const float* __restrict__ m, const size_t ncell,
                              float2* __restrict__ vm) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) vm[(size_t)comp[q] * ncell + (size_t)idx[q]] = make_float2(vv[q], m[q]);
}

__device__ __forceinline__ unsigned hash64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
  return (unsigned)k;
}

// insert into an open-addressing set; returns false when the set is full
template <typename CAS>
__device__ __forceinline__ bool set_insert(unsigned long long* tab, const unsigned long long key, CAS cas, int* fresh) {
  unsigned s = hash64(key) & (TAB - 1);
  for (int probe = 0; probe < TAB; ++probe) {
    const unsigned long long cur = tab[s];
    if (cur == key) return true;
    if (cur == EMPTY) {
      const unsigned long long old = cas(&tab[s], EMPTY, key);
      if (old == EMPTY) { if (fresh) atomicAdd(fresh, 1); return true; }
      if (old == key) return true;
    }
    s = (s + 1) & (TAB - 1);
  }
  return false;
}

// Distinct keys of a stream -> global set.  mode 0: (vv, m) pairs of vm[0 .. n); mode 1: class triples of
// cls[0..ncell) | cls[ncell..] << 8 | cls[2 ncell..] << 16.  Block-local LDS set first, then one global insert per
// distinct local key.  *state: [0] = number of keys in the global set, [1] = overflow flag.
__global__ __launch_bounds__(256) void k_op_collect(const int mode, const float2* __restrict__ vm, const uint8_t* __restrict__ cls,
                                                    const size_t n, const size_t ncell, unsigned long long* __restrict__ gtab,
                                                    int* __restrict__ state) {
  __shared__ unsigned long long s_tab[TAB];
  __shared__ int s_cnt;
  for (int q = threadIdx.x; q < TAB; q += 256) s_tab[q] = EMPTY;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  auto lds_cas = [](unsigned long long* a, unsigned long long cmp, unsigned long long val) { return atomicCAS(a, cmp, val); };
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    unsigned long long key;
    if (mode == 0) {
      const float2 v = vm[q];
      key = ((unsigned long long)__float_as_uint(v.x) << 32) | __float_as_uint(v.y);
    } else {
      key = (unsigned long long)cls[q] | ((unsigned long long)cls[ncell + q] << 8) | ((unsigned long long)cls[2 * ncell + q] << 16);
    }
    if (s_cnt > MAXK * 2 || !set_insert(s_tab, key, lds_cas, &s_cnt)) { state[1] = 1; break; }
  }
  __syncthreads();
  auto g_cas = [](unsigned long long* a, unsigned long long cmp, unsigned long long val) { return atomicCAS(a, cmp, val); };
  for (int q = threadIdx.x; q < TAB; q += 256) {
    const unsigned long long key = s_tab[q];
    if (key == EMPTY) continue;
    if (state[0] > MAXK * 2 || !set_insert(gtab, key, g_cas, &state[0])) { state[1] = 1; break; }
  }
}

__device__ __forceinline__ int find_sorted(const unsigned long long* keys, const int n, const unsigned long long key) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// class byte of every edge: position of its (vv, m) pair in the sorted key list
__global__ __launch_bounds__(256) void k_op_classify(const float2* __restrict__ vm, const size_t n, const unsigned long long* __restrict__ keys,
                                                     const int nkeys, uint8_t* __restrict__ cls) {
  __shared__ unsigned long long s_keys[MAXK];
  for (int q = threadIdx.x; q < nkeys; q += 256) s_keys[q] = keys[q];
  __syncthreads();
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
    const float2 v = vm[q];
    cls[q] = (uint8_t)find_sorted(s_keys, nkeys, ((unsigned long long)__float_as_uint(v.x) << 32) | __float_as_uint(v.y));
  }
}

// dense class bytes -> the kernels' layout (pitch P).  packed: one byte per cell = position of the triple.
__global__ __launch_bounds__(256) void k_op_write_classes(const int packed, const uint8_t* __restrict__ cls, const int nx, const int P,
                                                          const size_t rows, const size_t ncell, const size_t nloc,
                                                          const unsigned long long* __restrict__ keys, const int nkeys,
                                                          uint8_t* __restrict__ ecls) {
  __shared__ unsigned long long s_keys[MAXK];
  if (packed) for (int q = threadIdx.x; q < nkeys; q += 256) s_keys[q] = keys[q];
  __syncthreads();
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < ncell; q += (size_t)gridDim.x * 256) {
    const size_t row = q / nx;
    const int i = (int)(q - row * nx);
    const size_t dst = row * P + i;
    if (packed) {
      const unsigned long long key = (unsigned long long)cls[q] | ((unsigned long long)cls[ncell + q] << 8) | ((unsigned long long)cls[2 * ncell + q] << 16);
      ecls[dst] = (uint8_t)find_sorted(s_keys, nkeys, key);
    } else {
      ecls[dst] = cls[q];
      ecls[nloc + dst] = cls[ncell + q];
      ecls[2 * nloc + dst] = cls[2 * ncell + q];
    }
  }
}

// raw form: the four coefficient arrays, expanded with the float32 association of the C ABI
__global__ __launch_bounds__(256) void k_op_write_raw(const DevParams p, const float2* __restrict__ vm, const int nx, const size_t ncell,
                                                      float* __restrict__ vv, float* __restrict__ vi, float* __restrict__ ii,
                                                      float* __restrict__ iv) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= ncell) return;
  const int i = (int)(q % nx), j = (int)((q / nx) % p.ny), k = (int)(q / ((size_t)nx * p.ny));
  const size_t dst = (size_t)k * p.plane + (size_t)j * p.P + i;
  for (int c = 0; c < 3; ++c) {
    const float2 v = vm[(size_t)c * ncell + q];
    const float eyz = p.emet[c][1][j] * p.emet[c][2][k];
    const float hyz = p.hmet[c][1][j] * p.hmet[c][2][k];
    vv[(size_t)c * p.nloc + dst] = v.x;
    vi[(size_t)c * p.nloc + dst] = v.y * (p.emet[c][0][i] * eyz);
    ii[(size_t)c * p.nloc + dst] = 1.0f;
    iv[(size_t)c * p.nloc + dst] = p.hmet[c][0][i] * hyz;
  }
}

// any operator form -> dense raw coefficient `which` (0 vv, 1 vi, 2 ii, 3 iv) of all three components
__global__ __launch_bounds__(256) void k_op_expand(const DevParams p, const int form, const int which, const int nx, const size_t ncell,
                                                   float* __restrict__ out) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= ncell) return;
  const int i = (int)(q % nx), j = (int)((q / nx) % p.ny), k = (int)(q / ((size_t)nx * p.ny));
  const size_t src = (size_t)k * p.plane + (size_t)j * p.P + i;
  for (int c = 0; c < 3; ++c) {
    float r;
    if (form == 3) {
      const float* a = which == 0 ? p.vv : which == 1 ? p.vi : which == 2 ? p.ii : p.iv;
      r = a[(size_t)c * p.nloc + src];
    } else {
      const float2 l = form == 2 ? p.lut[3 * p.ecls[src] + c] : p.lut[p.ecls[(size_t)c * p.nloc + src]];
      if (which == 0) r = l.x;
      else if (which == 1) r = l.y * (p.emet[c][0][i] * (p.emet[c][1][j] * p.emet[c][2][k]));
      else if (which == 2) r = 1.0f;
      else r = p.hmet[c][0][i] * (p.hmet[c][1][j] * p.hmet[c][2][k]);
    }
    out[(size_t)c * ncell + q] = r;
  }
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)); }
  hipError_t from(const T* src, size_t n) {
    hipError_t e = alloc(n);
    return e != hipSuccess || n == 0 ? e : hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice);
  }
};

// distinct keys of the stream, sorted; returns false when there are more than MAXK
int collect_sorted(fdtd_ctx* c, int mode, const float2* vm, const uint8_t* cls, size_t n, size_t ncell, std::vector<unsigned long long>& keys,
                   bool& fits) {
  DevBuf<unsigned long long> tab;
  DevBuf<int> state;
  HIPCK(c, tab.alloc(TAB));
  HIPCK(c, state.alloc(2));
  HIPCK(c, hipMemsetAsync(tab.p, 0xFF, TAB * sizeof(unsigned long long), c->stream));
  HIPCK(c, hipMemsetAsync(state.p, 0, 2 * sizeof(int), c->stream));
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_op_collect, dim3(blocks), dim3(256), 0, c->stream, mode, vm, cls, n, ncell, tab.p, state.p);
  std::vector<unsigned long long> h(TAB);
  int st[2];
  HIPCK(c, hipMemcpyAsync(h.data(), tab.p, TAB * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipMemcpyAsync(st, state.p, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  keys.clear();
  for (auto k : h) if (k != EMPTY) keys.push_back(k);
  std::sort(keys.begin(), keys.end());
  fits = !st[1] && keys.size() <= (size_t)MAXK && !keys.empty();
  return FDTD_OK;
}

}  // namespace

extern "C" {

int fdtd_build_operator(fdtd_ctx* c, const double* dx, const double* dy, const double* dz, const double* eps_r, const double* kappa,
                        const uint8_t* pec, double eps0, int n_over, const int64_t* over_edge, const int8_t* over_comp,
                        const float* over_vv, const float* over_m, const float* emet, const float* hmet, int prefer_classes) {
  if (!c || !dx || !dy || !dz || !eps_r || !kappa || !pec || !emet || !hmet || n_over < 0 ||
      (n_over > 0 && (!over_edge || !over_comp || !over_vv || !over_m)))
    return fdtd_fail(c, FDTD_E_ARG, "null argument");
  HIPCK(c, hipSetDevice(c->d.device));
  const int nx = c->d.nx, ny = c->d.ny, nz = c->d.nz, k0 = c->d.k0, nk = c->d.nk;
  if (nx < 3 || ny < 3 || nz < 3) return fdtd_fail(c, FDTD_E_ARG, "grid too small for an operator");
  const size_t ncell = (size_t)nk * ny * nx;
  OpGeom g{};
  g.nx = nx; g.ny = ny; g.nz = nz; g.k0 = k0; g.nk = nk; g.dt = c->d.dt; g.eps0 = eps0;
  // cells the slab's edges touch: z-cell planes k0-1 .. k0+nk-1, clipped to the grid
  const int kc_lo = std::max(k0 - 1, 0), kc_hi = std::min(k0 + nk - 1, nz - 2);
  g.kc_lo = kc_lo;
  const size_t cplane = (size_t)(nx - 1) * (ny - 1), ncells_up = cplane * (size_t)(kc_hi - kc_lo + 1);
  DevBuf<double> d_d[3], d_eps, d_kap;
  DevBuf<uint8_t> d_pec;
  DevBuf<float2> vm;
  const double* hd[3] = {dx, dy, dz};
  const int nn[3] = {nx, ny, nz};
  for (int a = 0; a < 3; ++a) { HIPCK(c, d_d[a].from(hd[a], nn[a])); g.d[a] = d_d[a].p; }
  HIPCK(c, d_eps.from(eps_r + (size_t)kc_lo * cplane, ncells_up));
  HIPCK(c, d_kap.from(kappa + (size_t)kc_lo * cplane, ncells_up));
  g.eps = d_eps.p; g.kap = d_kap.p;
  HIPCK(c, d_pec.alloc(3 * ncell));
  const size_t gplane = (size_t)ny * nx;
  for (int comp = 0; comp < 3; ++comp) {
    HIPCK(c, hipMemcpy(d_pec.p + comp * ncell, pec + ((size_t)comp * nz + k0) * gplane, ncell, hipMemcpyHostToDevice));
    g.pec[comp] = d_pec.p + comp * ncell;
  }
  HIPCK(c, vm.alloc(3 * ncell));
  hipLaunchKernelGGL(k_op_vm, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, g, vm.p);
  // host-fixed edges (lumped elements) inside this slab
  {
    std::vector<long long> li;
    std::vector<int8_t> lc;
    std::vector<float> lv, lm;
    for (int q = 0; q < n_over; ++q) {
      const int64_t e = over_edge[q];
      if (e < 0 || e >= (int64_t)nz * (int64_t)gplane || over_comp[q] < 0 || over_comp[q] > 2) return fdtd_fail(c, FDTD_E_ARG, "override %d out of range", q);
      const int64_t kg = e / (int64_t)gplane;
      if (kg < k0 || kg >= k0 + nk) continue;
      li.push_back((long long)(e - (int64_t)k0 * (int64_t)gplane)); lc.push_back(over_comp[q]); lv.push_back(over_vv[q]); lm.push_back(over_m[q]);
    }
    if (!li.empty()) {
      DevBuf<long long> di; DevBuf<int8_t> dc; DevBuf<float> dv, dm;
      HIPCK(c, di.from(li.data(), li.size())); HIPCK(c, dc.from(lc.data(), lc.size()));
      HIPCK(c, dv.from(lv.data(), lv.size())); HIPCK(c, dm.from(lm.data(), lm.size()));
      hipLaunchKernelGGL(k_op_override, dim3((unsigned)((li.size() + 63) / 64)), dim3(64), 0, c->stream, (int)li.size(), di.p, dc.p, dv.p, dm.p, ncell, vm.p);
      HIPCK(c, hipStreamSynchronize(c->stream));
    }
  }
  { int r = upload_metric_tables(c, emet, hmet); if (r) return r; }

  std::vector<unsigned long long> keys;
  bool fits = false;
  if (prefer_classes) { int r = collect_sorted(c, 0, vm.p, nullptr, 3 * ncell, ncell, keys, fits); if (r) return r; }
  if (fits) {
    DevBuf<unsigned long long> d_keys;
    DevBuf<uint8_t> cls;
    HIPCK(c, d_keys.from(keys.data(), keys.size()));
    HIPCK(c, cls.alloc(3 * ncell));
    const unsigned blocks = (unsigned)std::min<size_t>((3 * ncell + 255) / 256, 4096);
    hipLaunchKernelGGL(k_op_classify, dim3(blocks), dim3(256), 0, c->stream, vm.p, 3 * ncell, d_keys.p, (int)keys.size(), cls.p);
    std::vector<unsigned long long> triples;
    bool packed = false;
    { int r = collect_sorted(c, 1, nullptr, cls.p, ncell, ncell, triples, packed); if (r) return r; }
    if (!packed && 3 * c->nloc >= ((size_t)1 << 31))
      return fdtd_fail(c, FDTD_E_UNSUPPORTED, "per-edge class operator: slab exceeds 2^31 / 3 elements per component (use more z-slabs)");
    const size_t n_alloc = 3 * c->nloc + 64;   // same slack as fdtd_set_operator_classes
    if (!c->ecls) HIPCK(c, hipMalloc(&c->ecls, n_alloc));
    HIPCK(c, hipMemsetAsync(c->ecls, 0, n_alloc, c->stream));
    DevBuf<unsigned long long> d_tri;
    HIPCK(c, d_tri.from(triples.data(), packed ? triples.size() : 0));
    const unsigned wb = (unsigned)std::min<size_t>((ncell + 255) / 256, 4096);
    hipLaunchKernelGGL(k_op_write_classes, dim3(wb), dim3(256), 0, c->stream, packed ? 1 : 0, cls.p, nx, c->P, (size_t)nk * ny, ncell, c->nloc,
                       d_tri.p, packed ? (int)triples.size() : 0, c->ecls);
    std::vector<float2> lut(packed ? 768 : 256, make_float2(0.f, 0.f));
    auto pair_of = [&](int cl) {
      union { unsigned u; float f; } hi, lo;
      hi.u = (unsigned)(keys[cl] >> 32); lo.u = (unsigned)keys[cl];
      return make_float2(hi.f, lo.f);
    };
    if (packed) {
      for (size_t t = 0; t < triples.size(); ++t)
        for (int comp = 0; comp < 3; ++comp) lut[3 * t + comp] = pair_of((int)((triples[t] >> (8 * comp)) & 0xFF));
      c->p.lut_n = (int)(3 * triples.size());
    } else {
      for (size_t q = 0; q < keys.size(); ++q) lut[q] = pair_of((int)q);
      c->p.lut_n = (int)keys.size();
    }
    hipFree(c->lut); c->lut = nullptr;
    HIPCK(c, hipMalloc(&c->lut, lut.size() * sizeof(float2)));
    HIPCK(c, hipMemcpy(c->lut, lut.data(), lut.size() * sizeof(float2), hipMemcpyHostToDevice));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->packed_op = packed;
    c->p.ecls = c->ecls; c->p.lut = c->lut;
    c->have_op = true; c->raw_op = false;
    c->op_nclasses = (int)keys.size();
  } else {
    if (3 * c->nloc >= ((size_t)1 << 31))
      return fdtd_fail(c, FDTD_E_UNSUPPORTED, "raw operator: slab exceeds 2^31 / 3 elements per component (use more z-slabs)");
    const size_t bytes = 3 * c->nloc * sizeof(float);
    float** dst[4] = {&c->vv, &c->vi, &c->ii, &c->iv};
    for (int n = 0; n < 4; ++n) {
      if (!*dst[n]) HIPCK(c, hipMalloc(dst[n], bytes));
      HIPCK(c, hipMemsetAsync(*dst[n], 0, bytes, c->stream));
    }
    hipLaunchKernelGGL(k_op_write_raw, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, c->p, vm.p, nx, ncell, c->vv, c->vi, c->ii, c->iv);
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->p.vv = c->vv; c->p.vi = c->vi; c->p.ii = c->ii; c->p.iv = c->iv;
    c->have_op = true; c->raw_op = true; c->packed_op = false;
    c->op_nclasses = 0;
  }
  HIPCK(c, hipGetLastError());
  return FDTD_OK;
}

int fdtd_operator_form(fdtd_ctx* c, int* form, int* nclasses) {
  if (!c || !form) return FDTD_E_ARG;
  *form = !c->have_op ? 0 : c->raw_op ? 3 : c->packed_op ? 2 : 1;
  if (nclasses) *nclasses = c->op_nclasses;
  return FDTD_OK;
}

int fdtd_get_operator(fdtd_ctx* c, float* vv, float* vi, float* ii, float* iv) {
  if (!c || !vv || !vi || !ii || !iv) return fdtd_fail(c, FDTD_E_ARG, "null argument");
  if (!c->have_op) return fdtd_fail(c, FDTD_E_STATE, "operator not set");
  HIPCK(c, hipSetDevice(c->d.device));
  const size_t ncell = (size_t)c->d.nk * c->d.ny * c->d.nx;
  DevBuf<float> tmp;
  HIPCK(c, tmp.alloc(3 * ncell));
  float* out[4] = {vv, vi, ii, iv};
  const int form = c->raw_op ? 3 : c->packed_op ? 2 : 1;
  for (int which = 0; which < 4; ++which) {
    hipLaunchKernelGGL(k_op_expand, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, c->p, form, which, c->d.nx, ncell, tmp.p);
    HIPCK(c, hipMemcpyAsync(out[which], tmp.p, 3 * ncell * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
  }
  return FDTD_OK;
}

}  // extern "C"
