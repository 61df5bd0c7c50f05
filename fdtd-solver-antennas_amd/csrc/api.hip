// api.hip — C ABI of libfdtd_hip.so (include/fdtd_hip.h): context management, uploads, the
// time-stepping loop and the z-slab halo exchange (RCCL over xGMI, overlapped with interior updates).
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <mutex>
#include <chrono>
#include <vector>

#include "kernel_common.hpp"   // fdtd_ctx.h + the device helpers of the mailbox protocol (self-test kernels below)

static thread_local std::string g_err;

int fdtd_fail(fdtd_ctx* c, int code, const char* fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_err = buf;
  return code;
}

#define NCCLCK(c, expr)                                                                      \
  do {                                                                                       \
    ncclResult_t r_ = (expr);                                                                \
    if (r_ != ncclSuccess)                                                                   \
      return fdtd_fail(c, FDTD_E_DEVICE, "%s: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// dense host [rows][nx] -> device [rows][P]
template <typename T>
static hipError_t upload_rows(T* dst, int P, const T* src, int nx, size_t rows) {
  return hipMemcpy2D(dst, (size_t)P * sizeof(T), src, (size_t)nx * sizeof(T), (size_t)nx * sizeof(T), rows, hipMemcpyHostToDevice);
}

template <typename T>
static hipError_t to_device(T** dst, const std::vector<T>& v) {
  hipFree(*dst); *dst = nullptr;
  hipError_t e = hipMalloc(dst, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess) return e;
  if (v.empty()) return hipSuccess;
  return hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

// Per strip-plane source lists (CSR) for a tiling of `tys` rows per strip: a source at (k, j) is listed in the
// strip-plane that holds row j of plane k.
static hipError_t build_source_lists(const fdtd_ctx* c, int tys, int nstrips, int2** d_rng, int** d_ids) {
  const int nk = c->d.nk;
  std::vector<std::vector<int>> lists((size_t)nk * nstrips);
  for (size_t e = 0; e < c->h_src_off.size(); ++e) {
    const int off = c->h_src_off[e];
    const int k = off / c->plane, j = (off - k * c->plane) / c->P;
    lists[(size_t)k * nstrips + j / tys].push_back((int)e);
  }
  std::vector<int2> rng(lists.size());
  std::vector<int> ids;
  int most = 0;
  for (size_t q = 0; q < lists.size(); ++q) most = std::max(most, (int)lists[q].size());
  const_cast<fdtd_ctx*>(c)->src_max_per_strip_plane = most;
  for (size_t q = 0; q < lists.size(); ++q) {
    rng[q].x = (int)ids.size();
    ids.insert(ids.end(), lists[q].begin(), lists[q].end());
    rng[q].y = (int)ids.size();
  }
  hipFree(*d_rng); hipFree(*d_ids); *d_rng = nullptr; *d_ids = nullptr;
  hipError_t err = hipMalloc(d_rng, rng.size() * sizeof(int2));
  if (err != hipSuccess) return err;
  err = hipMemcpy(*d_rng, rng.data(), rng.size() * sizeof(int2), hipMemcpyHostToDevice);
  if (err != hipSuccess) return err;
  err = hipMalloc(d_ids, std::max<size_t>(ids.size(), 1) * sizeof(int));
  if (err != hipSuccess) return err;
  if (!ids.empty()) err = hipMemcpy(*d_ids, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice);
  return err;
}

// metric tables of the class operator: per (E|H, comp): x (padded to P, zeros), y, z — each segment 4-float aligned
int upload_metric_tables(fdtd_ctx* c, const float* emet, const float* hmet) {
  const int nx = c->d.nx, ny = c->d.ny, nk = c->d.nk, P = c->P;
  auto al4 = [](int v) { return (v + 3) / 4 * 4; };
  // +4 floats of zero slack per table
  const int sx_ = P + 4, sy_ = al4(ny) + 4, sz_ = al4(nk) + 4;
  const int seg = sx_ + sy_ + sz_;
  std::vector<float> host((size_t)6 * seg, 0.f);
  const int tl = nx + ny + nk;
  for (int eh = 0; eh < 2; ++eh)
    for (int comp = 0; comp < 3; ++comp) {
      const float* src = (eh ? hmet : emet) + (size_t)comp * tl;
      float* dst = host.data() + (size_t)(eh * 3 + comp) * seg;
      memcpy(dst, src, nx * sizeof(float));
      memcpy(dst + sx_, src + nx, ny * sizeof(float));
      memcpy(dst + sx_ + sy_, src + nx + ny, nk * sizeof(float));
    }
  if (!c->met) HIPCK(c, hipMalloc(&c->met, host.size() * sizeof(float)));
  HIPCK(c, hipMemcpy(c->met, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  for (int comp = 0; comp < 3; ++comp) {
    const float* e = c->met + (size_t)comp * seg;
    const float* h = c->met + (size_t)(3 + comp) * seg;
    c->p.emet[comp][0] = e; c->p.emet[comp][1] = e + sx_; c->p.emet[comp][2] = e + sx_ + sy_;
    c->p.hmet[comp][0] = h; c->p.hmet[comp][1] = h + sx_; c->p.hmet[comp][2] = h + sx_ + sy_;
  }
  return FDTD_OK;
}

// Streams outlive contexts: hipStreamCreate takes 3 - 9 ms on this platform (the reference GUI's default call creates a context per run:
// 12 ms of a 94 ms call went into its two streams).  A SINGLE-SLAB context — nothing it launches ever waits for another context's work —
// steps on one stream per device shared by all such contexts of the process, and has no communication stream; slabs of a decomposed grid
// (their kernels wait for each other's halos: they must be able to run side by side) take their own, and a destroyed context hands its
// (drained) streams to the next one on the same device.
static std::mutex g_stream_mu;
static std::vector<hipStream_t> g_stream_pool[64];
static hipStream_t g_stream_shared[64];
static hipError_t stream_take(int device, hipStream_t* out) {
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto& pool = g_stream_pool[device & 63];
    if (!pool.empty()) { *out = pool.back(); pool.pop_back(); return hipSuccess; }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static hipError_t stream_shared(int device, hipStream_t* out) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  hipStream_t& s = g_stream_shared[device & 63];
  if (!s) { const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking); if (e != hipSuccess) { s = nullptr; return e; } }
  *out = s;
  return hipSuccess;
}
static void stream_give(int device, hipStream_t s) {   // (the caller has synchronised it)
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto& pool = g_stream_pool[device & 63];
    if (pool.size() < 16) { pool.push_back(s); return; }
  }
  hipStreamDestroy(s);
}

extern "C" {

int fdtd_version(void) { return FDTD_ABI_VERSION; }

int fdtd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* fdtd_backend(void) { return "hip:gfx950"; }

const char* fdtd_last_error(const fdtd_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

static int n_axis(const fdtd_ctx* c, int a) { return a == 0 ? c->d.nx : a == 1 ? c->d.ny : c->d.nk; }

int fdtd_create(const fdtd_desc* d, fdtd_ctx** out) {
  if (!d || !out) return fdtd_fail(nullptr, FDTD_E_ARG, "null argument");
  if (d->nx < 2 || d->ny < 2 || d->nz < 2 || d->nk < 1 || d->k0 < 0 || d->k0 + d->nk > d->nz)
    return fdtd_fail(nullptr, FDTD_E_ARG, "bad grid/slab %dx%dx%d k0=%d nk=%d", d->nx, d->ny, d->nz, d->k0, d->nk);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fdtd_fail(nullptr, FDTD_E_DEVICE, "no HIP device visible (libfdtd_hip.so has no CPU fallback)");
  if (d->device < 0 || d->device >= ndev) return fdtd_fail(nullptr, FDTD_E_ARG, "device %d of %d", d->device, ndev);
  const long P = (d->nx + 3) / 4 * 4;
  const long plane = P * d->ny;
  // the update kernels address every array as scalar base + 32-bit BYTE offset
  if (plane * (long)(d->nk + 2) >= (1L << 30)) return fdtd_fail(nullptr, FDTD_E_UNSUPPORTED, "slab exceeds 2^30 elements (4 GiB) per field component: use more z-slabs (several slabs may share one GPU, fdtd_link)");
  fdtd_ctx* c = new (std::nothrow) fdtd_ctx();
  if (!c) return fdtd_fail(nullptr, FDTD_E_NOMEM, "ctx");
  c->d = *d;
  c->P = (int)P; c->plane = (int)plane; c->nloc = (size_t)plane * d->nk;
#define CK(expr)                                                                                        \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      fdtd_fail(nullptr, FDTD_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));                        \
      fdtd_destroy(c);                                                                                  \
      return FDTD_E_DEVICE;                                                                             \
    }                                                                                                   \
  } while (0)
  CK(hipSetDevice(d->device));
  if (d->world <= 1) { CK(stream_shared(d->device, &c->stream)); c->stream_shared = true; }
  else { CK(stream_take(d->device, &c->stream)); CK(stream_take(d->device, &c->comm_stream)); }
  CK(hipEventCreateWithFlags(&c->ev_E, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_H, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_haloE, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_haloH, hipEventDisableTiming));
  const size_t fbytes = (size_t)plane * (d->nk + 2) * sizeof(float);
  // behind every voltage array: room for the Mur candidates of its component (four faces: kernels.hip build_mur_table, mur_load_V)
  c->mur_tail = 6 * ((size_t)d->ny * P + (size_t)d->nk * P + (size_t)d->nk * d->ny) + 96 + P;   // (cd twice, and st)
  for (int n = 0; n < 6; ++n) {
    const size_t bytes = fbytes + (n < 3 ? c->mur_tail * sizeof(float) : 0);
    CK(hipMalloc(&c->fieldbase[n], bytes));
    CK(hipMemset(c->fieldbase[n], 0, bytes));
  }
  CK(hipMalloc(&c->d_energy, (2 + 2 * ENERGY_BLOCKS + 1) * sizeof(double)));
  CK(hipMemset(c->d_energy, 0, (2 + 2 * ENERGY_BLOCKS + 1) * sizeof(double)));
  CK(hipMalloc(&c->d_probe, sizeof(DevProbe) * FDTD_MAX_PROBES));
  CK(hipMalloc(&c->d_box, sizeof(DevBox) * FDTD_MAX_BOXES));
  CK(hipMalloc(&c->lut, 256 * sizeof(float2)));
  CK(hipMemset(c->lut, 0, 256 * sizeof(float2)));
  // a one-sample zero signal so k_post always has a valid pointer
  CK(hipMalloc(&c->sig, sizeof(float)));
  CK(hipMemset(c->sig, 0, sizeof(float)));
  c->nsig = 0;
#undef CK
  DevParams& p = c->p;
  p.nx = d->nx; p.ny = d->ny; p.nk = d->nk; p.P = (int)P; p.P4 = (int)(P / 4);
  p.plane = (int)plane; p.nloc = (int)c->nloc;
  for (int n = 0; n < 3; ++n) {
    p.V[n] = c->fieldbase[n] + plane;
    p.I[n] = c->fieldbase[3 + n] + plane;
  }
  for (int a = 0; a < 3; ++a) { p.pml_lo[a] = 0; p.pml_hi[a] = 1 << 30; p.pml_hi_slot[a] = 0; p.nslot[a] = 0; }
  // six field arrays + class bytes vs the 256 MiB Infinity Cache
  p.nt = ((size_t)c->nloc * (6 * sizeof(float) + 1) > (size_t)200 << 20) ? 1 : 0;
  choose_tiling(c);
  p.sweep_rev = getenv("FDTD_NO_SWEEP_REV") ? 0 : 1;
  if (const char* v = getenv("FDTD_OCC_E")) c->occ_e = std::max(0, std::min(16, atoi(v)));
  if (const char* v = getenv("FDTD_NT")) p.nt = atoi(v) ? 1 : 0;                       // experiments
  if (const char* v = getenv("FDTD_OCC_WF")) c->occ_wf = std::max(0, std::min(16, atoi(v)));
  if (const char* e = getenv("FDTD_MUR_APPLY_PASS")) c->mur_no_apply = atoi(e) == 0;   // =1: keep the apply pass as a launch of its own (A/B, tests)
  if (getenv("FDTD_MUR_UNFUSED")) c->mur_fuse_post = false;   // experiments: the Mur post pass as a launch of its own
  if (const char* v = getenv("FDTD_WAVEFRONT")) c->wf_mode = atoi(v) ? 1 : 0;
  if (const char* v = getenv("FDTD_WF_LAG")) c->wf_lag = std::max(0, std::min(4096, atoi(v)));
  if (const char* v = getenv("FDTD_RCCL_INLINE")) c->rccl_inline_mode = atoi(v) ? 1 : 0;  // RCCL halos in stream order on the compute stream (1) / overlapped on the communication stream (0)
  if (const char* v = getenv("FDTD_RESIDENT")) c->res_mode = atoi(v) ? 1 : 0;          // 1: the resident schedule whenever it is possible, 0: never
  if (const char* v = getenv("FDTD_RES_CHUNK")) c->res_chunk = std::max(1, std::min(4096, atoi(v)));
  if (const char* v = getenv("FDTD_WF_MULTI")) c->wf_multi = std::max(1, std::min(4096, atoi(v)));   // timesteps per launch at most (1: one launch per timestep)
  if (const char* v = getenv("FDTD_OCC_H")) c->occ_h = std::max(0, std::min(16, atoi(v)));
  if (const char* v = getenv("FDTD_P2P_FAULT_STEP")) c->p2p_fault_step = atoll(v);   // test hook: see fdtd_run
  if (const char* v = getenv("FDTD_WF_FAULT_STEP")) c->wf_fault_step = atoll(v);   // test hook: the H blocks of that step wait for flags nobody sets (bounded wait -> error word)
  if (const char* v = getenv("FDTD_XCD_BALANCE")) c->xcd_balance = atoi(v) != 0;        // experiments: 0 = XCD shares of equal length
  if (const char* v = getenv("FDTD_XCD_ADAPT")) c->xcd_adapt = atoi(v) != 0;            // 0 = the cost model's cuts, never the measured ones
  if (const char* v = getenv("FDTD_XCD_WY")) c->xw_y = std::max(0.0, std::min(4.0, atof(v)));
  if (const char* v = getenv("FDTD_XCD_WZ")) c->xw_z = std::max(0.0, std::min(4.0, atof(v)));
  if (const char* v = getenv("FDTD_XCD_WYZ")) c->xw_yz = std::max(0.0, std::min(4.0, atof(v)));
  // The block -> XCD mapping of the update kernels (index % 8), the cost-weighted shares and their measured correction describe the WHOLE chip:
  // 8 XCDs of 32 CUs.  On a partitioned gfx950 (CPX / DPX logical devices: 32 / 128 CUs, 1 / 4 XCDs) results are unaffected, but eight weighted
  // shares would split one XCD's work into padded pieces and the adaptation would fit noise: equal shares, no adaptation there.
  if (chip_cus(d->device) != 256 && !getenv("FDTD_XCD_BALANCE")) { c->xcd_balance = false; c->xcd_adapt = false; }
  p.src_dense_ok = 1;
  p.src_rng = nullptr; p.src_ids = nullptr; p.nsrc = 0; p.src_off = nullptr; p.src_comp = nullptr; p.src_amp = nullptr; p.src_delay = nullptr;
  p.sig = c->sig; p.nsig = 0;
  p.probes = c->d_probe; p.nprobe = 0; p.max_steps = d->max_steps;
  *out = c;
  return FDTD_OK;
}

void fdtd_destroy(fdtd_ctx* c) {
  if (!c) return;
  hipSetDevice(c->d.device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
  if (c->comm) ncclCommDestroy((ncclComm_t)c->comm);
  for (int n = 0; n < 6; ++n) hipFree(c->fieldbase[n]);
  hipFree(c->vv); hipFree(c->vi); hipFree(c->ii); hipFree(c->iv); hipFree(c->ecls); hipFree(c->lut); hipFree(c->met);
  hipFree(c->cpcoef); hipFree(c->xc_tab);
  hipFree(c->xstamp);
  res_free(c);
  hipFree(c->wf_flags); hipFree(c->wf_err); hipFree(c->wf_flagsH); hipFree(c->wf_prb_sp); hipFree(c->wf_prb_blk); hipFree(c->wf_prb_rng);
  hipFree(c->wf_prbV_sp); hipFree(c->wf_prb_done);
  for (int n = 0; n < 12; ++n) hipFree(c->psi[n]);
  hipFree(c->d_mur);
  hipFree(c->rt_tw[0]); hipFree(c->rt_tw[1]); hipFree(c->rt_out);
  hipFree(c->sig); hipFree(c->src_off); hipFree(c->src_comp); hipFree(c->src_amp); hipFree(c->src_delay);
  for (int q = 0; q < c->nprobe; ++q) {
    hipFree((void*)c->probe[q].off); hipFree((void*)c->probe[q].comp); hipFree((void*)c->probe[q].w); hipFree(c->probe[q].series);
  }
  for (int b = 0; b < c->nbox; ++b) { hipFree(c->box[b].acc); hipFree(c->box[b].rec); }
  hipFree(c->d_probe); hipFree(c->d_box); hipFree(c->tw_v); hipFree(c->tw_i);
  hipFree(c->d_energy); hipFree(c->src_rng); hipFree(c->src_ids);
  if (c->peer_lo && c->peer_lo_ipc) hipIpcCloseMemHandle(c->peer_lo);
  if (c->peer_hi && c->peer_hi_ipc) hipIpcCloseMemHandle(c->peer_hi);
  hipFree(c->mbox);
  if (c->ev_E) hipEventDestroy(c->ev_E);
  if (c->ev_H) hipEventDestroy(c->ev_H);
  if (c->ev_haloE) hipEventDestroy(c->ev_haloE);
  if (c->ev_haloH) hipEventDestroy(c->ev_haloH);
  if (!c->stream_shared) stream_give(c->d.device, c->stream);          // (both were synchronised at the top)
  stream_give(c->d.device, c->comm_stream);
  delete c;
}


int fdtd_set_operator_raw(fdtd_ctx* c, const float* vv, const float* vi, const float* ii, const float* iv) {
  if (!c || !vv || !vi || !ii || !iv) return fdtd_fail(c, FDTD_E_ARG, "null operator array");
  if (3 * c->nloc >= ((size_t)1 << 31)) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "raw operator: slab exceeds 2^31 / 3 elements per component (use more z-slabs)");
  HIPCK(c, hipSetDevice(c->d.device));
  const size_t bytes = 3 * c->nloc * sizeof(float);
  const float* src[4] = {vv, vi, ii, iv};
  float** dst[4] = {&c->vv, &c->vi, &c->ii, &c->iv};
  const size_t rows = (size_t)3 * c->d.nk * c->d.ny;
  for (int n = 0; n < 4; ++n) {
    if (!*dst[n]) HIPCK(c, hipMalloc(dst[n], bytes));
    HIPCK(c, hipMemset(*dst[n], 0, bytes));
    HIPCK(c, upload_rows(*dst[n], c->P, src[n], c->d.nx, rows));
  }
  c->p.vv = c->vv; c->p.vi = c->vi; c->p.ii = c->ii; c->p.iv = c->iv;
  c->have_op = true; c->raw_op = true; c->op_nclasses = 0;
  return FDTD_OK;
}

int fdtd_set_operator_classes(fdtd_ctx* c, const uint8_t* ecls, int ncls, const float* cls_vv, const float* cls_m,
                              const float* emet, const float* hmet) {
  if (!c || !ecls || !cls_vv || !cls_m || !emet || !hmet || ncls < 1 || ncls > 256)
    return fdtd_fail(c, FDTD_E_ARG, "bad class operator");
  HIPCK(c, hipSetDevice(c->d.device));
  const int nx = c->d.nx, ny = c->d.ny, nk = c->d.nk, P = c->P;
  const size_t n = 3 * c->nloc;
  for (size_t q = 0; q < (size_t)3 * nk * ny * nx; ++q)
    if (ecls[q] >= ncls) return fdtd_fail(c, FDTD_E_ARG, "class %d >= ncls %d", (int)ecls[q], ncls);
  const size_t n_alloc = n + 64;
  if (!c->ecls) HIPCK(c, hipMalloc(&c->ecls, n_alloc));
  HIPCK(c, hipMemset(c->ecls, 0, n_alloc));
  // One byte per CELL when the scene has <= 256 distinct (cx, cy, cz) class triples (1 B instead of
  // 3 B per cell-step of coefficient traffic); otherwise one byte per edge.
  {
    const size_t ncell = (size_t)nk * ny * nx;
    std::vector<int> slot(1 << 24, -1);
    std::vector<uint32_t> triples;
    std::vector<uint8_t> packed(ncell);
    bool ok = true;
    for (size_t q = 0; q < ncell && ok; ++q) {
      const uint32_t key = (uint32_t)ecls[q] | ((uint32_t)ecls[ncell + q] << 8) | ((uint32_t)ecls[2 * ncell + q] << 16);
      int sidx = slot[key];
      if (sidx < 0) {
        if (triples.size() == 256) { ok = false; break; }
        sidx = slot[key] = (int)triples.size();
        triples.push_back(key);
      }
      packed[q] = (uint8_t)sidx;
    }
    c->packed_op = ok;
    if (!ok && 3 * c->nloc >= ((size_t)1 << 31))
      return fdtd_fail(c, FDTD_E_UNSUPPORTED, "per-edge class operator: slab exceeds 2^31 / 3 elements per component (use more z-slabs)");
    if (ok) {
      HIPCK(c, upload_rows(c->ecls, P, packed.data(), nx, (size_t)nk * ny));
      std::vector<float2> lut3(768, make_float2(0.f, 0.f));
      for (size_t t = 0; t < triples.size(); ++t)
        for (int comp = 0; comp < 3; ++comp) {
          const int cl = (triples[t] >> (8 * comp)) & 0xFF;
          lut3[3 * t + comp] = make_float2(cls_vv[cl], cls_m[cl]);
        }
      hipFree(c->lut); c->lut = nullptr;
      HIPCK(c, hipMalloc(&c->lut, 768 * sizeof(float2)));
      HIPCK(c, hipMemcpy(c->lut, lut3.data(), 768 * sizeof(float2), hipMemcpyHostToDevice));
      c->p.lut_n = (int)(3 * triples.size());
    } else {
      HIPCK(c, upload_rows(c->ecls, P, ecls, nx, (size_t)3 * nk * ny));
      std::vector<float2> lut(256, make_float2(0.f, 0.f));
      for (int q = 0; q < ncls; ++q) lut[q] = make_float2(cls_vv[q], cls_m[q]);
      hipFree(c->lut); c->lut = nullptr;
      HIPCK(c, hipMalloc(&c->lut, 256 * sizeof(float2)));
      HIPCK(c, hipMemcpy(c->lut, lut.data(), 256 * sizeof(float2), hipMemcpyHostToDevice));
      c->p.lut_n = ncls;
    }
  }
  { int r = upload_metric_tables(c, emet, hmet); if (r) return r; }
  c->p.ecls = c->ecls; c->p.lut = c->lut;
  c->have_op = true; c->raw_op = false;
  c->op_nclasses = ncls;
  return FDTD_OK;
}

int fdtd_set_cpml(fdtd_ctx* c, const int32_t* sx, const int32_t* sy, const int32_t* sz, int nsx, int nsy, int nsz,
                  const float* coef) {
  if (!c || !sx || !sy || !sz || !coef) return fdtd_fail(c, FDTD_E_ARG, "null cpml argument");
  HIPCK(c, hipSetDevice(c->d.device));
  const int32_t* s[3] = {sx, sy, sz};
  const int ns[3] = {nsx, nsy, nsz};
  // the kernels address psi through two contiguous index ranges per axis: [0,lo) and [hi,n)
  for (int a = 0; a < 3; ++a) {
    const int n = n_axis(c, a);
    int lo = 0;
    while (lo < n && s[a][lo] == lo) ++lo;
    int hi = lo;
    while (hi < n && s[a][hi] < 0) ++hi;
    for (int q = hi; q < n; ++q)
      if (s[a][q] != lo + (q - hi)) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "cpml slots on axis %d are not two end-anchored contiguous ranges", a);
    const int used = lo + (n - hi);
    if (used != ns[a]) return fdtd_fail(c, FDTD_E_ARG, "cpml axis %d: %d slots declared, %d used", a, ns[a], used);
    c->p.pml_lo[a] = lo; c->p.pml_hi[a] = hi < n ? hi : (1 << 30); c->p.pml_hi_slot[a] = lo; c->p.nslot[a] = ns[a];
    if (a == 0) {
      // internal x layout: both ranges start on a 4-cell boundary, so a thread's four cells are one aligned
      // float4 of psi (cells drawn in that are outside the real layer carry identity coefficients)
      const int lo4 = (lo + 3) / 4 * 4;
      int ahi = hi < n ? hi / 4 * 4 : (1 << 30);
      if (ahi < lo4) ahi = lo4;
      c->p.pml_lo[0] = lo4; c->p.pml_hi[0] = ahi; c->p.pml_hi_slot[0] = lo4;
      c->p.nslot[0] = lo4 + (hi < n ? c->P - ahi : 0);
    }
  }
  // coefficient tables -> device, x tables padded to P (identity)
  const int nx = c->d.nx, ny = c->d.ny, nk = c->d.nk, P = c->P;
  auto al4 = [](int v) { return (v + 3) / 4 * 4; };
  const int len[3] = {P + 4, al4(ny) + 4, al4(nk) + 4};
  const int nn[3] = {nx, ny, nk};
  size_t tot = 0;
  for (int a = 0; a < 3; ++a) tot += (size_t)6 * len[a];
  std::vector<float> host(tot, 0.f);
  size_t doff = 0, soff = 0;
  size_t tab_off[3][2][3];
  for (int a = 0; a < 3; ++a)
    for (int eh = 0; eh < 2; ++eh)
      for (int w = 0; w < 3; ++w) {
        float* dst = host.data() + doff;
        for (int q = 0; q < len[a]; ++q) dst[q] = (w == 2) ? 1.f : 0.f;
        memcpy(dst, coef + soff, nn[a] * sizeof(float));
        tab_off[a][eh][w] = doff;
        doff += len[a]; soff += nn[a];
      }
  hipFree(c->cpcoef); c->cpcoef = nullptr;
  HIPCK(c, hipMalloc(&c->cpcoef, tot * sizeof(float)));
  HIPCK(c, hipMemcpy(c->cpcoef, host.data(), tot * sizeof(float), hipMemcpyHostToDevice));
  for (int a = 0; a < 3; ++a)
    for (int eh = 0; eh < 2; ++eh)
      for (int w = 0; w < 3; ++w) c->p.cp[a][eh][w] = c->cpcoef + tab_off[a][eh][w];
  // x-layer coefficients once more, compact and in psi-slot order, for the kernels' LDS table (kernel_common.hpp)
  hipFree(c->xc_tab); c->xc_tab = nullptr; c->p.xc_tab = nullptr;
  if (c->p.nslot[0] > 0 && c->p.nslot[0] <= 128) {
    std::vector<float> xt((size_t)2 * 3 * 128);
    for (int eh = 0; eh < 2; ++eh)
      for (int w = 0; w < 3; ++w)
        for (int sx = 0; sx < 128; ++sx) {
          const int i = sx < c->p.pml_lo[0] ? sx : sx - c->p.pml_hi_slot[0] + c->p.pml_hi[0];
          xt[((size_t)eh * 3 + w) * 128 + sx] = (sx < c->p.nslot[0] && i < len[0]) ? host[tab_off[0][eh][w] + i] : (w == 2 ? 1.f : 0.f);
        }
    HIPCK(c, hipMalloc(&c->xc_tab, xt.size() * sizeof(float)));
    HIPCK(c, hipMemcpy(c->xc_tab, xt.data(), xt.size() * sizeof(float), hipMemcpyHostToDevice));
    c->p.xc_tab = c->xc_tab;
  }
  // psi: axis x -> rows of xrs floats per (k, j); y -> [nk][nsy][P]; z -> [nsz][ny][P].
  // x rows: a wave that straddles a row boundary holds the three lanes at the end of row j - 1 (high layer) and the three at the
  // start of row j (low layer); row j of the psi arrays holds exactly those 24 values, on a 128-byte line of its own (xrs = 32 floats):
  // one line per wave and array instead of pieces of two lines shared with the neighbouring waves — what write-through stores
  // (several timesteps per launch) need: north-star grid with the x layers alone 62.6 -> 60.9 us per timestep (profiles/r03/x_psi_rows_on_lines_ab.txt).
  // $FDTD_XPSI_PACKED=1: rows of nslot floats back to back (the round-2 layout).
  {
    DevParams& q = c->p;
    const int lo4 = q.pml_lo[0], hs = q.pml_hi[0] < (1 << 30) ? P - q.pml_hi[0] : 0;
    if (getenv("FDTD_XPSI_PACKED")) { q.xrs = q.nslot[0]; q.xplane = ny * q.xrs; q.xlo_off = 0; q.xhi_off = lo4; }
    else { q.xrs = (q.nslot[0] + 31) / 32 * 32; q.xplane = (ny + 1) * q.xrs; q.xlo_off = hs; q.xhi_off = q.xrs; }
    if ((long)nk * q.xplane >= (1L << 30)) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "x-directed psi arrays exceed 2^30 elements: use more z-slabs");
  }
  const size_t psz[3] = {(size_t)nk * c->p.xplane + (size_t)c->p.xrs, (size_t)nk * nsy * P, (size_t)nsz * ny * P};
  for (int n = 0; n < 12; ++n) { hipFree(c->psi[n]); c->psi[n] = nullptr; }
  for (int eh = 0; eh < 2; ++eh)
    for (int comp = 0; comp < 3; ++comp)
      for (int w = 0; w < 2; ++w) {
        const int a = (comp + 1 + w) % 3;
        float* ptr = nullptr;
        // allocate at least 16 B so interior kernels always hold a valid pointer
        const size_t bytes = psz[a] * sizeof(float) + 64;
        HIPCK(c, hipMalloc(&ptr, bytes));
        HIPCK(c, hipMemset(ptr, 0, bytes));
        c->psi[(eh * 3 + comp) * 2 + w] = ptr;
        (eh ? c->p.psiH : c->p.psiE)[comp][w] = ptr;
      }
  c->have_cpml = (nsx + nsy + nsz) > 0;
  xcd_shares_reset(c);   // the layers decide what a block costs
  return FDTD_OK;
}

int fdtd_set_mur(fdtd_ctx* c, const int32_t enable[6], const float coeff[6]) {
  if (!c || !enable || !coeff) return fdtd_fail(c, FDTD_E_ARG, "null mur argument");
  HIPCK(c, hipSetDevice(c->d.device));
  c->any_mur = false;
  for (int f = 0; f < 6; ++f) {
    const int a = f / 2;
    int on = enable[f] != 0;
    if (a == 2) {
      const int b = (f & 1) ? c->d.nz - 1 : 0;
      if (b < c->d.k0 || b >= c->d.k0 + c->d.nk) on = 0;
      else if (on && c->d.nk < 2) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "Mur z face needs nk >= 2");
    }
    const size_t n = a == 0 ? (size_t)c->d.nk * c->d.ny : a == 1 ? (size_t)c->d.nk * c->d.nx : (size_t)c->d.ny * c->d.nx;
    c->mur[f].on = on; c->mur[f].coeff = coeff[f]; c->mur[f].n = (int)n;
    c->any_mur |= on != 0;
  }
  for (int n = 0; n < 3; ++n)   // the candidates behind the voltage arrays: a new set of faces starts from zeros
    HIPCK(c, hipMemset(c->fieldbase[n] + (size_t)c->plane * (c->d.nk + 2), 0, c->mur_tail * sizeof(float)));
  return build_mur_table(c);
}

int fdtd_set_signal(fdtd_ctx* c, const float* sig, int n) {
  if (!c || !sig || n < 1) return fdtd_fail(c, FDTD_E_ARG, "bad signal");
  HIPCK(c, hipSetDevice(c->d.device));
  hipFree(c->sig); c->sig = nullptr;
  HIPCK(c, hipMalloc(&c->sig, n * sizeof(float)));
  HIPCK(c, hipMemcpy(c->sig, sig, n * sizeof(float), hipMemcpyHostToDevice));
  c->nsig = n;
  c->p.sig = c->sig; c->p.nsig = n;
  return FDTD_OK;
}

// global flat node index -> padded local offset; -1 not owned, -2 outside the grid
static long to_local(const fdtd_ctx* c, int64_t g) {
  const int64_t gplane = (int64_t)c->d.nx * c->d.ny;
  if (g < 0 || g >= gplane * c->d.nz) return -2;
  const int64_t k = g / gplane, r = g - k * gplane;
  if (k < c->d.k0 || k >= c->d.k0 + c->d.nk) return -1;
  const int64_t j = r / c->d.nx, i = r - j * c->d.nx;
  return (long)((k - c->d.k0) * c->plane + j * c->P + i);
}


int fdtd_add_source(fdtd_ctx* c, int n, const int64_t* idx, const int8_t* comp, const float* amp, const int32_t* delay) {
  if (!c || n < 0 || (n && (!idx || !comp || !amp || !delay))) return fdtd_fail(c, FDTD_E_ARG, "bad source");
  HIPCK(c, hipSetDevice(c->d.device));
  for (int e = 0; e < n; ++e) {
    const long l = to_local(c, idx[e]);
    if (l == -2 || comp[e] < 0 || comp[e] > 2) return fdtd_fail(c, FDTD_E_ARG, "source edge %d out of grid", e);
    if (l < 0) continue;
    c->h_src_off.push_back((int)l); c->h_src_comp.push_back(comp[e]);
    c->h_src_amp.push_back(amp[e]); c->h_src_delay.push_back(delay[e]);
  }
  c->nsrc = (int)c->h_src_off.size();
  HIPCK(c, to_device(&c->src_off, c->h_src_off));
  HIPCK(c, to_device(&c->src_comp, c->h_src_comp));
  HIPCK(c, to_device(&c->src_amp, c->h_src_amp));
  HIPCK(c, to_device(&c->src_delay, c->h_src_delay));
  HIPCK(c, build_source_lists(c, c->p.tys, c->p.nstrips, &c->src_rng, &c->src_ids));
  c->p.src_rng = c->src_rng; c->p.src_ids = c->src_ids;
  {   // two sources on one edge?  (then the dense form of body_E, which gives every edge one slot, is not taken)
    std::vector<long long> key(c->h_src_off.size());
    for (size_t e = 0; e < key.size(); ++e) key[e] = (long long)c->h_src_off[e] * 4 + c->h_src_comp[e];
    std::sort(key.begin(), key.end());
    c->p.src_dense_ok = std::adjacent_find(key.begin(), key.end()) == key.end() ? 1 : 0;
  }
  c->p.nsrc = c->nsrc; c->p.src_off = c->src_off; c->p.src_comp = c->src_comp; c->p.src_amp = c->src_amp;
  c->p.src_delay = c->src_delay;
  return FDTD_OK;
}

int fdtd_add_probe(fdtd_ctx* c, int kind, int n, const int64_t* idx, const int8_t* comp, const float* w, int* id_out) {
  if (!c || n < 0 || (n && (!idx || !comp || !w)) || (kind != 0 && kind != 1)) return fdtd_fail(c, FDTD_E_ARG, "bad probe");
  if (c->nprobe >= FDTD_MAX_PROBES) return fdtd_fail(c, FDTD_E_NOMEM, "too many probes");
  HIPCK(c, hipSetDevice(c->d.device));
  std::vector<int> off; std::vector<int8_t> cm; std::vector<float> ww;
  for (int e = 0; e < n; ++e) {
    const long l = to_local(c, idx[e]);
    if (l == -2 || comp[e] < 0 || comp[e] > 2) return fdtd_fail(c, FDTD_E_ARG, "probe edge %d out of grid", e);
    if (l < 0) continue;
    off.push_back((int)l); cm.push_back(comp[e]); ww.push_back(w[e]);
  }
  DevProbe& p = c->probe[c->nprobe];
  p = DevProbe{};
  p.kind = kind; p.n = (int)off.size();
  int* d_off = nullptr; int8_t* d_cm = nullptr; float* d_w = nullptr;
  HIPCK(c, to_device(&d_off, off));
  HIPCK(c, to_device(&d_cm, cm));
  HIPCK(c, to_device(&d_w, ww));
  p.off = d_off; p.comp = d_cm; p.w = d_w;
  c->h_prb_off[c->nprobe] = off; c->wf_prb_dirty = true;
  const size_t cap = std::max(c->d.max_steps, 1);
  HIPCK(c, hipMalloc(&p.series, cap * sizeof(double)));
  HIPCK(c, hipMemset(p.series, 0, cap * sizeof(double)));
  if (id_out) *id_out = c->nprobe;
  c->nprobe++;
  HIPCK(c, hipMemcpy(c->d_probe, c->probe, sizeof(DevProbe) * FDTD_MAX_PROBES, hipMemcpyHostToDevice));
  c->p.probes = c->d_probe; c->p.nprobe = c->nprobe;
  return FDTD_OK;
}

int fdtd_get_probe(fdtd_ctx* c, int id, double* out, int cap, int* n_out) {
  if (!c || id < 0 || id >= c->nprobe) return fdtd_fail(c, FDTD_E_ARG, "bad probe id");
  HIPCK(c, hipSetDevice(c->d.device));
  const int n = (int)std::min<int64_t>(c->step, c->d.max_steps);
  if (n_out) *n_out = n;
  const int m = std::min(n, cap);
  if (out && m > 0) {
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemcpy(out, c->probe[id].series, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
  }
  return FDTD_OK;
}

int fdtd_set_dft(fdtd_ctx* c, int nfreq, int every, int nsamples, const double* tw_v, const double* tw_i) {
  if (!c || nfreq < 1 || every < 1 || nsamples < 1 || !tw_v || !tw_i) return fdtd_fail(c, FDTD_E_ARG, "bad dft setup");
  if (c->nbox) return fdtd_fail(c, FDTD_E_STATE, "set_dft must precede add_dft_box");
  HIPCK(c, hipSetDevice(c->d.device));
  const size_t bytes = (size_t)nsamples * nfreq * 2 * sizeof(double);
  hipFree(c->tw_v); hipFree(c->tw_i); c->tw_v = c->tw_i = nullptr;
  HIPCK(c, hipMalloc(&c->tw_v, bytes));
  HIPCK(c, hipMalloc(&c->tw_i, bytes));
  HIPCK(c, hipMemcpy(c->tw_v, tw_v, bytes, hipMemcpyHostToDevice));
  HIPCK(c, hipMemcpy(c->tw_i, tw_i, bytes, hipMemcpyHostToDevice));
  c->nfreq = nfreq; c->every = every; c->nsamples = nsamples;
  return FDTD_OK;
}

// Time-domain recording instead of running sums: the boxes keep their raw samples in HBM and fdtd_rec_transform turns them
// into any frequency set afterwards — what CalcNF2FF(sim_path, f, ...) does with the engine's dumps (fixed.py:220,296).
int fdtd_set_recorder(fdtd_ctx* c, int every, int nsamples) {
  if (!c || every < 1 || nsamples < 1) return fdtd_fail(c, FDTD_E_ARG, "bad recorder setup");
  if (c->nbox) return fdtd_fail(c, FDTD_E_STATE, "set_recorder must precede add_dft_box");
  HIPCK(c, hipSetDevice(c->d.device));
  hipFree(c->tw_v); hipFree(c->tw_i); c->tw_v = c->tw_i = nullptr;
  c->nfreq = 0; c->recorder = true; c->every = every; c->nsamples = nsamples;
  return FDTD_OK;
}

int fdtd_add_dft_box(fdtd_ctx* c, int kind, int comp, const int32_t lo[3], const int32_t hi[3], int* id_out) {
  if (!c || !lo || !hi || comp < 0 || comp > 2 || (kind != 0 && kind != 1)) return fdtd_fail(c, FDTD_E_ARG, "bad dft box");
  if (!c->nfreq && !c->recorder) return fdtd_fail(c, FDTD_E_STATE, "set_dft or set_recorder first");
  if (c->nbox >= FDTD_MAX_BOXES) return fdtd_fail(c, FDTD_E_NOMEM, "too many dft boxes");
  HIPCK(c, hipSetDevice(c->d.device));
  const int dims[3] = {c->d.nx, c->d.ny, c->d.nz};
  for (int a = 0; a < 3; ++a)
    if (lo[a] < 0 || hi[a] >= dims[a] || hi[a] < lo[a]) return fdtd_fail(c, FDTD_E_ARG, "dft box outside grid");
  const int b = c->nbox;
  int32_t olo[3] = {lo[0], lo[1], std::max(lo[2], c->d.k0)};
  int32_t ohi[3] = {hi[0], hi[1], std::min(hi[2], c->d.k0 + c->d.nk - 1)};
  for (int a = 0; a < 3; ++a) { c->box_lo[b][a] = olo[a]; c->box_hi[b][a] = ohi[a]; }
  DevBox& bx = c->box[b];
  bx = DevBox{};
  bx.kind = kind; bx.comp = comp;
  if (ohi[2] >= olo[2]) {
    bx.lo[0] = olo[0]; bx.lo[1] = olo[1]; bx.lo[2] = olo[2] - c->d.k0;
    bx.ni = ohi[0] - olo[0] + 1; bx.nj = ohi[1] - olo[1] + 1; bx.nkk = ohi[2] - olo[2] + 1;
    bx.npts = (long)bx.ni * bx.nj * bx.nkk;
    if (c->recorder) {
      const size_t bytes = (size_t)bx.npts * (size_t)c->nsamples * sizeof(float);
      if (hipMalloc(&bx.rec, bytes) != hipSuccess) {
        (void)hipGetLastError();
        bx = DevBox{};
        return fdtd_fail(c, FDTD_E_NOMEM, "recorder box: %zu bytes of device memory (%ld points x %d samples); use fdtd_set_dft", bytes, (long)((ohi[0] - olo[0] + 1) * (long)(ohi[1] - olo[1] + 1) * (ohi[2] - olo[2] + 1)), c->nsamples);
      }
    } else {
      const size_t bytes = (size_t)bx.npts * c->nfreq * 2 * sizeof(double);
      HIPCK(c, hipMalloc(&bx.acc, bytes));
      HIPCK(c, hipMemset(bx.acc, 0, bytes));
    }
    c->box_maxpts[kind] = std::max(c->box_maxpts[kind], bx.npts);
  }
  if (id_out) *id_out = b;
  c->nbox++;
  HIPCK(c, hipMemcpy(c->d_box, c->box, sizeof(DevBox) * FDTD_MAX_BOXES, hipMemcpyHostToDevice));
  return FDTD_OK;
}

int fdtd_get_dft_box(fdtd_ctx* c, int id, double* out, int32_t lo_own[3], int32_t hi_own[3]) {
  if (!c || id < 0 || id >= c->nbox) return fdtd_fail(c, FDTD_E_ARG, "bad dft box id");
  HIPCK(c, hipSetDevice(c->d.device));
  for (int a = 0; a < 3; ++a) { if (lo_own) lo_own[a] = c->box_lo[id][a]; if (hi_own) hi_own[a] = c->box_hi[id][a]; }
  const DevBox& bx = c->box[id];
  if (out && c->recorder) return fdtd_fail(c, FDTD_E_STATE, "recorder mode: use fdtd_rec_transform");
  if (out && bx.npts) {
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemcpy(out, bx.acc, (size_t)bx.npts * c->nfreq * 2 * sizeof(double), hipMemcpyDeviceToHost));
  }
  return FDTD_OK;
}

int fdtd_rec_transform(fdtd_ctx* c, int id, int nfreq, const double* tw, double* out, int32_t lo_own[3], int32_t hi_own[3]) {
  if (!c || id < 0 || id >= c->nbox) return fdtd_fail(c, FDTD_E_ARG, "bad box id");
  if (!c->recorder) return fdtd_fail(c, FDTD_E_STATE, "not in recorder mode");
  for (int a = 0; a < 3; ++a) { if (lo_own) lo_own[a] = c->box_lo[id][a]; if (hi_own) hi_own[a] = c->box_hi[id][a]; }
  const DevBox& bx = c->box[id];
  if (!out || !bx.npts) return FDTD_OK;
  if (nfreq < 1 || !tw) return fdtd_fail(c, FDTD_E_ARG, "bad transform");
  HIPCK(c, hipSetDevice(c->d.device));
  // samples taken so far: steps 0, every, 2*every, ... of the half-steps already done
  const int ns = (int)std::min<int64_t>((c->step + c->every - 1) / c->every, c->nsamples);
  // Device scratch kept in the context (CalcNF2FF transforms 24 boxes with two twiddle tables — E-located and H-located sample times —, one call
  // each: allocating, uploading and freeing per call was most of the 6 ms the transform took on the reference's default scene): two twiddle slots
  // recognised by content, one output buffer grown on demand.
  const size_t tw_n = (size_t)std::max(ns, 0) * nfreq * 2, out_bytes = (size_t)bx.npts * nfreq * 2 * sizeof(double);
  int slot = -1;
  for (int q = 0; q < 2; ++q)
    if (c->rt_tw[q] && c->rt_tw_host[q].size() == tw_n && (tw_n == 0 || memcmp(c->rt_tw_host[q].data(), tw, tw_n * sizeof(double)) == 0)) slot = q;
  hipError_t e = hipSuccess;
  if (slot < 0) {
    slot = c->rt_next; c->rt_next ^= 1;
    hipFree(c->rt_tw[slot]); c->rt_tw[slot] = nullptr; c->rt_tw_host[slot].clear();
    e = hipMalloc(&c->rt_tw[slot], std::max<size_t>(tw_n, 1) * sizeof(double));
    if (e == hipSuccess && tw_n) e = hipMemcpyAsync(c->rt_tw[slot], tw, tw_n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);     // (the caller's table may go away after this call)
    if (e == hipSuccess) c->rt_tw_host[slot].assign(tw, tw + tw_n);
    else { hipFree(c->rt_tw[slot]); c->rt_tw[slot] = nullptr; }
  }
  if (e == hipSuccess && c->rt_out_bytes < out_bytes) {
    hipFree(c->rt_out); c->rt_out = nullptr; c->rt_out_bytes = 0;
    e = hipMalloc(&c->rt_out, out_bytes);
    if (e == hipSuccess) c->rt_out_bytes = out_bytes;
  }
  if (e == hipSuccess) {
    launch_rec_dft(bx.rec, bx.npts, ns, nfreq, c->rt_tw[slot], c->rt_out, c->stream);
    e = hipMemcpyAsync(out, c->rt_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  HIPCK(c, e);
  return FDTD_OK;
}

// ------------------------------------------------------------------------------------------------
// stepping
// ------------------------------------------------------------------------------------------------
static int check_ready(fdtd_ctx* c) {
  if (!c) return FDTD_E_ARG;
  if (!c->have_op) return fdtd_fail(c, FDTD_E_STATE, "operator not set");
  return FDTD_OK;
}

// Halo exchange on the communication stream.  which = FDTD_HALO_E_DOWN: Vx,Vy plane 0 -> rank-1, ghost
// plane nk <- rank+1.  FDTD_HALO_H_UP: Ix,Iy plane nk-1 -> rank+1, ghost plane -1 <- rank-1.
// (a) RCCL: one grouped ncclSend/ncclRecv per exchange, xGMI peer-to-peer.
static int exchange_rccl(fdtd_ctx* c, int which, hipStream_t cs) {
  ncclComm_t comm = (ncclComm_t)c->comm;
  const int r = c->d.rank, w = c->d.world;
  // FDTD_FLAG_LOOPBACK (transport self-test on one GPU): both neighbours are this rank itself, in a communicator of 1
  const bool loop = (c->d.flags & FDTD_FLAG_LOOPBACK) != 0;
  const int down = loop ? 0 : r - 1, up = loop ? 0 : r + 1;
  const size_t cnt = (size_t)c->plane;
  const long top = (long)(c->d.nk - 1) * c->plane;
  NCCLCK(c, ncclGroupStart());
  if (which == FDTD_HALO_E_DOWN) {
    if (r > 0) {
      NCCLCK(c, ncclSend(c->p.V[0], cnt, ncclFloat, down, comm, cs));
      NCCLCK(c, ncclSend(c->p.V[1], cnt, ncclFloat, down, comm, cs));
    }
    if (r < w - 1) {
      NCCLCK(c, ncclRecv(c->p.V[0] + c->nloc, cnt, ncclFloat, up, comm, cs));
      NCCLCK(c, ncclRecv(c->p.V[1] + c->nloc, cnt, ncclFloat, up, comm, cs));
    }
  } else {
    if (r < w - 1) {
      NCCLCK(c, ncclSend(c->p.I[0] + top, cnt, ncclFloat, up, comm, cs));
      NCCLCK(c, ncclSend(c->p.I[1] + top, cnt, ncclFloat, up, comm, cs));
    }
    if (r > 0) {
      NCCLCK(c, ncclRecv(c->p.I[0] - c->plane, cnt, ncclFloat, down, comm, cs));
      NCCLCK(c, ncclRecv(c->p.I[1] - c->plane, cnt, ncclFloat, down, comm, cs));
    }
  }
  NCCLCK(c, ncclGroupEnd());
  return FDTD_OK;
}

// (b) contexts linked inside one process (fdtd_link): every context PULLS its ghost plane from the
// neighbour with a peer copy on its own communication stream, after the neighbour's phase event.
static int exchange_linked(fdtd_ctx* c, int which) {
  const size_t bytes = (size_t)c->plane * sizeof(float);
  if (which == FDTD_HALO_E_DOWN) {
    fdtd_ctx* up = c->link_hi;
    if (up) {
      HIPCK(c, hipStreamWaitEvent(c->comm_stream, up->ev_E, 0));
      for (int q = 0; q < 2; ++q)
        HIPCK(c, hipMemcpyPeerAsync(c->p.V[q] + c->nloc, c->d.device, up->p.V[q], up->d.device, bytes, c->comm_stream));
    }
  } else {
    fdtd_ctx* lo = c->link_lo;
    if (lo) {
      const long top = (long)(lo->d.nk - 1) * lo->plane;
      HIPCK(c, hipStreamWaitEvent(c->comm_stream, lo->ev_H, 0));
      for (int q = 0; q < 2; ++q)
        HIPCK(c, hipMemcpyPeerAsync(c->p.I[q] - c->plane, c->d.device, lo->p.I[q] + top, lo->d.device, bytes, c->comm_stream));
    }
  }
  return FDTD_OK;
}

// RCCL exchange INLINE on the compute stream: [E sweep][grouped send/recv][H sweep][grouped send/recv], nothing else.  The overlapped schedule
// (communication stream, four stream-to-stream event hops per timestep, sweeps split into interior + halo-dependent plane) costs a thin slab
// 80-85 us per timestep whatever the payload — each hop is 10-20 us of latency, there is nothing of that length to overlap with (an 8-plane
// north-star slab: 14 us of kernels) — and 8 GPUs deliver less than one (profiles/r01/halo_transport_thin_slab_timing.txt).  In stream order no
// event is needed at all: the exchange starts when the sweep before it has finished and the next sweep starts when the exchange has.  Taken when
// a sweep is short (fewer than 4096 blocks of 1024 cells: < ~50 us; larger slabs have sweeps worth overlapping), $FDTD_RCCL_INLINE=0/1 decides.
static bool rccl_inline(const fdtd_ctx* c) {
  if (!c->comm) return false;
  if (c->rccl_inline_mode >= 0) return c->rccl_inline_mode != 0;
  return (size_t)c->d.nk * c->p.nstrips * c->p.nbs < 4096;
}

static int exchange(fdtd_ctx* c, int which) {
  HIPCK(c, hipSetDevice(c->d.device));
  if (rccl_inline(c)) return exchange_rccl(c, which, c->stream);
  // nothing of this exchange may start before this slab's own half-step is complete (it both produces the
  // plane that leaves and is the last reader of the ghost plane that is about to be overwritten)
  HIPCK(c, hipStreamWaitEvent(c->comm_stream, which == FDTD_HALO_E_DOWN ? c->ev_E : c->ev_H, 0));
  int r = c->comm ? exchange_rccl(c, which, c->comm_stream) : exchange_linked(c, which);
  if (r) return r;
  if (which == FDTD_HALO_E_DOWN) { HIPCK(c, hipEventRecord(c->ev_haloE, c->comm_stream)); c->haloE_pending = true; }
  else { HIPCK(c, hipEventRecord(c->ev_haloH, c->comm_stream)); c->haloH_pending = true; }
  return FDTD_OK;
}

// Soft sources may stay inside update_E (and the probes in the extra blocks of the main kernels) also in a scene with
// Mur faces, as long as no source edge sits on a Mur face or on the plane next to it: the Mur "post" pass reads the
// freshly updated voltages of that inner plane BEFORE the sources are added in the unfused order, so only there the
// order of the two matters.  (The reference's ports sit in the middle of the box.)  Saves two launches per step.
static bool sources_fusable(const fdtd_ctx* c) {
  if (!c->p.src_dense_ok && c->src_max_per_strip_plane > FDTD_BLOCK) return false;   // (duplicate edges AND more than the scanning stage holds: k_post)
  if (!c->any_mur) return true;
  const int n[3] = {c->d.nx, c->d.ny, c->d.nz};
  for (int off : c->h_src_off) {
    const int k = off / c->plane, j = (off - k * c->plane) / c->P, i = off - k * c->plane - j * c->P;
    const int pos[3] = {i, j, c->d.k0 + k};
    for (int f = 0; f < 6; ++f) {
      const int a = f / 2;
      const bool enabled = a == 2 ? true : c->mur[f].on != 0;   // a z face may live on another rank: be conservative
      if (!enabled) continue;
      if ((f & 1) ? pos[a] >= n[a] - 2 : pos[a] <= 1) return false;
    }
  }
  return true;
}

// Mur faces without an apply pass (two launches per timestep instead of three).  Between update_E and update_H the boundary
// voltages in memory are then the E update's own, and inside update_H they are being overwritten: whoever reads V there —
// V-probes (sampled by update_H's extra block), NF2FF / DFT boxes (sampled between the two launches) — must not hold a
// node of a Mur face.  (The reference's probes and boxes sit inside the grid.)  Same conditions as the post pass inside
// update_E otherwise (phase_E).
static bool mur_direct_possible(const fdtd_ctx* c, bool multi, bool fused) {
  if (!c->mur_no_apply || !c->any_mur || multi || !fused || !c->d_mur || !c->mur_fuse_post) return false;
  if (c->d.nx < 6 || c->d.ny < 5 || c->d.nz < 5) return false;
  const int dim[3] = {c->d.nx, c->d.ny, c->d.nk};
  auto on_face = [&](const int lo[3], const int hi[3]) {
    for (int f = 0; f < 6; ++f) {
      if (!c->mur[f].on) continue;
      const int a = f / 2, b = (f & 1) ? dim[a] - 1 : 0;
      if (lo[a] <= b && b <= hi[a]) return true;
    }
    return false;
  };
  for (int q = 0; q < c->nprobe; ++q) {
    if (c->probe[q].kind != FDTD_KIND_V) continue;
    for (int off : c->h_prb_off[q]) {
      const int k = off / c->plane, j = (off - k * c->plane) / c->P, i = off - k * c->plane - j * c->P;
      const int pos[3] = {i, j, k};
      if (on_face(pos, pos)) return false;
    }
  }
  for (int b = 0; b < c->nbox; ++b) {
    const DevBox& bx = c->box[b];
    if (bx.kind != FDTD_KIND_V || bx.npts <= 0) continue;
    const int hi[3] = {bx.lo[0] + bx.ni - 1, bx.lo[1] + bx.nj - 1, bx.lo[2] + bx.nkk - 1};
    if (on_face(bx.lo, hi)) return false;
  }
  return true;
}

// a main-kernel launch the runtime refused (kernels.hip: launch_main keeps the first one): reported once, as an error code
static int launch_status(fdtd_ctx* c) {
  const int r = c->launch_failed;
  c->launch_failed = 0;
  return r;
}

struct ProfEvents {
  std::vector<hipEvent_t> e0, e1, h0, h1;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  int launches = -1;   // main launches of the profiled run when they are not one per timestep (several timesteps per launch)
};

// One leapfrog step = two main launches.  Without Mur faces the soft sources are injected inside update_E
// and the probes are sampled by one extra block of the main kernels (update_H(n): V-probes of step n;
// update_E(n+1): I-probes of step n; the last step's I-probes are flushed at the end of the call).
//
// Multi-slab schedule (RCCL ranks or linked contexts): the E halo is only needed by the TOP plane of the H
// sweep and the H halo only by the BOTTOM plane of the next E sweep, so every sweep launches all other planes
// first (overlapping the exchange in flight on the communication stream), then waits for the halo event and
// launches the one dependent plane.
// Split a sweep into "all planes but one" + "the halo-dependent plane" so that the exchange in flight overlaps
// the first part.  Measured with 8 linked NS slabs on one MI355X the split schedule is the faster one even for
// 7-plane slabs (220 vs 251 us per step for all eight), so it is the default; FDTD_FLAG_OVERLAP_OFF disables it.
static bool rccl_inline(const fdtd_ctx* c);
static bool overlap_split(const fdtd_ctx* c) { return !(c->d.flags & FDTD_FLAG_OVERLAP_OFF) && !rccl_inline(c); }

static int phase_E(fdtd_ctx* c, bool multi, bool fused, ProfEvents* pe, int n) {
  HIPCK(c, hipSetDevice(c->d.device));
  const int nk = c->d.nk;
  const long long step = c->step;
  hipStream_t s = c->stream;
  if (c->mur_pre_step != step) launch_mur(c, 0, s);   // else the previous update_H launch has done it (extra blocks)
  if (pe) { c->kev0 = pe->e0[n]; c->kev1 = pe->e1[n]; }   // the first main launch below carries them (kernel begin / end timestamps)
  const bool lower = multi && c->d.rank > 0;           // plane 0 reads the H ghost and is the plane that leaves
  const bool split = lower && overlap_split(c);
  auto wait_halo = [&]() -> int {
    if (c->haloH_pending) { HIPCK(c, hipStreamWaitEvent(s, c->ev_haloH, 0)); c->haloH_pending = false; }
    // linked transport: the lower neighbour pulls my plane 0 itself; do not overwrite it before that copy ran
    if (lower && !c->comm && c->link_lo && c->link_lo->haloE_issued) HIPCK(c, hipStreamWaitEvent(s, c->link_lo->ev_haloE, 0));
    return FDTD_OK;
  };
  if (!split) { int r = wait_halo(); if (r) return r; }
  // single slab with Mur faces and fused sources: the post pass rides in the update_E launch (E+post, apply, H+pre: three
  // launches per timestep instead of four — on the reference's default 56x55x50 scene every launch is a ~4 us latency floor)
  c->mur_post_in_E = fused && c->any_mur && !multi && c->d_mur != nullptr && c->mur_fuse_post &&
                     c->d.nx >= 6 && c->d.ny >= 5 && c->d.nz >= 5;   // (nx = 5: both inner x nodes, 1 and 3, sit in ONE thread's four cells, and MurVals holds one x pair)
  launch_update_E(c, split ? 1 : 0, nk, step, fused, true, s);
  c->kev0 = c->kev1 = nullptr;
  if (split) {
    int r = wait_halo();
    if (r) return r;
    launch_update_E(c, 0, 1, step, fused, false, s);
  }
  if (!c->mur_post_in_E) launch_mur(c, 1, s);   // post + apply (no-ops without Mur faces)
  // no apply launch when update_H takes the candidates itself (step_loop decides: mur_direct_possible)
  if (!(c->mur_direct && c->mur_post_in_E)) { launch_mur(c, 2, s); c->mur_direct = false; }
  c->mur_post_in_E = false;
  if (!fused) launch_post(c, FDTD_KIND_V, step, true, s);
  launch_dft(c, FDTD_KIND_V, step, s);
  if (multi && !rccl_inline(c)) HIPCK(c, hipEventRecord(c->ev_E, s));
  return FDTD_OK;
}

static int phase_H(fdtd_ctx* c, bool multi, bool fused, ProfEvents* pe, int n) {
  HIPCK(c, hipSetDevice(c->d.device));
  const int nk = c->d.nk;
  const long long step = c->step;
  hipStream_t s = c->stream;
  if (pe) { c->kev0 = pe->h0[n]; c->kev1 = pe->h1[n]; }
  const bool upper = multi && c->d.rank < c->d.world - 1;   // top plane reads the E ghost and is the plane that leaves
  const bool split = upper && overlap_split(c);
  auto wait_halo = [&]() -> int {
    if (c->haloE_pending) { HIPCK(c, hipStreamWaitEvent(s, c->ev_haloE, 0)); c->haloE_pending = false; }
    if (upper && !c->comm && c->link_hi && c->link_hi->haloH_issued) HIPCK(c, hipStreamWaitEvent(s, c->link_hi->ev_haloH, 0));
    return FDTD_OK;
  };
  if (!split) { int r = wait_halo(); if (r) return r; }
  // Mur scenes: the pre pass of step + 1 rides in this launch (it reads V only, which is final and not written here)
  launch_update_H(c, 0, split ? nk - 1 : nk, step, fused, s, fused && c->any_mur);
  if (fused && c->any_mur && (c->p.mur_nb > 0 || c->p.mur_direct)) c->mur_pre_step = step + 1;   // (mur_direct: the main blocks ran the pre pass)
  c->kev0 = c->kev1 = nullptr;
  if (split) {
    int r = wait_halo();
    if (r) return r;
    launch_update_H(c, nk - 1, nk, step, false, s);
  }
  if (!fused) launch_post(c, FDTD_KIND_I, step, false, s);
  launch_dft(c, FDTD_KIND_I, step, s);
  if (multi && !rccl_inline(c)) HIPCK(c, hipEventRecord(c->ev_H, s));
  return FDTD_OK;
}

static int step_loop_p2p(fdtd_ctx* c, int nsteps, struct ProfEvents* pe);
static int p2p_check(fdtd_ctx* c);

// One launch per timestep (k_step, kernels.hip): single slab, no Mur faces.  AUTO picks it where it measured faster: grids
// whose fields do not fit the 256 MiB Infinity Cache (there the H sweep finds what the E sweep just touched in that cache
// instead of in HBM); FDTD_FLAG_KERNEL_WAVEFRONT / $FDTD_WAVEFRONT=1 force it, FDTD_FLAG_KERNEL_DIRECT / =0 forbid it.
// Mur faces inside the one launch (k_step<..., MUR>): what the two-launch schedule without an apply pass needs (mur_direct_possible), a single slab
// whose fields fit the Infinity Cache (all E blocks, then all H blocks), and strips of at most 28 blocks (wf_wait_mur polls 9 * nbs flags, one thread each).
static bool wf_mur_possible(const fdtd_ctx* c) {
  return c->d.world == 1 && !c->p.p2p && mur_direct_possible(c, false, sources_fusable(c)) && 9 * c->p.nbs <= FDTD_BLOCK && wf_lag_for(c) >= c->d.nk;
}
static bool wavefront_possible(const fdtd_ctx* c) {
  // (an H block polls at most 64 flags with one wave: 2 * (1 + P4 / 256) + 3 <= 64, i.e. rows of at most 30 720 cells)
  return (c->d.world == 1 || c->p.p2p) && (!c->any_mur || wf_mur_possible(c)) && c->d.nk >= 2 && 2 * (1 + c->p.P4 / FDTD_BLOCK) + 3 <= 64 &&
         (c->p.src_dense_ok || c->src_max_per_strip_plane <= FDTD_BLOCK);   // (sources are always fused into k_step)
}
static bool wavefront_active(const fdtd_ctx* c) {
  const unsigned sel = c->d.flags & FDTD_FLAG_KERNEL_MASK;
  if (!wavefront_possible(c) || sel == FDTD_FLAG_KERNEL_DIRECT) return false;
  if (sel == FDTD_FLAG_KERNEL_WAVEFRONT) return true;
  if (c->wf_mode >= 0) return c->wf_mode != 0;
  // single slab: always (beyond the Infinity Cache with H a few planes behind E, below it with all E blocks first: wf_lag_for).
  // Slabs of a decomposed grid on the mailbox transport: only beyond the Infinity Cache (a thin slab whose halos go to itself
  // steps in 20-21 us with one launch against 17.6 us with two: the hand-off behind flags plus the halo granules make a
  // longer chain than a kernel boundary).
  const bool big = (size_t)(c->d.nk + 2) * c->plane * 6 * sizeof(float) > ((size_t)FDTD_WF_AUTO_MIB << 20);
  // (small grids WITHOUT CPML are the exception: their half-step kernels are so short that the flags cost more than the kernel
  // boundary saves — 200x200x40 without CPML: 83.5 Gcells/s with two launches, 76.5 with one; with CPML 55.6 -> 57.1)
  const size_t blocks = (size_t)c->d.nk * c->p.nstrips * c->p.nbs;   // per sweep
  // (re-measured in round 4 with several timesteps per launch, PEC: 128x128x40 (800 blocks) 13.2 us with two launches / 16.9 with one; 200x200x40 (1600) 19.5 / 19.3;
  //  143x129x89 (1780) 20.1 / 16.8; 167x143x101 (2424) 28.0 / 23.4; 256x256x48 (3360) 37.0 / 32.5; 300x300x60 58.5 / 55.9 — the threshold was 3000)
  if (c->d.world == 1) return big || c->have_cpml || blocks >= 1700;
  // Slabs that SHARE a device with a neighbour (several contexts of one process on one GPU, or several ranks on one GPU: the test boxes)
  // take two launches: fewer workgroups that can sit resident waiting for another kernel (p2p_pinned_blocks below: one plane's blocks
  // instead of three), i.e. the starvation-freedom condition of p2p_shared_device_ok holds for three times as many slabs.  On its own GPU a
  // slab cannot starve its neighbour.  (FDTD_FLAG_LOOPBACK: a slab timed ALONE as it runs in an N-GPU job keeps the N-GPU rule.)
  if (!(c->d.flags & FDTD_FLAG_LOOPBACK) && (c->link_info[0][7] == 1 || c->link_info[1][7] == 1)) return false;
  // slabs on the mailbox transport: when a sweep is more than one round of resident blocks (an interior north-star slab whose
  // halos go to itself: 20 planes 29.5 -> 25.3 us per step with one launch; 15 planes 23.3 -> 25.4, 8 planes 17.8 -> 19.8)
  return big || blocks >= 1800;
}

// The grid resident in registers for the length of a launch (k_resident, resident.hip): small single slabs — the reference GUI's default
// scenes, MUR and PML_8 alike.  FDTD_FLAG_KERNEL_RESIDENT / $FDTD_RESIDENT=1 take it wherever it is possible, DIRECT / WAVEFRONT /
// $FDTD_RESIDENT=0 never, AUTO as below.
static bool resident_active(fdtd_ctx* c) {
  const unsigned sel = c->d.flags & FDTD_FLAG_KERNEL_MASK;
  if (sel == FDTD_FLAG_KERNEL_DIRECT || sel == FDTD_FLAG_KERNEL_WAVEFRONT || c->res_mode == 0) return false;
  if (!sources_fusable(c) || !res_possible(c, nullptr)) return false;
  if (sel == FDTD_FLAG_KERNEL_RESIDENT || c->res_mode == 1) return true;
  // Mur faces: whenever it is possible (the alternative is three latency-bound launches per timestep).  PEC / CPML: while the tiles are at most two per
  // CU — 175 ... 400 tiles step in 4.4 ... 7.3 us against 9.7 ... 19.9 us of the flag-coupled launches (x 2.2 - 3.3); at 640 tiles the hop
  // latency between loaded CUs has grown so much that the two schedules tie (profiles/r04/resident_vs_multi_small_grids.txt)
  if (c->any_mur) return true;
  if (res_prepare(c, c->res_chunk) != FDTD_OK) return false;
  return c->res.nblocks <= 2 * chip_cus(c->d.device);
}
static int res_check(fdtd_ctx* c) {
  if (!c->res.err) return FDTD_OK;
  int e = 0;
  HIPCK(c, hipMemcpy(&e, c->res.err, sizeof(int), hipMemcpyDeviceToHost));
  if (e) {
    hipMemset(c->res.err, 0, sizeof(int));
    return fdtd_fail(c, FDTD_E_DEVICE, "resident schedule: a workgroup waited more than 2 s for a neighbour tile's halo (not all workgroups resident at once?); the fields of this run are invalid — re-initialise them and select FDTD_FLAG_KERNEL_DIRECT (simulation.Simulation.run does both by itself)");
  }
  return FDTD_OK;
}
static int step_loop_res(fdtd_ctx* c, int nsteps, ProfEvents* pe) {
  HIPCK(c, hipSetDevice(c->d.device));
  hipStream_t s = c->stream;
  int r = res_prepare(c, c->res_chunk);
  if (r) return r;
  // NF2FF faces: the time-domain record is written by the kernel itself (res_record); running-DFT sums (k_dft) read the arrays, so there
  // a launch ends at every sampled timestep
  const bool sampling = c->nfreq && !c->recorder && c->nbox && c->every > 0;
  int launches = 0;
  for (int n = 0; n < nsteps;) {
    int chunk = std::min(c->res_chunk, nsteps - n);
    if (sampling) {   // ... so that the sampled timestep (a multiple of `every`) is the launch's last
      const long long next = (c->step + c->every - 1) / c->every * c->every;
      chunk = (int)std::min<long long>(chunk, next - c->step + 1);
    }
    if (pe) { c->kev0 = pe->e0[launches]; c->kev1 = pe->e1[launches]; }
    r = launch_resident(c, c->step, chunk, s);
    c->kev0 = c->kev1 = nullptr;
    if (r) return r;
    c->step += chunk;
    if (sampling) launch_dft(c, -1, c->step - 1, s);
    n += chunk;
    ++launches;
  }
  c->mur_pre_step = -1;   // (the Mur state arrays are not kept by the resident kernel: the next two-launch timestep runs its own pre pass)
  if (pe) pe->launches = launches;
  HIPCK(c, hipGetLastError());
  return launch_status(c);
}

// ---- slabs on the mailbox transport that share ONE GPU: when can they not starve each other? ------------------------------------------
// A workgroup that waits for a halo granule spins while it holds its slot on the chip.  The granule is produced by ANOTHER kernel (the
// neighbour slab's); on the neighbour's own GPU that kernel always finds slots, on a shared GPU it needs slots the spinning workgroups
// may hold.  PINNED workgroups of a launch = those that can be resident without being able to finish until another KERNEL has run:
//   two launches per timestep:  the halo plane's blocks (update_E: plane 0, update_H: the top plane): B = nstrips * nbs;
//   one launch per timestep:    the E blocks of plane 0 (halo from below), the H blocks of plane 0 (they wait for those E blocks' flags), the
//                               H blocks of the top plane (halo from above) and the probe blocks: 3 B + nprobe.
// Every other workgroup of a launch depends on nothing outside its launch (or only on flags of EARLIER workgroups of its own launch, which
// in-order dispatch has made resident before it) and retires.  The producer of a granule is never itself waiting for anything that is still
// to be submitted: H halo of timestep s - 1 <- the neighbour's H blocks, which wait for MY E halo of s - 1, sent before my launch of s began
// (same stream); E halo of s <- the neighbour's E blocks of plane 0, which wait for MY H halo of s - 1, likewise sent.  So the chain of waits
// ends, at the latest at an end slab, in workgroups that need nothing but a slot — and if the pinned workgroups of ALL launches that can be
// resident together (one per slab on the device: a stream runs one kernel at a time) are fewer than the chip's slots, a slot is always free
// for them: progress, by induction along the chain.  The recorded timeout (tests/fuzz_parity.py --slabs, seed 7 case 176: six slabs of 199 x 233 x
// 45 in one-row strips, B = 233; four slabs under one launch, two under two) had 4 * 699 + 2 * 233 = 3262 pinned workgroups against 1792 slots;
// all-one-launch (4194) violates the bound as well — that it was "gone" there was a handful of lucky runs of an intermittent event, not safety —
// and two launches everywhere (1398) satisfy it.  The rule is enforced, not assumed: a configuration that violates it is FDTD_E_UNSUPPORTED
// (the run-time ladders take the next transport; fdtd_link's event-ordered peer copies never spin).
static unsigned p2p_pinned_blocks(const fdtd_ctx* c, bool one_launch) {
  const unsigned pb = (unsigned)c->p.nstrips * (unsigned)c->p.nbs;
  return one_launch ? 3u * pb + (unsigned)c->nprobe : pb;
}
static unsigned chip_slots(const fdtd_ctx* c) {   // resident workgroups of the update kernels, at the LOWEST occupancy any variant is compiled for / capped to
  int per_cu = std::min(FDTD_E_MINBLOCKS, std::min(FDTD_H_MINBLOCKS, FDTD_WF_MINBLOCKS)) - 1;
  for (int cap : {c->occ_e, c->occ_h, c->occ_wf}) if (cap > 0) per_cu = std::min(per_cu, cap);
  return (unsigned)chip_cus(c->d.device) * (unsigned)std::max(per_cu, 1);
}
static int p2p_shared_device_refuse(fdtd_ctx* c, unsigned pinned, unsigned slabs) {
  return fdtd_fail(c, FDTD_E_UNSUPPORTED, "p2p transport between slabs that share one GPU: %u slabs could pin %u workgroups waiting for each other's halos and the chip holds %u — not starvation-free; use fewer / thicker strips, FDTD_FLAG_KERNEL_DIRECT, or fdtd_link (event-ordered copies)",
                   slabs, pinned, chip_slots(c));
}

static void p2p_prime_if_needed(fdtd_ctx* c);
static int step_loop_wf(fdtd_ctx* c, int nsteps, ProfEvents* pe) {
  HIPCK(c, hipSetDevice(c->d.device));
  hipStream_t s = c->stream;
  // the last launch of a call of at least 16 timesteps is a calibration launch of the XCD shares (kernels.hip: xcd_adapt) —
  // often in the first eight calls of a context, every eighth call after that
  const bool calibrate = !pe && nsteps >= 16 && c->d.world == 1 && (c->xcd_adapt_calls < 8 || (c->xcd_adapt_calls & 7) == 0);
  if (!pe && nsteps >= 16) c->xcd_adapt_calls++;
  // Cache-resident single slabs: SEVERAL timesteps per launch (k_step<.., MULTI>) — up to the next timestep whose NF2FF faces
  // are sampled (a launch of its own reads them), the calibration launch on its own.
  const int multi = wf_multi_max(c);
  const bool sampling = (c->nfreq || c->recorder) && c->nbox && c->every > 0;
  int launches = 0;
  c->wf_mur = c->any_mur;   // (wavefront_possible has checked that this slab can carry them)
  if (c->wf_mur && nsteps > 0 && c->mur_pre_step != c->step) launch_mur(c, 0, s);   // the pre pass of the first timestep (later ones: the H blocks)
  for (int n = 0; n < nsteps;) {
    p2p_prime_if_needed(c);
    int chunk = 1;
    if (multi > 1) {
      chunk = std::min(multi, nsteps - n);
      if (sampling) {   // ... so that the sampled timestep (a multiple of `every`) is the launch's last
        const long long next = (c->step + c->every - 1) / c->every * c->every;
        chunk = (int)std::min<long long>(chunk, next - c->step + 1);
      }
      if (calibrate && n + chunk == nsteps && chunk > 1) chunk -= 1;
    }
    if (pe) { c->kev0 = pe->e0[launches]; c->kev1 = pe->e1[launches]; }
    if (calibrate && n + chunk == nsteps && chunk == 1) { int ra = xcd_stamp_arm(c, s); if (ra) return ra; }
    int r = launch_step_wf(c, c->step, s, chunk);
    c->kev0 = c->kev1 = nullptr;
    if (r) return r;
    c->step += chunk;
    launch_dft(c, -1, c->step - 1, s);       // V and I boxes of the launch's last timestep in ONE launch
    n += chunk;
    ++launches;
  }
  if (pe) pe->launches = launches;
  if (c->wf_mur && nsteps > 0) c->mur_pre_step = c->step;
  HIPCK(c, hipGetLastError());
  return launch_status(c);
}
// after the stream has drained: did a flag wait of the wavefront schedule time out?
static int wf_check(fdtd_ctx* c) {
  if (!c->wf_err) return FDTD_OK;
  int e[8] = {};
  HIPCK(c, hipMemcpy(e, c->wf_err, sizeof(e), hipMemcpyDeviceToHost));
  if (e[0]) {
    hipMemset(c->wf_err, 0, sizeof(e));
    // the record of the first wait that gave up: which flag, what it should have held, what it held
    char what[160] = "";
    const int per_plane = c->p.nstrips * c->p.nbs;
    if (e[5] == 2) snprintf(what, sizeof what, "the 'cells read' word of probe %d", e[1]);
    else if (per_plane > 0) snprintf(what, sizeof what, "the %s flag of plane %d, strip %d, block %d", e[5] ? "H" : "E", e[1] / per_plane, (e[1] % per_plane) / c->p.nbs, e[1] % c->p.nbs);
    return fdtd_fail(c, FDTD_E_DEVICE, "wavefront schedule: rank %d, workgroup %d of the launch waited more than %.3g s for %s to reach %u and found %u (dispatch not in order?); the fields of this run are invalid — re-initialise them and select FDTD_FLAG_KERNEL_DIRECT (simulation.Simulation.run does both by itself)",
                     c->d.rank, e[4], (double)c->p.wf_limit * 1e-8, what, (unsigned)e[2], (unsigned)e[3]);
  }
  return FDTD_OK;
}

static int step_loop(fdtd_ctx* c, int nsteps, ProfEvents* pe) {
  const unsigned sel = c->d.flags & FDTD_FLAG_KERNEL_MASK;
  if (sel == FDTD_FLAG_KERNEL_RESIDENT) {
    const char* why = "";
    if (!res_possible(c, &why)) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "resident schedule: %s", why);
    if (!sources_fusable(c)) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "resident schedule: a source edge lies on or next to a Mur face");
  }
  if (sel > FDTD_FLAG_KERNEL_DIRECT && sel != FDTD_FLAG_KERNEL_WAVEFRONT && sel != FDTD_FLAG_KERNEL_RESIDENT)
    return fdtd_fail(c, FDTD_E_UNSUPPORTED, "kernel selection %u: the one-pass variants were removed (measured slower than the two-pass kernels on every workload)", sel);
  if (sel == FDTD_FLAG_KERNEL_WAVEFRONT && !wavefront_possible(c))
    return fdtd_fail(c, FDTD_E_UNSUPPORTED, "wavefront schedule: single slab or slabs on the p2p mailbox transport, at least 2 planes, rows of at most %d cells; with Mur faces a single slab within the Infinity Cache, no source edge, voltage probe or NF2FF box on or next to a face", 30 * FDTD_BLOCK * 4);
  if (c->p.p2p) return step_loop_p2p(c, nsteps, pe);
  const bool multi = c->d.world > 1;
  if (resident_active(c)) return step_loop_res(c, nsteps, pe);
  if (multi && !c->comm) return fdtd_fail(c, FDTD_E_STATE, "world > 1: call fdtd_p2p_attach (mailbox transport), fdtd_comm_init (RCCL), fdtd_link + fdtd_run_linked, or drive fdtd_half_step + fdtd_halo_*");
  if (wavefront_active(c)) return step_loop_wf(c, nsteps, pe);
  const bool fused = sources_fusable(c);
  if (multi && c->step == 0 && !c->p2p_primed) {   // the H halo of "step -1": the initial fields (runs from non-zero fields decompose too)
    int r = exchange(c, FDTD_HALO_H_UP);
    if (r) return r;
    c->p2p_primed = true;
  }
  const bool direct = mur_direct_possible(c, multi, fused);
  for (int n = 0; n < nsteps; ++n) {
    c->mur_direct = direct;
    int r = phase_E(c, multi, fused, pe, n);
    if (r) return r;
    if (multi && (r = exchange(c, FDTD_HALO_E_DOWN))) return r;
    if ((r = phase_H(c, multi, fused, pe, n))) return r;
    if (multi && (r = exchange(c, FDTD_HALO_H_UP))) return r;
    c->step++;
  }
  c->mur_direct = false;
  if (fused && nsteps > 0) launch_post(c, FDTD_KIND_I, c->step - 1, false, c->stream);   // flush the last step's I-probes
  HIPCK(c, hipGetLastError());
  return launch_status(c);
}

// P2P mailbox transport: the halos travel inside the update kernels, so a step is two launches on ONE stream —
// no communication stream, no events, no RCCL call; neighbouring ranks couple only through the mailbox flags.
// p2p transport, before the first timestep: the halo of "step -1" = the INITIAL Ix, Iy of this slab's top plane goes up
static void p2p_prime_if_needed(fdtd_ctx* c) {
  if (!c->p.p2p || c->step != 0 || c->p2p_primed) return;
  hipSetDevice(c->d.device);
  launch_p2p_prime(c, c->stream);
  c->p2p_primed = true;
}
static int p2p_enqueue_E(fdtd_ctx* c, ProfEvents* pe, int n) {
  HIPCK(c, hipSetDevice(c->d.device));
  p2p_prime_if_needed(c);
  if (pe) { c->kev0 = pe->e0[n]; c->kev1 = pe->e1[n]; }
  launch_update_E(c, 0, c->d.nk, c->step, true, true, c->stream);
  c->kev0 = c->kev1 = nullptr;
  launch_dft(c, FDTD_KIND_V, c->step, c->stream);
  return FDTD_OK;
}
static int p2p_enqueue_H(fdtd_ctx* c, ProfEvents* pe, int n) {
  HIPCK(c, hipSetDevice(c->d.device));
  if (pe) { c->kev0 = pe->h0[n]; c->kev1 = pe->h1[n]; }
  launch_update_H(c, 0, c->d.nk, c->step, true, c->stream);
  c->kev0 = c->kev1 = nullptr;
  launch_dft(c, FDTD_KIND_I, c->step, c->stream);
  return FDTD_OK;
}
static bool wavefront_active(const fdtd_ctx* c);
static int step_loop_wf(fdtd_ctx* c, int nsteps, ProfEvents* pe);
static int step_loop_p2p(fdtd_ctx* c, int nsteps, ProfEvents* pe) {
  if (c->any_mur || c->d.nk < 2) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "p2p transport: needs >= 2 planes per slab and no Mur faces");
  if (!c->p.src_dense_ok && c->src_max_per_strip_plane > FDTD_BLOCK) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "p2p transport: more than %d source edges in one strip-plane with several sources on one edge", FDTD_BLOCK);
  const bool one_launch = wavefront_active(c);
  // a neighbour's slab on THIS device (ranks sharing a GPU): all `world` slabs may be here, under this slab's schedule — the bound above
  if (!(c->d.flags & FDTD_FLAG_LOOPBACK) && (c->link_info[0][7] == 1 || c->link_info[1][7] == 1)) {
    const unsigned pinned = (unsigned)c->d.world * p2p_pinned_blocks(c, one_launch);
    if (pinned >= chip_slots(c)) return p2p_shared_device_refuse(c, pinned, (unsigned)c->d.world);
  }
  if (one_launch) return step_loop_wf(c, nsteps, pe);   // one launch per timestep, halos inside it as well
  for (int n = 0; n < nsteps; ++n) {
    int r;
    if ((r = p2p_enqueue_E(c, pe, n)) || (r = p2p_enqueue_H(c, pe, n))) return r;
    c->step++;
  }
  if (nsteps > 0) launch_post(c, FDTD_KIND_I, c->step - 1, false, c->stream);   // flush the last step's I-probes
  HIPCK(c, hipGetLastError());
  return launch_status(c);
}

int fdtd_run(fdtd_ctx* c, int nsteps) {
  int r = check_ready(c);
  if (r) return r;
  HIPCK(c, hipSetDevice(c->d.device));
  const bool p2p_fault = c->p.p2p && c->p2p_fault_step >= c->step && c->p2p_fault_step < c->step + nsteps;
  const unsigned long long p2p_limit = c->p.p2p_limit;
  if (p2p_fault) { c->p.p2p_tag_bias = 0x40000000u; c->p.p2p_limit = 2000ull; }   // test hook: tags nobody sends, 20 us
  r = step_loop(c, nsteps, nullptr);
  if (p2p_fault) { c->p.p2p_tag_bias = 0u; c->p.p2p_limit = p2p_limit; c->p2p_fault_step = -1; }
  if (r) return r;
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (c->comm_stream) HIPCK(c, hipStreamSynchronize(c->comm_stream));
  if (c->p.p2p) { r = p2p_check(c); if (r) return r; }
  if ((r = wf_check(c))) return r;
  if ((r = res_check(c))) return r;
  return xcd_adapt(c);     // (a no-op unless the call's last launch was a calibration launch of the XCD shares)
}

int fdtd_run_profiled(fdtd_ctx* c, int nsteps, fdtd_profile* out) {
  int r = check_ready(c);
  if (r) return r;
  if (!out || nsteps < 1 || nsteps > 4096) return fdtd_fail(c, FDTD_E_ARG, "profiled run: 1..4096 steps (four events per step)");
  HIPCK(c, hipSetDevice(c->d.device));
  ProfEvents pe;
  auto destroy_all = [&]() {
    for (auto* v : {&pe.e0, &pe.e1, &pe.h0, &pe.h1}) for (auto e : *v) if (e) hipEventDestroy(e);
    if (pe.t0) hipEventDestroy(pe.t0);
    if (pe.t1) hipEventDestroy(pe.t1);
  };
  hipError_t ce = hipSuccess;
  for (auto* v : {&pe.e0, &pe.e1, &pe.h0, &pe.h1}) {
    v->assign(nsteps, nullptr);
    for (auto& e : *v) if (ce == hipSuccess) ce = hipEventCreate(&e);
  }
  if (ce == hipSuccess) ce = hipEventCreate(&pe.t0);
  if (ce == hipSuccess) ce = hipEventCreate(&pe.t1);
  if (ce != hipSuccess) {
    destroy_all();
    return fdtd_fail(c, FDTD_E_DEVICE, "profiled run: hipEventCreate: %s", hipGetErrorString(ce));
  }
  r = FDTD_OK;
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipEventRecord(pe.t0, c->stream);
  if (e == hipSuccess) r = step_loop(c, nsteps, &pe);
  c->kev0 = c->kev1 = nullptr;
  if (e == hipSuccess && r == FDTD_OK) {
    e = hipEventRecord(pe.t1, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && c->comm_stream) e = hipStreamSynchronize(c->comm_stream);
  }
  if (e == hipSuccess && r == FDTD_OK) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, pe.t0, pe.t1);
    memset(out, 0, sizeof(*out));
    out->ms_total = ms; out->steps = nsteps;
    // Every main launch carried its own start / stop events (hipExtLaunchKernelGGL), which take the dispatch's begin
    // and end timestamps — the interval a kernel trace reports, no event-packet time inside, nothing to calibrate away.
    double se = 0, sh = 0;
    const int nl = pe.launches >= 0 ? pe.launches : nsteps;
    for (int n = 0; n < nl; ++n) {
      if (hipEventElapsedTime(&ms, pe.e0[n], pe.e1[n]) == hipSuccess) se += ms;
      if (pe.launches < 0 && hipEventElapsedTime(&ms, pe.h0[n], pe.h1[n]) == hipSuccess) sh += ms;
    }
    (void)hipGetLastError();
    out->ms_event_overhead = 0.0;
    out->ms_update_e = se / nsteps;             // per TIMESTEP (launches of several timesteps: their durations summed, over the timesteps)
    out->ms_update_h = sh / nsteps;
    out->fused = (pe.launches >= 0) ? 1 : 0;   // 1: ms_update_e is the one launch of a whole timestep, ms_update_h = 0
    out->launches_e = nl; out->launches_h = out->fused ? 0 : nsteps;
  }
  destroy_all();
  if (r) return r;
  HIPCK(c, e);
  if ((r = res_check(c))) return r;
  return wf_check(c);
}

int fdtd_get_step(fdtd_ctx* c, int64_t* step) {
  if (!c || !step) return FDTD_E_ARG;
  *step = c->step;
  return FDTD_OK;
}

int fdtd_energy(fdtd_ctx* c, double sums[2]) {
  if (!c || !sums) return FDTD_E_ARG;
  HIPCK(c, hipSetDevice(c->d.device));
  launch_energy(c, c->stream);
  HIPCK(c, hipStreamSynchronize(c->stream));
  HIPCK(c, hipMemcpy(sums, c->d_energy, 2 * sizeof(double), hipMemcpyDeviceToHost));
  return FDTD_OK;
}

// ---- external halo transport --------------------------------------------------------------------
int fdtd_half_step(fdtd_ctx* c, int phase) {
  int r = check_ready(c);
  if (r) return r;
  if (c->p.p2p) return fdtd_fail(c, FDTD_E_STATE, "fdtd_half_step drives an external halo transport; detach the p2p transport first");
  HIPCK(c, hipSetDevice(c->d.device));
  hipStream_t s = c->stream;
  c->mur_pre_step = -1;
  if (phase == FDTD_PHASE_E) {
    launch_mur(c, 0, s);
    launch_update_E(c, 0, c->d.nk, c->step, false, false, s);
    launch_mur(c, 1, s);
    launch_mur(c, 2, s);
    launch_post(c, FDTD_KIND_V, c->step, true, s);
    launch_dft(c, FDTD_KIND_V, c->step, s);
  } else if (phase == FDTD_PHASE_H) {
    launch_update_H(c, 0, c->d.nk, c->step, false, s);
    launch_post(c, FDTD_KIND_I, c->step, false, s);
    launch_dft(c, FDTD_KIND_I, c->step, s);
    c->step++;
  } else return fdtd_fail(c, FDTD_E_ARG, "bad phase");
  HIPCK(c, hipGetLastError());
  HIPCK(c, hipStreamSynchronize(s));
  return launch_status(c);
}

static hipError_t plane_d2h(const fdtd_ctx* c, float* host, const float* dev) {
  return hipMemcpy2D(host, (size_t)c->d.nx * 4, dev, (size_t)c->P * 4, (size_t)c->d.nx * 4, c->d.ny, hipMemcpyDeviceToHost);
}
static hipError_t plane_h2d(const fdtd_ctx* c, float* dev, const float* host) {
  return hipMemcpy2D(dev, (size_t)c->P * 4, host, (size_t)c->d.nx * 4, (size_t)c->d.nx * 4, c->d.ny, hipMemcpyHostToDevice);
}

int fdtd_halo_get(fdtd_ctx* c, int which, float* buf) {
  if (!c || !buf) return FDTD_E_ARG;
  HIPCK(c, hipSetDevice(c->d.device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  const size_t hp = (size_t)c->d.nx * c->d.ny;
  if (which == FDTD_HALO_H_UP) {
    const long top = (long)(c->d.nk - 1) * c->plane;
    HIPCK(c, plane_d2h(c, buf, c->p.I[0] + top));
    HIPCK(c, plane_d2h(c, buf + hp, c->p.I[1] + top));
  } else if (which == FDTD_HALO_E_DOWN) {
    HIPCK(c, plane_d2h(c, buf, c->p.V[0]));
    HIPCK(c, plane_d2h(c, buf + hp, c->p.V[1]));
  } else return fdtd_fail(c, FDTD_E_ARG, "bad halo id");
  return FDTD_OK;
}

int fdtd_halo_put(fdtd_ctx* c, int which, const float* buf) {
  if (!c || !buf) return FDTD_E_ARG;
  HIPCK(c, hipSetDevice(c->d.device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  const size_t hp = (size_t)c->d.nx * c->d.ny;
  if (which == FDTD_HALO_H_UP) {
    HIPCK(c, plane_h2d(c, c->p.I[0] - c->plane, buf));
    HIPCK(c, plane_h2d(c, c->p.I[1] - c->plane, buf + hp));
  } else if (which == FDTD_HALO_E_DOWN) {
    HIPCK(c, plane_h2d(c, c->p.V[0] + c->nloc, buf));
    HIPCK(c, plane_h2d(c, c->p.V[1] + c->nloc, buf + hp));
  } else return fdtd_fail(c, FDTD_E_ARG, "bad halo id");
  return FDTD_OK;
}

// ---- in-process transport: several slabs driven by one host thread ---------------------------------------
// ---- P2P mailbox transport ----------------------------------------------------------------------------------
// Mailbox of a context: [E: 2 parities x 2 comps x 2*plane][H: the same] words (8-byte granules {value, tag}), then 64
// control words: [4] error word (the rest is unused since the granule protocol: no flags, no arrival counters).
__global__ void k_wallclock(unsigned long long* out) { *out = (unsigned long long)wall_clock64(); }

struct P2pBlob { hipIpcMemHandle_t h; uint64_t bytes; uint64_t raw; int32_t pid, device, nx, ny; char busid[16]; };
static_assert(sizeof(P2pBlob) <= 128, "blob must fit the 128-byte exchange buffer");

static size_t p2p_floats(const fdtd_ctx* c) { return (size_t)16 * c->plane; }

// The mailbox is only ever touched with system-scope (write-through / cache-bypassing) accesses — also when it is zeroed,
// so that no plain fill leaves copies of its lines in some XCD's L2 for a later mailbox load to find.
__global__ __launch_bounds__(FDTD_BLOCK) void k_p2p_zero(float* mbox, const size_t n4, unsigned* ctl, const int nctl) {
  const size_t t = (size_t)blockIdx.x * FDTD_BLOCK + threadIdx.x;
  if (t < n4) st4_sys(mbox + 4 * t, make_float4(0.f, 0.f, 0.f, 0.f));
  if (ctl && t < (size_t)nctl) __hip_atomic_store(ctl + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__global__ void k_p2p_clear_err(int* err) { for (int q = 7; q >= 0; --q) __hip_atomic_store(err + q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // error word + record
// zero the halo planes (and, with `ctl`, the 64 control words) of this context's own mailbox, on its stream
static void p2p_zero(fdtd_ctx* c, bool ctl) {
  const size_t n4 = p2p_floats(c) / 4;      // 16-byte groups of the halo part
  hipLaunchKernelGGL(k_p2p_zero, dim3((unsigned)((n4 + FDTD_BLOCK - 1) / FDTD_BLOCK)), dim3(FDTD_BLOCK), 0, c->stream,
                     (float*)c->mbox, n4, ctl ? (unsigned*)((float*)c->mbox + p2p_floats(c)) : nullptr, 64);
}

static int p2p_alloc(fdtd_ctx* c) {
  if (c->mbox) return FDTD_OK;
  HIPCK(c, hipSetDevice(c->d.device));
  c->mbox_bytes = p2p_floats(c) * sizeof(float) + 64 * sizeof(unsigned);
  // Fine-grained (system-coherent) device memory: another GPU writes it while this GPU's kernels poll and read it, so
  // it must never sit stale in this GPU's L2 (coarse-grained memory is only coherent at kernel boundaries).
  if (getenv("FDTD_P2P_COARSE") || hipExtMallocWithFlags(&c->mbox, c->mbox_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    c->mbox = nullptr;
    HIPCK(c, hipMalloc(&c->mbox, c->mbox_bytes));
    c->mbox_fine = false;
  } else {
    c->mbox_fine = true;
  }
  if (getenv("FDTD_P2P_DEBUG")) fprintf(stderr, "[fdtd-hip] rank %d mailbox: %zu bytes, %s device memory\n", c->d.rank, c->mbox_bytes, c->mbox_fine ? "fine-grained" : "coarse-grained");
  p2p_zero(c, true);
  HIPCK(c, hipStreamSynchronize(c->stream));   // the zeros must be there before a neighbour maps and writes the mailbox
  return FDTD_OK;
}

static void p2p_views(const fdtd_ctx* c, void* base, float** in_E, float** in_H, unsigned** ctl) {
  float* f = (float*)base;
  *in_E = f; *in_H = f + 8 * (size_t)c->plane; *ctl = (unsigned*)(f + p2p_floats(c));
}

int fdtd_p2p_export(fdtd_ctx* c, void* out128) {
  if (!c || !out128) return FDTD_E_ARG;
  int r = p2p_alloc(c);
  if (r) return r;
  P2pBlob b{};
  HIPCK(c, hipIpcGetMemHandle(&b.h, c->mbox));
  b.bytes = c->mbox_bytes; b.raw = (uint64_t)(uintptr_t)c->mbox; b.pid = (int32_t)getpid(); b.device = c->d.device;
  b.nx = c->d.nx; b.ny = c->d.ny;
  if (hipDeviceGetPCIBusId(b.busid, (int)sizeof b.busid, c->d.device) != hipSuccess) { (void)hipGetLastError(); b.busid[0] = 0; }
  b.busid[sizeof b.busid - 1] = 0;
  memset(out128, 0, 128);
  memcpy(out128, &b, sizeof(b));
  return FDTD_OK;
}

// what the runtime knows about the way from this context's GPU to the GPU a neighbour's mailbox lives on (fdtd_p2p_link_info)
static void p2p_note_link(fdtd_ctx* c, int which, const P2pBlob& b, bool ipc) {
  int32_t* info = c->link_info[which];
  for (int q = 0; q < 8; ++q) info[q] = -1;
  info[1] = ipc ? 2 : 1;
  int nb = -1;
  if (!ipc) nb = b.device;                                   // same process: same ordinals
  else if (b.busid[0] && hipDeviceGetByPCIBusId(&nb, b.busid) != hipSuccess) { (void)hipGetLastError(); nb = -1; }
  info[0] = nb >= 0 ? nb : -2;
  if (nb < 0) return;
  info[7] = nb == c->d.device ? 1 : 0;
  if (nb == c->d.device) { info[3] = 0; return; }
  uint32_t type = 0, hops = 0;
  if (hipExtGetLinkTypeAndHopCount(c->d.device, nb, &type, &hops) == hipSuccess) { info[2] = (int32_t)type; info[3] = (int32_t)hops; }
  int v = 0;
  if (hipDeviceGetP2PAttribute(&v, hipDevP2PAttrPerformanceRank, c->d.device, nb) == hipSuccess) info[4] = v;
  if (hipDeviceGetP2PAttribute(&v, hipDevP2PAttrAccessSupported, c->d.device, nb) == hipSuccess) info[5] = v;
  if (hipDeviceGetP2PAttribute(&v, hipDevP2PAttrNativeAtomicSupported, c->d.device, nb) == hipSuccess) info[6] = v;
  (void)hipGetLastError();
}

static int p2p_open(fdtd_ctx* c, int which, const void* blob128, void** out, bool* ipc) {
  P2pBlob b;
  memcpy(&b, blob128, sizeof(b));
  b.busid[sizeof b.busid - 1] = 0;
  if (b.nx != c->d.nx || b.ny != c->d.ny || b.bytes != c->mbox_bytes) return fdtd_fail(c, FDTD_E_ARG, "p2p: neighbour mailbox belongs to another grid");
  p2p_note_link(c, which, b, b.pid != (int32_t)getpid());
  if (b.pid == (int32_t)getpid()) {   // same process: the pointer itself — on another device only through peer access
    if (b.device != c->d.device) {
      int can = 0;
      HIPCK(c, hipDeviceCanAccessPeer(&can, c->d.device, b.device));
      if (!can) return fdtd_fail(c, FDTD_E_UNSUPPORTED, "p2p: device %d cannot access device %d (no peer path); use the rccl or host transport", c->d.device, b.device);
      const int pair[2][2] = {{c->d.device, b.device}, {b.device, c->d.device}};
      for (int q = 0; q < 2; ++q) {
        HIPCK(c, hipSetDevice(pair[q][0]));
        const hipError_t e = hipDeviceEnablePeerAccess(pair[q][1], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
          (void)hipGetLastError();
          hipSetDevice(c->d.device);
          return fdtd_fail(c, FDTD_E_UNSUPPORTED, "p2p: enabling peer access %d -> %d failed: %s", pair[q][0], pair[q][1], hipGetErrorString(e));
        }
        (void)hipGetLastError();
      }
      HIPCK(c, hipSetDevice(c->d.device));
    }
    *out = (void*)(uintptr_t)b.raw; *ipc = false;
    return FDTD_OK;
  }
  HIPCK(c, hipIpcOpenMemHandle(out, b.h, hipIpcMemLazyEnablePeerAccess));
  *ipc = true;
  return FDTD_OK;
}

// lower128 / upper128: blobs exported by ranks rank-1 / rank+1 (null where there is no such neighbour)
int fdtd_p2p_attach(fdtd_ctx* c, const void* lower128, const void* upper128) {
  if (!c) return FDTD_E_ARG;
  if (c->d.world < 2) return fdtd_fail(c, FDTD_E_ARG, "p2p transport needs world > 1");
  // FDTD_FLAG_LOOPBACK: a slab timed alone on one GPU — both blobs are its own; it pulls what a slab in its place pulls and
  // pushes both halo planes into its own mailbox (include/fdtd_hip.h)
  const bool echo = (c->d.flags & FDTD_FLAG_LOOPBACK) != 0;
  if (echo && (!lower128 || !upper128 || memcmp(lower128, upper128, 128) != 0))
    return fdtd_fail(c, FDTD_E_ARG, "p2p loopback: pass this context's own blob for both neighbours");
  if (!echo && ((c->d.rank > 0) != (lower128 != nullptr) || (c->d.rank < c->d.world - 1) != (upper128 != nullptr)))
    return fdtd_fail(c, FDTD_E_ARG, "p2p: rank %d of %d needs exactly its existing neighbours' blobs", c->d.rank, c->d.world);
  if (c->comm || c->link_lo || c->link_hi) return fdtd_fail(c, FDTD_E_STATE, "p2p: another halo transport is already attached");
  // the mailbox flags count steps from zero (a kernel waits for flag >= step): a context that has already stepped
  // would wait 10 s for halos nobody sends and then go on with an empty mailbox
  if (c->step != 0) return fdtd_fail(c, FDTD_E_STATE, "p2p: attach before the first timestep (context is at step %lld)", (long long)c->step);
  int r = p2p_alloc(c);
  if (r) return r;
  HIPCK(c, hipSetDevice(c->d.device));
  if (lower128 && (r = p2p_open(c, 0, lower128, &c->peer_lo, &c->peer_lo_ipc))) return r;
  if (upper128 && (r = p2p_open(c, 1, upper128, &c->peer_hi, &c->peer_hi_ipc))) return r;
  float *in_E, *in_H; unsigned* ctl;
  p2p_views(c, c->mbox, &in_E, &in_H, &ctl);
  DevParams& p = c->p;
  p.mb_in_E = c->d.rank < c->d.world - 1 ? in_E : nullptr;
  p.mb_in_H = c->d.rank > 0 ? in_H : nullptr;
  p.p2p_err = (int*)(ctl + 4);
  p.mb_out_E = nullptr; p.mb_out_H = nullptr;
  if (c->peer_lo) { float *e, *h; unsigned* f; p2p_views(c, c->peer_lo, &e, &h, &f); p.mb_out_E = e; }
  if (c->peer_hi) { float *e, *h; unsigned* f; p2p_views(c, c->peer_hi, &e, &h, &f); p.mb_out_H = h; }
  // wall_clock64() tick rate: measured against the host clock (the attribute is not reliable on every part)
  {
    unsigned long long* d_t = nullptr;
    HIPCK(c, hipMalloc(&d_t, 2 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_wallclock, dim3(1), dim3(1), 0, c->stream, d_t);
    HIPCK(c, hipStreamSynchronize(c->stream));
    const auto h0 = std::chrono::steady_clock::now();
    usleep(20000);
    hipLaunchKernelGGL(k_wallclock, dim3(1), dim3(1), 0, c->stream, d_t + 1);
    HIPCK(c, hipStreamSynchronize(c->stream));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
    unsigned long long t[2];
    HIPCK(c, hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost));
    hipFree(d_t);
    double hz = (double)(t[1] - t[0]) / dt;
    if (getenv("FDTD_P2P_DEBUG")) fprintf(stderr, "[fdtd-hip] wall_clock64: %llu -> %llu in %.4f s = %.3e Hz\n", t[0], t[1], dt, hz);
    if (!(hz > 1e6 && hz < 1e11)) hz = 1e8;
    p.p2p_limit = (unsigned long long)(hz * 10.0);
  }
  p.p2p_dep_first = getenv("FDTD_P2P_DEP_LAST") ? 0 : 1;
  p.p2p = 1;
  return FDTD_OK;
}

// Self-test over the attached mailboxes, THROUGH THE DATA PATH of the update kernels: every rank fills both parities of
// both halo components in its neighbours' mailboxes with a token-derived pattern through the very function the update
// kernels push with (mb_push: 16-byte write-through stores of {value, tag} granules, tag = token); then every thread polls
// ITS groups of its own mailbox with the kernels' loads until all tags are the token (bounded) and compares every value.
// A link on which stores tear below 8 bytes, vanish or land in the wrong place fails here instead of corrupting halos.
// Two launches (post, then poll + verify), so that a grid larger than the chip can never wait on its own blocks.
// Call on all ranks at about the same time.
__device__ __forceinline__ float4 p2p_pattern(const unsigned token, const unsigned dir, const unsigned slot, const unsigned t) {
  const unsigned h = (token * 2654435761u) ^ (dir * 0x9E3779B9u) ^ (slot * 0x85EBCA6Bu) ^ (t * 0xC2B2AE35u);
  // finite float bit patterns only (exponent field forced into the normal range): the words are never computed on
  const unsigned m = 0x807FFFFFu, e = 0x3F000000u;
  return make_float4(__uint_as_float(((h) & m) | e), __uint_as_float(((h * 3u + 1u) & m) | e),
                     __uint_as_float(((h * 5u + 2u) & m) | e), __uint_as_float(((h * 7u + 3u) & m) | e));
}

__global__ __launch_bounds__(FDTD_BLOCK) void k_p2p_selftest_post(const DevParams p, const unsigned token) {
  const unsigned t = blockIdx.x * FDTD_BLOCK + threadIdx.x, n4 = (unsigned)p.plane / 4u;
  if (t >= n4) return;
  for (unsigned slot = 0; slot < 4u; ++slot) {   // slot = parity * 2 + component
    if (p.mb_out_E) mb_push(p.mb_out_E + slot * mb_slot_words(p), 4u * t, token, p2p_pattern(token, 0u, slot, t));
    if (p.mb_out_H) mb_push(p.mb_out_H + slot * mb_slot_words(p), 4u * t, token, p2p_pattern(token, 1u, slot, t));
  }
}

// result[0]: waves that gave up waiting for the neighbour's granules; result[1]: 16-byte groups that read back wrong;
// result[2]: index (slot * n4 + t, +2^30 for the H mailbox) of one wrong group, result[3]: its first word as read
__global__ __launch_bounds__(FDTD_BLOCK) void k_p2p_selftest_verify(const DevParams p, const unsigned token, unsigned* result, int* err) {
  const unsigned t0 = blockIdx.x * FDTD_BLOCK + threadIdx.x, n4 = (unsigned)p.plane / 4u;
  const unsigned t = t0 < n4 ? t0 : 0u;   // lanes beyond the plane pull group 0 (mb_pull2 needs whole waves)
  unsigned bad = 0;
  for (unsigned pair = 0; pair < 2u; ++pair) {   // the two components of one parity at a time, as the kernels pull them
    for (int side = 0; side < 2; ++side) {
      const float* in = side == 0 ? p.mb_in_E : p.mb_in_H;   // written by the upper neighbour as ITS E-down halo (dir 0) / by the lower as ITS H-up halo (dir 1)
      if (!in) continue;
      float4 ga, gb;
      mb_pull2(in + (2u * pair) * mb_slot_words(p), in + (2u * pair + 1u) * mb_slot_words(p), 4u * t, token, ga, gb, err, p.p2p_limit);
      for (unsigned q = 0; q < 2u; ++q) {
        const float4 g = q ? gb : ga, w = p2p_pattern(token, (unsigned)side, 2u * pair + q, t);
        const unsigned b = (__float_as_uint(g.x) != __float_as_uint(w.x)) | (__float_as_uint(g.y) != __float_as_uint(w.y)) |
                           (__float_as_uint(g.z) != __float_as_uint(w.z)) | (__float_as_uint(g.w) != __float_as_uint(w.w));
        if (b && t0 < n4) { result[2] = ((unsigned)side << 30) + (2u * pair + q) * n4 + t; result[3] = __float_as_uint(g.x); bad += b; }
      }
    }
  }
  if (bad) atomicAdd(result + 1, bad);
  if (threadIdx.x == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) atomicAdd(result + 0, 1u);
}

int fdtd_p2p_selftest(fdtd_ctx* c, unsigned token) {
  if (!c) return FDTD_E_ARG;
  if (!c->p.p2p) return fdtd_fail(c, FDTD_E_STATE, "p2p transport not attached");
  if (c->step != 0) return fdtd_fail(c, FDTD_E_STATE, "p2p self-test overwrites the mailboxes: run it before the first timestep");
  if (token == 0u) return fdtd_fail(c, FDTD_E_ARG, "p2p self-test token must be non-zero");
  HIPCK(c, hipSetDevice(c->d.device));
  unsigned* d_res = nullptr;
  HIPCK(c, hipMalloc(&d_res, 16 * sizeof(unsigned)));                       // [0..3] results, [4..11] the self-test's own error word + record
  HIPCK(c, hipMemsetAsync(d_res, 0, 16 * sizeof(unsigned), c->stream));   // the context's stream is non-blocking: keep everything on it
  int* d_err = reinterpret_cast<int*>(d_res + 4);                          // the self-test's own error word (a failed test must not poison the run's)
  const unsigned nb = (unsigned)((c->plane / 4 + FDTD_BLOCK - 1) / FDTD_BLOCK);
  hipLaunchKernelGGL(k_p2p_selftest_post, dim3(nb), dim3(FDTD_BLOCK), 0, c->stream, c->p, token);
  hipLaunchKernelGGL(k_p2p_selftest_verify, dim3(nb), dim3(FDTD_BLOCK), 0, c->stream, c->p, token, d_res, d_err);
  unsigned res[4] = {0, 0, 0, 0};
  hipError_t e = hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, c->stream);
  // every granule of the neighbours' patterns has been seen in THIS mailbox (or the test has failed): restore the zeros
  // (tag 0 = no halo: the first timestep's halo of "step -1" is pushed by k_p2p_prime)
  if (e == hipSuccess) p2p_zero(c, false);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_res);
  HIPCK(c, e);
  if (res[0])
    return fdtd_fail(c, FDTD_E_DEVICE, "p2p self-test: a neighbour's halo granules did not arrive within the time limit (expected tag %#x; %u block(s) gave up)", token, res[0]);
  if (res[1])
    return fdtd_fail(c, FDTD_E_DEVICE, "p2p self-test: %u of %u 16-byte halo groups read back wrong although their tags arrived (e.g. group %u of the %s mailbox reads %#x; %s device memory): payload stores do not reach this mailbox intact",
                     res[1], (unsigned)(c->plane / 4) * 4u * ((c->p.mb_in_E ? 1u : 0u) + (c->p.mb_in_H ? 1u : 0u)),
                     res[2] & ((1u << 30) - 1u), (res[2] >> 30) ? "H" : "E", res[3], c->mbox_fine ? "fine-grained" : "coarse-grained");
  return FDTD_OK;
}

// Drop the mailbox transport again (e.g. to fall back to RCCL after a failed self-test).
int fdtd_p2p_detach(fdtd_ctx* c) {
  if (!c) return FDTD_E_ARG;
  HIPCK(c, hipSetDevice(c->d.device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (c->peer_lo && c->peer_lo_ipc) hipIpcCloseMemHandle(c->peer_lo);
  if (c->peer_hi && c->peer_hi_ipc) hipIpcCloseMemHandle(c->peer_hi);
  c->peer_lo = c->peer_hi = nullptr; c->peer_lo_ipc = c->peer_hi_ipc = false;
  for (int w = 0; w < 2; ++w) for (int q = 0; q < 8; ++q) c->link_info[w][q] = -1;
  DevParams& p = c->p;
  p.p2p = 0; p.mb_in_E = p.mb_in_H = p.mb_out_E = p.mb_out_H = nullptr;
  if (c->mbox) { p2p_zero(c, true); HIPCK(c, hipStreamSynchronize(c->stream)); }
  return FDTD_OK;
}

static int p2p_check(fdtd_ctx* c) {
  int err[8] = {};
  HIPCK(c, hipMemcpy(err, c->p.p2p_err, sizeof(err), hipMemcpyDeviceToHost));
  if (err[0]) {
    // reported once: cleared (with a system-scope store, like every access to the mailbox) so that a later run on this
    // context is judged on its own — the fields of THIS run are invalid
    hipLaunchKernelGGL(k_p2p_clear_err, dim3(1), dim3(1), 0, c->stream, c->p.p2p_err);
    hipStreamSynchronize(c->stream);
    // the record the first wait to give up left (kernel_common.hpp: mb_pull2): who waited, for which tag, and what it found
    const unsigned who = (unsigned)err[1], want = (unsigned)err[2], seen = (unsigned)err[3];
    const bool h_half = (who >> 28) & 1u, one_launch = (who >> 29) & 1u;
    const int from = h_half ? c->d.rank + 1 : c->d.rank - 1;
    return fdtd_fail(c, FDTD_E_DEVICE, "p2p: a halo wait timed out: rank %d, %s half-step of timestep %u (%s), plane %d, strip %u, block %u, cell group %d awaited tag %u from rank %d and found tag %u (%s) — neighbour rank not stepping, starved of the chip by spinning workgroups, or peer memory not visible; the fields of this run are invalid",
                     c->d.rank, h_half ? "H" : "E", want - 1u, one_launch ? "one launch per timestep" : "two launches per timestep", h_half ? c->d.nk - 1 : 0,
                     (who >> 14) & 0x3FFFu, who & 0x3FFFu, err[4], want, from, seen,
                     seen == 0u ? "nothing ever arrived" : seen < want ? "the neighbour is behind: its halo of an earlier timestep" : "a later timestep: the slot was overwritten before it was read");
  }
  return FDTD_OK;
}

int fdtd_p2p_link_info(fdtd_ctx* c, int which, int32_t info[8]) {
  if (!c || !info || (which != 0 && which != 1)) return fdtd_fail(c, FDTD_E_ARG, "bad link query");
  const bool attached = which == 0 ? c->peer_lo != nullptr : c->peer_hi != nullptr;
  for (int q = 0; q < 8; ++q) info[q] = attached ? c->link_info[which][q] : -1;
  return FDTD_OK;
}

int fdtd_schedule_info(fdtd_ctx* c, int32_t info[8]) {
  if (!c || !info) return fdtd_fail(c, FDTD_E_ARG, "null argument");
  for (int q = 0; q < 8; ++q) info[q] = 0;
  const bool multi = c->d.world > 1;
  const bool steppable = c->have_op && (!multi || c->p.p2p || c->comm || c->link_lo || c->link_hi);
  const bool res = steppable && resident_active(c);
  const bool wf = steppable && !res && wavefront_active(c);
  info[0] = !steppable ? 0 : res ? 1 : wf ? 1 : !c->any_mur ? 2 : mur_direct_possible(c, multi, sources_fusable(c)) ? 2 : (sources_fusable(c) && !multi && c->mur_fuse_post) ? 3 : 5;
  info[1] = res ? -1 : wf ? wf_lag_for(c) : 0;
  info[2] = c->p.tys;
  info[3] = c->d.nk * c->p.nstrips * c->p.nbs;
  info[4] = !multi ? 0 : c->p.p2p ? 1 : c->comm ? 2 : (c->link_lo || c->link_hi) ? 3 : 4;
  info[5] = (c->xcd_balance && c->have_cpml) ? 1 : 0;
  info[6] = c->xcd_adapt_done;
  info[7] = res ? c->res_chunk : wf ? wf_multi_max(c) : 0;
  if (res) {   // tiles instead of strip blocks
    if (res_prepare(c, c->res_chunk) == FDTD_OK) { info[2] = c->res.nstrips; info[3] = c->res.nblocks; }
  }
  return FDTD_OK;
}

int fdtd_link(fdtd_ctx* lower, fdtd_ctx* upper) {
  if (!lower || !upper) return FDTD_E_ARG;
  if (lower == upper) {   // FDTD_FLAG_LOOPBACK: an interior slab linked to itself (transport self-test / timing)
    fdtd_ctx* c = lower;
    if (!(c->d.flags & FDTD_FLAG_LOOPBACK) || c->d.rank == 0 || c->d.rank == c->d.world - 1)
      return fdtd_fail(c, FDTD_E_ARG, "fdtd_link(c, c) needs FDTD_FLAG_LOOPBACK on an interior slab");
    c->link_lo = c->link_hi = c;
    return FDTD_OK;
  }
  if (lower->d.world != upper->d.world || upper->d.rank != lower->d.rank + 1 || lower->d.k0 + lower->d.nk != upper->d.k0 ||
      lower->d.nx != upper->d.nx || lower->d.ny != upper->d.ny)
    return fdtd_fail(lower, FDTD_E_ARG, "fdtd_link: contexts are not adjacent slabs of one grid");
  if (lower->d.device != upper->d.device) {
    int can = 0;
    HIPCK(lower, hipDeviceCanAccessPeer(&can, lower->d.device, upper->d.device));
    if (can) {
      hipSetDevice(lower->d.device); hipDeviceEnablePeerAccess(upper->d.device, 0);
      hipSetDevice(upper->d.device); hipDeviceEnablePeerAccess(lower->d.device, 0);
      (void)hipGetLastError();   // "already enabled" is fine
    }
  }
  lower->link_hi = upper; upper->link_lo = lower;
  return FDTD_OK;
}

int fdtd_run_linked(fdtd_ctx** ctxs, int n, int nsteps) {
  if (!ctxs || n < 1) return FDTD_E_ARG;
  for (int r = 0; r < n; ++r) {
    fdtd_ctx* c = ctxs[r];
    int rc = check_ready(c);
    if (rc) return rc;
    const bool loop = n == 1 && (c->d.flags & FDTD_FLAG_LOOPBACK) && c->link_lo == c && c->link_hi == c;
    if (loop) { if (c->comm) return fdtd_fail(c, FDTD_E_ARG, "loopback: linked and RCCL transports are exclusive"); continue; }
    if (c->d.world != n || c->d.rank != r || c->comm) return fdtd_fail(c, FDTD_E_ARG, "fdtd_run_linked: contexts must be ranks 0..n-1 of a world of n without an RCCL communicator");
    if (!c->p.p2p && ((r > 0 && c->link_lo != ctxs[r - 1]) || (r < n - 1 && c->link_hi != ctxs[r + 1]))) return fdtd_fail(c, FDTD_E_STATE, "fdtd_run_linked: call fdtd_link (or fdtd_p2p_attach) on every adjacent pair first");
  }
  const bool multi = n > 1 || (ctxs[0]->d.flags & FDTD_FLAG_LOOPBACK);
  if (ctxs[0]->p.p2p) {   // mailbox transport between contexts of this process: interleave the ranks' launches
    for (int r = 0; r < n; ++r) if (!ctxs[r]->p.p2p || ctxs[r]->any_mur) return fdtd_fail(ctxs[r], FDTD_E_STATE, "fdtd_run_linked: every context must use the p2p transport (no Mur)");
    for (int r = 0; r < n; ++r) p2p_prime_if_needed(ctxs[r]);   // every rank's initial halo is on its way before any rank's first launch
    // Slabs of different size may step under different schedules (AUTO: one launch per timestep only above a block count).
    // Submission order of a timestep, TOP rank first: the one launch of a one-launch slab / the E launch of a two-launch
    // slab; then, bottom-up, the H launches of the two-launch slabs.  Streams of one process may share a hardware queue,
    // where a launch waits for the one submitted before it, so whatever a launch waits for on the device must have been
    // submitted EARLIER: H of a slab's top plane (inside its one launch, or its H launch) needs the E blocks of plane 0 of
    // the rank above for the SAME step — submitted before it in the first pass; E of plane 0 needs the H halo of the rank
    // below of the PREVIOUS step.  (Bottom rank first timed out in exactly that way.)
    std::vector<char> wf((size_t)n);
    for (int r = 0; r < n; ++r) wf[r] = wavefront_active(ctxs[r]) ? 1 : 0;
    // slabs of this process that share a device: the starvation-freedom bound (p2p_pinned_blocks), with every slab's own schedule
    for (int r = 0; r < n; ++r) {
      unsigned pinned = 0, slabs = 0;
      for (int q = 0; q < n; ++q) if (ctxs[q]->d.device == ctxs[r]->d.device) { pinned += p2p_pinned_blocks(ctxs[q], wf[q] != 0); ++slabs; }
      if (slabs > 1 && pinned >= chip_slots(ctxs[r])) return p2p_shared_device_refuse(ctxs[r], pinned, slabs);
    }
    for (int s = 0; s < nsteps; ++s) {
      int rc;
      for (int r = n - 1; r >= 0; --r) { if ((rc = wf[r] ? step_loop_wf(ctxs[r], 1, nullptr) : p2p_enqueue_E(ctxs[r], nullptr, 0))) return rc; }
      for (int r = 0; r < n; ++r) if (!wf[r]) { if ((rc = p2p_enqueue_H(ctxs[r], nullptr, 0))) return rc; ctxs[r]->step++; }
    }
    for (int r = 0; r < n; ++r) {
      fdtd_ctx* c = ctxs[r];
      if (wf[r]) continue;
      HIPCK(c, hipSetDevice(c->d.device));
      if (nsteps > 0) launch_post(c, FDTD_KIND_I, c->step - 1, false, c->stream);
    }
    for (int r = 0; r < n; ++r) {
      fdtd_ctx* c = ctxs[r];
      HIPCK(c, hipSetDevice(c->d.device));
      HIPCK(c, hipStreamSynchronize(c->stream));
      int rc = p2p_check(c);
      if (rc) return rc;
      if ((rc = wf_check(c))) return rc;
      if ((rc = launch_status(c))) return rc;
    }
    return FDTD_OK;
  }
  if (multi)   // the H halo of "step -1": the initial fields
    for (int r = 0; r < n; ++r) {
      fdtd_ctx* c = ctxs[r];
      if (c->step != 0 || c->p2p_primed) continue;
      int rc = exchange(c, FDTD_HALO_H_UP);
      if (rc) return rc;
      c->haloH_issued = true; c->p2p_primed = true;
    }
  for (int s = 0; s < nsteps; ++s) {
    int rc;
    for (int r = 0; r < n; ++r) if ((rc = phase_E(ctxs[r], multi, sources_fusable(ctxs[r]), nullptr, 0))) return rc;
    if (multi) for (int r = 0; r < n; ++r) { if ((rc = exchange(ctxs[r], FDTD_HALO_E_DOWN))) return rc; ctxs[r]->haloE_issued = true; }
    for (int r = 0; r < n; ++r) if ((rc = phase_H(ctxs[r], multi, sources_fusable(ctxs[r]), nullptr, 0))) return rc;
    if (multi) for (int r = 0; r < n; ++r) { if ((rc = exchange(ctxs[r], FDTD_HALO_H_UP))) return rc; ctxs[r]->haloH_issued = true; }
    for (int r = 0; r < n; ++r) ctxs[r]->step++;
  }
  for (int r = 0; r < n; ++r) {
    fdtd_ctx* c = ctxs[r];
    HIPCK(c, hipSetDevice(c->d.device));
    if (sources_fusable(c) && nsteps > 0) launch_post(c, FDTD_KIND_I, c->step - 1, false, c->stream);
    HIPCK(c, hipGetLastError());
  }
  for (int r = 0; r < n; ++r) {
    fdtd_ctx* c = ctxs[r];
    HIPCK(c, hipSetDevice(c->d.device));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIPCK(c, hipStreamSynchronize(c->comm_stream));
    if (int rc = launch_status(c)) return rc;
  }
  return FDTD_OK;
}

// ---- RCCL transport -----------------------------------------------------------------------------
int fdtd_comm_unique_id(void* out128) {
  if (!out128) return fdtd_fail(nullptr, FDTD_E_ARG, "null id buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCLCK(nullptr, ncclGetUniqueId(&id));
  memcpy(out128, &id, 128);
  return FDTD_OK;
}

int fdtd_comm_init(fdtd_ctx* c, const void* uid128) {
  if (!c || !uid128) return fdtd_fail(c, FDTD_E_ARG, "null argument");
  if (c->comm) return fdtd_fail(c, FDTD_E_STATE, "communicator already initialised");
  if (c->p.p2p) return fdtd_fail(c, FDTD_E_STATE, "the p2p transport is attached; detach it before creating an RCCL communicator");
  HIPCK(c, hipSetDevice(c->d.device));
  ncclUniqueId id;
  memcpy(&id, uid128, 128);
  ncclComm_t comm;
  if (c->d.flags & FDTD_FLAG_LOOPBACK) {
    if (c->d.rank == 0 || c->d.rank == c->d.world - 1) return fdtd_fail(c, FDTD_E_ARG, "loopback self-test needs an interior slab (0 < rank < world-1)");
    NCCLCK(c, ncclCommInitRank(&comm, 1, id, 0));
  } else {
    NCCLCK(c, ncclCommInitRank(&comm, c->d.world, id, c->d.rank));
  }
  c->comm = comm;
  return FDTD_OK;
}

int fdtd_comm_nranks(fdtd_ctx* c, int* nranks) {
  if (!c || !nranks) return fdtd_fail(c, FDTD_E_ARG, "null argument");
  *nranks = 0;
  if (c->comm) NCCLCK(c, ncclCommCount((ncclComm_t)c->comm, nranks));
  return FDTD_OK;
}

// ---- field access -------------------------------------------------------------------------------
int fdtd_get_field(fdtd_ctx* c, int kind, int comp, float* out) {
  if (!c || !out || comp < 0 || comp > 2 || (kind != 0 && kind != 1)) return fdtd_fail(c, FDTD_E_ARG, "bad field id");
  HIPCK(c, hipSetDevice(c->d.device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  const float* src = kind == FDTD_KIND_V ? c->p.V[comp] : c->p.I[comp];
  HIPCK(c, hipMemcpy2D(out, (size_t)c->d.nx * 4, src, (size_t)c->P * 4, (size_t)c->d.nx * 4, (size_t)c->d.nk * c->d.ny,
                       hipMemcpyDeviceToHost));
  return FDTD_OK;
}

int fdtd_set_field(fdtd_ctx* c, int kind, int comp, const float* in) {
  if (!c || !in || comp < 0 || comp > 2 || (kind != 0 && kind != 1)) return fdtd_fail(c, FDTD_E_ARG, "bad field id");
  HIPCK(c, hipSetDevice(c->d.device));
  HIPCK(c, hipStreamSynchronize(c->stream));
  c->mur_pre_step = -1;   // a Mur pre pass computed on the old voltages is void
  c->p2p_primed = false;  // ... and so is an initial halo pushed from the old fields
  float* dst = kind == FDTD_KIND_V ? c->p.V[comp] : c->p.I[comp];
  HIPCK(c, hipMemcpy2D(dst, (size_t)c->P * 4, in, (size_t)c->d.nx * 4, (size_t)c->d.nx * 4, (size_t)c->d.nk * c->d.ny,
                       hipMemcpyHostToDevice));
  return FDTD_OK;
}

}  // extern "C"
