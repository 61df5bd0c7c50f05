/*
 * fdtd_oracle.c — CPU restatement of the EC-FDTD time-stepping path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (fdtd-solver-antennas_amd/) never does and fails loudly without libfdtd_hip.so.
 *
 * PARITY UNPINNED against openEMS: the arithmetic of this path is not in /root/reference — it
 * lives in the third-party, un-vendored, un-pinned openEMS/CSXCAD install that the reference
 * imports at run time (antenna_sim/solver_fdtd_openems_fixed.py:131-133) and calls at
 *   FDTD.Run(...)        solver_fdtd_openems_fixed.py:280, _microstrip.py:401, _microstrip_3d.py:214,
 *                        _microstrip_multi_3d.py:610, solver_fdtd_openems.py:289
 *   nf2ff.CalcNF2FF(...) solver_fdtd_openems_fixed.py:296 (+ :433, :225, :621, :301)
 * The reference ships no golden vectors for it (test_openems.py:101-108 only prints SUCCESS).
 * This file therefore restates the PUBLISHED equivalent-circuit FDTD algorithm that engine
 * implements (Rennings et al., EC-FDTD; Liebig et al., "openEMS – a free and open source
 * equivalent-circuit (EC) FDTD simulation platform", IJNM 2013), CPML after Roden & Gedney 2000,
 * first-order Mur, and the surface-equivalence NF2FF integral (Balanis, Antenna Theory §12),
 * and is pinned by physics known-answer tests in tests/ (cavity eigenfrequency, CPML
 * reflection, matched port, energy decay), not by openEMS outputs.
 *
 * It exports the ABI of include/fdtd_hip.h with the float32 operation order spelled out with
 * explicit fmaf() so that the HIP kernels can be compared BIT FOR BIT:
 *     psi  = fmaf(b, psi, c*d)            t = fmaf(ik, d, psi)
 *     curl = t1 - t2                      F = fmaf(cA, F, cB*curl)
 * Build with -ffp-contract=off -mfma (oracle/Makefile).
 *
 * PRECISION SWITCH: -DFDTD_REAL=double builds libfdtd_oracle_f64.so — the same algorithm and the same C
 * ABI (float arrays in and out), with fields, psi, Mur state, operator coefficients, CPML tables, the excitation signal and
 * the NF2FF records held and computed in double.  It bounds what float32 costs over a whole run (12 000-30 000 timesteps,
 * solver_fdtd_openems_fixed.py:171): tests/fp32_error_budget.py, profiles/r04/fp32_error_budget.json.  Through the ABI's
 * float tables the double build sees float32-ROUNDED coefficients (it then measures the rounding of the time stepping
 * alone); the oracle-only entry fdtd_oracle_stage_f64() hands it the same tables in double before the setter that would
 * take the float ones, so that the coefficient rounding is inside the measured difference as well.
 */
#include "../include/fdtd_hip.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>

#ifndef FDTD_REAL
#define FDTD_REAL float
#endif
typedef FDTD_REAL real;
#define REAL_IS_DOUBLE (sizeof(real) == 8)
/* a*b + c with ONE rounding in the working precision (the float build: fmaf, bit for bit what the HIP kernels do) */
static inline real rfma(real a, real b, real c) { return REAL_IS_DOUBLE ? (real)fma((double)a, (double)b, (double)c) : (real)fmaf((float)a, (float)b, (float)c); }
static void real_to_float(float* dst, const real* src, size_t n) { for (size_t q = 0; q < n; ++q) dst[q] = (float)src[q]; }
static void float_to_real(real* dst, const float* src, size_t n) { for (size_t q = 0; q < n; ++q) dst[q] = (real)src[q]; }

#define MAX_PROBES 64
#define MAX_BOXES 64

typedef struct {
  int kind, n;
  int64_t* off; /* local flat offset into the padded field array */
  int8_t* comp;
  real* w;
  double* series;
} probe_t;

typedef struct {
  int kind, comp;
  int32_t lo[3], hi[3];     /* global box */
  int32_t olo[3], ohi[3];   /* owned part (global indices), empty if ohi<olo */
  size_t npts;
  double* acc;              /* running DFT: [nfreq][npts][2] */
  real* rec;                /* recorder: [nsamples][npts] raw samples */
} dftbox_t;

struct fdtd_ctx {
  fdtd_desc d;
  size_t plane, nloc;       /* ny*nx, nk*plane */
  real* Vb[3]; real* Ib[3];   /* base allocations, (nk+2) planes */
  real* V[3];  real* I[3];    /* pointers to local plane 0 */
  real *vv, *vi, *ii, *iv;    /* [3][nloc] */
  int have_op;
  int op_ncls;      /* distinct (vv, m) pairs when the operator came in (or could go out) in class form, else 0 */
  /* CPML */
  int have_cpml;
  int32_t *slot[3]; int nslot[3];
  real* coef;                /* [3][2][3][n_a] */
  size_t coef_off[3];        /* start of each axis block */
  real* psiE[3][2];          /* comp c, which (0: axis a1=(c+1)%3, 1: axis a2=(c+2)%3) */
  real* psiH[3][2];
  /* Mur */
  int mur_on[6]; real mur_c[6];
  real* mur_st[6][2];
  /* excitation */
  real* sig; int nsig;
  int nsrc; int64_t* src_off; int8_t* src_comp; real* src_amp; int32_t* src_delay;
  /* probes, dft */
  int nprobe; probe_t probe[MAX_PROBES];
  int nbox; dftbox_t box[MAX_BOXES];
  int nfreq, every, nsamples; double *tw_v, *tw_i;
  int recorder;             /* 1: boxes keep time-domain samples (fdtd_set_recorder) instead of running DFT sums */
  int64_t step;
  char err[512];
};

static __thread char g_err[512];

static int fail(fdtd_ctx* c, int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(c ? c->err : g_err, 512, fmt, ap);
  va_end(ap);
  return code;
}

/* oracle-only helpers (not part of the product ABI): thread count for the cpu_baseline timing leg */
void fdtd_oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int fdtd_oracle_get_threads(void) { return omp_get_max_threads(); }

int fdtd_oracle_real_bytes(void) { return (int)sizeof(real); }

/* Double-precision tables for the double build (oracle-only; FDTD_E_UNSUPPORTED in the float build).  The C ABI hands every
 * table over as float32; a table staged here (process-wide, one slot per kind) is taken INSTEAD by the next setter that would
 * have taken the float one, and the slot is cleared: EMET / HMET / OVER_VV / OVER_M by fdtd_build_operator (same layout and
 * length as its emet, hmet, over_vv, over_m arguments), CPML by fdtd_set_cpml (its coef argument), SIGNAL by fdtd_set_signal,
 * MUR (6 values) by fdtd_set_mur.  (Source amplitudes and probe weights stay the ABI's floats: per-edge scale factors of the
 * excitation and +-1 weights.) */
enum { STAGE_EMET = 0, STAGE_HMET, STAGE_OVER_VV, STAGE_OVER_M, STAGE_CPML, STAGE_SIGNAL, STAGE_MUR, STAGE_KINDS };
static double* g_stage[STAGE_KINDS];
static size_t g_stage_n[STAGE_KINDS];
int fdtd_oracle_stage_f64(int kind, const double* data, size_t n) {
  if (kind < 0 || kind >= STAGE_KINDS) return FDTD_E_ARG;
  free(g_stage[kind]); g_stage[kind] = NULL; g_stage_n[kind] = 0;
  if (!data || !n) return FDTD_OK;              /* (clears the slot) */
  if (!REAL_IS_DOUBLE) return FDTD_E_UNSUPPORTED;
  g_stage[kind] = (double*)malloc(n * sizeof(double));
  if (!g_stage[kind]) return FDTD_E_NOMEM;
  memcpy(g_stage[kind], data, n * sizeof(double));
  g_stage_n[kind] = n;
  return FDTD_OK;
}
/* table `kind` of n entries in the working precision: the staged doubles when there are exactly n of them, else the floats */
static real* take_table(int kind, const float* src, size_t n) {
  real* d = (real*)malloc((n ? n : 1) * sizeof(real));
  if (!d) return NULL;
  const double* st = (g_stage[kind] && g_stage_n[kind] == n) ? g_stage[kind] : NULL;
  for (size_t q = 0; q < n; ++q) d[q] = st ? (real)st[q] : (real)src[q];
  free(g_stage[kind]); g_stage[kind] = NULL; g_stage_n[kind] = 0;
  return d;
}

/* State and operator in the working precision's full width (oracle-only; the ABI's getters round to float32): what the exact discrete identities
 * of tests/test_oracle_invariants_cpu.py are checked on. */
int fdtd_oracle_get_field_f64(fdtd_ctx* c, int kind, int comp, double* out) {
  if (!c || !out || comp < 0 || comp > 2 || (kind != 0 && kind != 1)) return FDTD_E_ARG;
  const real* src = (kind == FDTD_KIND_V ? c->V : c->I)[comp];
  for (size_t q = 0; q < c->nloc; ++q) out[q] = (double)src[q];
  return FDTD_OK;
}
int fdtd_oracle_set_field_f64(fdtd_ctx* c, int kind, int comp, const double* in) {
  if (!c || !in || comp < 0 || comp > 2 || (kind != 0 && kind != 1)) return FDTD_E_ARG;
  real* dst = (kind == FDTD_KIND_V ? c->V : c->I)[comp];
  for (size_t q = 0; q < c->nloc; ++q) dst[q] = (real)in[q];
  return FDTD_OK;
}
int fdtd_oracle_get_operator_f64(fdtd_ctx* c, double* vv, double* vi, double* ii, double* iv) {
  if (!c || !vv || !vi || !ii || !iv) return FDTD_E_ARG;
  if (!c->have_op) return FDTD_E_STATE;
  for (size_t q = 0; q < 3 * c->nloc; ++q) { vv[q] = (double)c->vv[q]; vi[q] = (double)c->vi[q]; ii[q] = (double)c->ii[q]; iv[q] = (double)c->iv[q]; }
  return FDTD_OK;
}

int fdtd_version(void) { return FDTD_ABI_VERSION; }
int fdtd_device_count(void) { return 0; }
const char* fdtd_backend(void) { return "oracle:cpu"; }
const char* fdtd_last_error(const fdtd_ctx* c) { return c ? c->err : g_err; }

static int n_axis(const fdtd_ctx* c, int a) { return a == 0 ? c->d.nx : a == 1 ? c->d.ny : c->d.nk; }

int fdtd_create(const fdtd_desc* d, fdtd_ctx** out) {
  if (!d || !out) return fail(NULL, FDTD_E_ARG, "null argument");
  if (d->nx < 2 || d->ny < 2 || d->nz < 2 || d->nk < 1 || d->k0 < 0 || d->k0 + d->nk > d->nz)
    return fail(NULL, FDTD_E_ARG, "bad grid/slab %dx%dx%d k0=%d nk=%d", d->nx, d->ny, d->nz, d->k0, d->nk);
  fdtd_ctx* c = (fdtd_ctx*)calloc(1, sizeof(*c));
  if (!c) return fail(NULL, FDTD_E_NOMEM, "calloc");
  c->d = *d;
  c->plane = (size_t)d->nx * d->ny;
  c->nloc = c->plane * d->nk;
  for (int n = 0; n < 3; ++n) {
    c->Vb[n] = (real*)calloc(c->plane * (d->nk + 2), sizeof(real));
    c->Ib[n] = (real*)calloc(c->plane * (d->nk + 2), sizeof(real));
    if (!c->Vb[n] || !c->Ib[n]) { fdtd_destroy(c); return fail(NULL, FDTD_E_NOMEM, "fields"); }
    c->V[n] = c->Vb[n] + c->plane;
    c->I[n] = c->Ib[n] + c->plane;
  }
  *out = c;
  return FDTD_OK;
}

void fdtd_destroy(fdtd_ctx* c) {
  if (!c) return;
  for (int n = 0; n < 3; ++n) {
    free(c->Vb[n]); free(c->Ib[n]); free(c->slot[n]);
    for (int w = 0; w < 2; ++w) { free(c->psiE[n][w]); free(c->psiH[n][w]); }
  }
  free(c->vv); free(c->vi); free(c->ii); free(c->iv); free(c->coef);
  for (int f = 0; f < 6; ++f) { free(c->mur_st[f][0]); free(c->mur_st[f][1]); }
  free(c->sig); free(c->src_off); free(c->src_comp); free(c->src_amp); free(c->src_delay);
  for (int p = 0; p < c->nprobe; ++p) { free(c->probe[p].off); free(c->probe[p].comp); free(c->probe[p].w); free(c->probe[p].series); }
  for (int b = 0; b < c->nbox; ++b) { free(c->box[b].acc); free(c->box[b].rec); }
  free(c->tw_v); free(c->tw_i);
  free(c);
}

static int alloc_op(fdtd_ctx* c) {
  size_t n = 3 * c->nloc * sizeof(real);
  if (!c->vv) { c->vv = malloc(n); c->vi = malloc(n); c->ii = malloc(n); c->iv = malloc(n); }
  return (c->vv && c->vi && c->ii && c->iv) ? 0 : -1;
}

int fdtd_set_operator_raw(fdtd_ctx* c, const float* vv, const float* vi, const float* ii, const float* iv) {
  if (!c || !vv || !vi || !ii || !iv) return fail(c, FDTD_E_ARG, "null operator array");
  if (alloc_op(c)) return fail(c, FDTD_E_NOMEM, "operator");
  for (size_t q = 0; q < 3 * c->nloc; ++q) { c->vv[q] = (real)vv[q]; c->vi[q] = (real)vi[q]; c->ii[q] = (real)ii[q]; c->iv[q] = (real)iv[q]; }
  c->have_op = 1;
  c->op_ncls = 0;
  return FDTD_OK;
}

/* Expansion of the compressed operator with the float32 association fixed by fdtd_hip.h. */
int fdtd_set_operator_classes(fdtd_ctx* c, const uint8_t* ecls, int ncls, const float* cls_vv,
                              const float* cls_m, const float* emet, const float* hmet) {
  if (!c || !ecls || !cls_vv || !cls_m || !emet || !hmet || ncls < 1 || ncls > 256)
    return fail(c, FDTD_E_ARG, "bad class operator");
  if (alloc_op(c)) return fail(c, FDTD_E_NOMEM, "operator");
  const int nx = c->d.nx, ny = c->d.ny, nk = c->d.nk;
  const int tl = nx + ny + nk;
  for (int n = 0; n < 3; ++n) {
    const float *ex = emet + n * tl, *ey = ex + nx, *ez = ey + ny;
    const float *hx = hmet + n * tl, *hy = hx + nx, *hz = hy + ny;
    for (int k = 0; k < nk; ++k)
      for (int j = 0; j < ny; ++j) {
        const real eyz = (real)ey[j] * (real)ez[k];
        const real hyz = (real)hy[j] * (real)hz[k];
        size_t row = n * c->nloc + ((size_t)k * ny + j) * nx;
        for (int i = 0; i < nx; ++i) {
          int cl = ecls[row + i];
          if (cl >= ncls) return fail(c, FDTD_E_ARG, "class %d >= ncls %d", cl, ncls);
          c->vv[row + i] = (real)cls_vv[cl];
          c->vi[row + i] = (real)cls_m[cl] * ((real)ex[i] * eyz);
          c->ii[row + i] = (real)1;
          c->iv[row + i] = (real)hx[i] * hyz;
        }
      }
  }
  c->have_op = 1;
  c->op_ncls = ncls;
  return FDTD_OK;
}

/* Operator set-up from materials + mesh: plain-C restatement of the host formulation
 * (fdtd-solver-antennas_amd/ecoperator.py build_operator, which follows the [EXT] EC-FDTD operator the reference
 * triggers inside FDTD.Run, antenna_sim/solver_fdtd_openems_fixed.py:280; formulas in include/fdtd_hip.h).
 * The oracle keeps the expanded raw arrays whatever prefer_classes says; op_form/op_ncls only report what the
 * compressed form would be (pairs counted with a small open-addressing set). */
int fdtd_build_operator(fdtd_ctx* c, const double* dx, const double* dy, const double* dz, const double* eps_r,
                        const double* kappa, const uint8_t* pec, double eps0, int n_over, const int64_t* over_edge,
                        const int8_t* over_comp, const float* over_vv, const float* over_m, const float* emet,
                        const float* hmet, int prefer_classes) {
  if (!c || !dx || !dy || !dz || !eps_r || !kappa || !pec || !emet || !hmet || n_over < 0 ||
      (n_over > 0 && (!over_edge || !over_comp || !over_vv || !over_m)))
    return fail(c, FDTD_E_ARG, "null argument");
  const int nx = c->d.nx, ny = c->d.ny, nz = c->d.nz, k0 = c->d.k0, nk = c->d.nk;
  if (nx < 3 || ny < 3 || nz < 3) return fail(c, FDTD_E_ARG, "grid too small for an operator");
  if (alloc_op(c)) return fail(c, FDTD_E_NOMEM, "operator");
  const double* d[3] = {dx, dy, dz};
  const int nn[3] = {nx, ny, nz};
  const double dt = c->d.dt;
  const size_t crow = (size_t)(nx - 1), cplane = (size_t)(nx - 1) * (ny - 1), gplane = (size_t)nx * ny;
  const int tl = nx + ny + nk;
  real* mtmp = malloc(3 * c->nloc * sizeof(real));
  /* the 1-D metric tables and the lumped-edge overrides in the working precision (double build: the staged doubles, if any) */
  real* emet_r = take_table(STAGE_EMET, emet, (size_t)3 * tl);
  real* hmet_r = take_table(STAGE_HMET, hmet, (size_t)3 * tl);
  real* over_vv_r = take_table(STAGE_OVER_VV, over_vv, (size_t)n_over);
  real* over_m_r = take_table(STAGE_OVER_M, over_m, (size_t)n_over);
  if (!mtmp || !emet_r || !hmet_r || !over_vv_r || !over_m_r) {
    free(mtmp); free(emet_r); free(hmet_r); free(over_vv_r); free(over_m_r);
    return fail(c, FDTD_E_NOMEM, "operator");
  }
  for (int n = 0; n < 3; ++n) {
    const int a1 = (n + 1) % 3, a2 = (n + 2) % 3;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nk; ++k)
      for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
          const int pos[3] = {i, j, k0 + k};
          const size_t e = n * c->nloc + ((size_t)k * ny + j) * nx + i;
          const int dead = pec[((size_t)n * nz + pos[2]) * gplane + (size_t)j * nx + i] != 0 || pos[n] == nn[n] - 1 ||
                           pos[a1] == 0 || pos[a1] == nn[a1] - 1 || pos[a2] == 0 || pos[a2] == nn[a2] - 1;
          real vv = 0, m = 0;
          if (!dead) {
            double ne = 0.0, nkp = 0.0, den = 0.0;
            for (int o1 = -1; o1 <= 0; ++o1)
              for (int o2 = -1; o2 <= 0; ++o2) {
                int ci[3];
                ci[n] = pos[n]; ci[a1] = pos[a1] + o1; ci[a2] = pos[a2] + o2;
                const double w = d[a1][ci[a1]] * d[a2][ci[a2]];
                const size_t q = (size_t)ci[2] * cplane + (size_t)ci[1] * crow + ci[0];
                ne = ne + eps_r[q] * w;
                nkp = nkp + kappa[q] * w;
                den = den + w;
              }
            const double eps_e = (ne / den) * eps0;
            const double kap_e = nkp / den;
            const double x = ((0.5 * dt) * kap_e) / eps_e;
            vv = (real)((1.0 - x) / (1.0 + x));
            m = (real)(dt / (eps_e * (1.0 + x)));
          }
          c->vv[e] = vv;
          mtmp[e] = m;
        }
  }
  for (int q = 0; q < n_over; ++q) {
    const int64_t e = over_edge[q];
    if (e < 0 || e >= (int64_t)nz * (int64_t)gplane || over_comp[q] < 0 || over_comp[q] > 2) {
      free(mtmp); free(emet_r); free(hmet_r); free(over_vv_r); free(over_m_r);
      return fail(c, FDTD_E_ARG, "override %d out of range", q);
    }
    const int64_t kg = e / (int64_t)gplane;
    if (kg < k0 || kg >= k0 + nk) continue;
    const size_t l = (size_t)over_comp[q] * c->nloc + (size_t)(e - (int64_t)k0 * (int64_t)gplane);
    c->vv[l] = over_vv_r[q];
    mtmp[l] = over_m_r[q];
  }
  /* what the compressed form would be: distinct (vv, m) pairs */
  {
    enum { TAB = 4096 };
    uint64_t* tab = malloc(TAB * sizeof(uint64_t));
    int cnt = 0;
    if (tab) {
      memset(tab, 0xFF, TAB * sizeof(uint64_t));
      for (size_t e = 0; e < 3 * c->nloc && cnt <= 256; ++e) {
        uint32_t a, b;
        const float fa = (float)c->vv[e], fb = (float)mtmp[e];   /* (double build: pairs counted as the float build would see them) */
        memcpy(&a, &fa, 4); memcpy(&b, &fb, 4);
        const uint64_t key = ((uint64_t)a << 32) | b;
        uint64_t h = key * 0x9E3779B97F4A7C15ull;
        size_t s = (size_t)(h >> 40) & (TAB - 1);
        while (tab[s] != ~0ull && tab[s] != key) s = (s + 1) & (TAB - 1);
        if (tab[s] == ~0ull) { tab[s] = key; ++cnt; }
      }
      free(tab);
    }
    c->op_ncls = (prefer_classes && cnt <= 256) ? cnt : 0;
  }
  /* expansion with the float32 association fixed by fdtd_hip.h */
  for (int n = 0; n < 3; ++n) {
    const real *ex = emet_r + n * tl, *ey = ex + nx, *ez = ey + ny;
    const real *hx = hmet_r + n * tl, *hy = hx + nx, *hz = hy + ny;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nk; ++k)
      for (int j = 0; j < ny; ++j) {
        const real eyz = ey[j] * ez[k];
        const real hyz = hy[j] * hz[k];
        const size_t row = n * c->nloc + ((size_t)k * ny + j) * nx;
        for (int i = 0; i < nx; ++i) {
          c->vi[row + i] = mtmp[row + i] * (ex[i] * eyz);
          c->ii[row + i] = (real)1;
          c->iv[row + i] = hx[i] * hyz;
        }
      }
  }
  free(mtmp); free(emet_r); free(hmet_r); free(over_vv_r); free(over_m_r);
  c->have_op = 1;
  return FDTD_OK;
}

int fdtd_operator_form(fdtd_ctx* c, int* form, int* nclasses) {
  if (!c || !form) return FDTD_E_ARG;
  *form = !c->have_op ? 0 : (c->op_ncls > 0 ? 1 : 3);
  if (nclasses) *nclasses = c->op_ncls;
  return FDTD_OK;
}

int fdtd_get_operator(fdtd_ctx* c, float* vv, float* vi, float* ii, float* iv) {
  if (!c || !vv || !vi || !ii || !iv) return fail(c, FDTD_E_ARG, "null argument");
  if (!c->have_op) return fail(c, FDTD_E_STATE, "operator not set");
  for (size_t q = 0; q < 3 * c->nloc; ++q) { vv[q] = (float)c->vv[q]; vi[q] = (float)c->vi[q]; ii[q] = (float)c->ii[q]; iv[q] = (float)c->iv[q]; }
  return FDTD_OK;
}

/* psi extents: axis x -> [nk][ny][nslot_x]; y -> [nk][nslot_y][nx]; z -> [nslot_z][ny][nx] */
static size_t psi_size(const fdtd_ctx* c, int a) {
  if (c->nslot[a] <= 0) return 0;
  if (a == 0) return (size_t)c->d.nk * c->d.ny * c->nslot[0];
  if (a == 1) return (size_t)c->d.nk * c->nslot[1] * c->d.nx;
  return (size_t)c->nslot[2] * c->d.ny * c->d.nx;
}

int fdtd_set_cpml(fdtd_ctx* c, const int32_t* sx, const int32_t* sy, const int32_t* sz,
                  int nsx, int nsy, int nsz, const float* coef) {
  if (!c || !sx || !sy || !sz || !coef) return fail(c, FDTD_E_ARG, "null cpml argument");
  const int32_t* s[3] = {sx, sy, sz};
  int ns[3] = {nsx, nsy, nsz};
  size_t off = 0;
  for (int a = 0; a < 3; ++a) {
    int n = n_axis(c, a);
    free(c->slot[a]);
    c->slot[a] = (int32_t*)malloc(n * sizeof(int32_t));
    memcpy(c->slot[a], s[a], n * sizeof(int32_t));
    for (int q = 0; q < n; ++q)
      if (s[a][q] >= ns[a]) return fail(c, FDTD_E_ARG, "cpml slot out of range on axis %d", a);
    c->nslot[a] = ns[a];
    c->coef_off[a] = off;
    off += (size_t)6 * n;
  }
  free(c->coef);
  c->coef = take_table(STAGE_CPML, coef, off);
  if (!c->coef) return fail(c, FDTD_E_NOMEM, "cpml tables");
  for (int n = 0; n < 3; ++n)
    for (int w = 0; w < 2; ++w) {
      int a = (n + 1 + w) % 3;
      free(c->psiE[n][w]); free(c->psiH[n][w]);
      size_t sz_ = psi_size(c, a);
      c->psiE[n][w] = sz_ ? (real*)calloc(sz_, sizeof(real)) : NULL;
      c->psiH[n][w] = sz_ ? (real*)calloc(sz_, sizeof(real)) : NULL;
    }
  c->have_cpml = 1;
  return FDTD_OK;
}

static const real* cpml_tab(const fdtd_ctx* c, int a, int eh, int which) {
  return c->coef + c->coef_off[a] + (size_t)(eh * 3 + which) * n_axis(c, a);
}

int fdtd_set_mur(fdtd_ctx* c, const int32_t enable[6], const float coeff[6]) {
  if (!c || !enable || !coeff) return fail(c, FDTD_E_ARG, "null mur argument");
  real* co = take_table(STAGE_MUR, coeff, 6);
  if (!co) return fail(c, FDTD_E_NOMEM, "mur");
  for (int f = 0; f < 6; ++f) c->mur_c[f] = co[f];
  free(co);
  for (int f = 0; f < 6; ++f) {
    int a = f / 2;
    int on = enable[f] != 0;
    if (a == 2) { /* z faces live on the rank owning the boundary plane */
      int b = (f & 1) ? c->d.nz - 1 : 0;
      if (b < c->d.k0 || b >= c->d.k0 + c->d.nk) on = 0;
      else if (on && c->d.nk < 2) return fail(c, FDTD_E_UNSUPPORTED, "Mur z face needs nk >= 2");
    }
    c->mur_on[f] = on;
    size_t n = a == 0 ? (size_t)c->d.nk * c->d.ny : a == 1 ? (size_t)c->d.nk * c->d.nx : c->plane;
    for (int t = 0; t < 2; ++t) {
      free(c->mur_st[f][t]);
      c->mur_st[f][t] = on ? (real*)calloc(n, sizeof(real)) : NULL;
    }
  }
  return FDTD_OK;
}

int fdtd_set_signal(fdtd_ctx* c, const float* sig, int n) {
  if (!c || !sig || n < 1) return fail(c, FDTD_E_ARG, "bad signal");
  free(c->sig);
  c->sig = take_table(STAGE_SIGNAL, sig, (size_t)n);
  if (!c->sig) return fail(c, FDTD_E_NOMEM, "signal");
  c->nsig = n;
  return FDTD_OK;
}

/* global flat node index -> local offset; returns -1 if the node's plane is not owned */
static int64_t to_local(const fdtd_ctx* c, int64_t g) {
  int64_t k = g / (int64_t)c->plane;
  if (g < 0 || k >= c->d.nz) return -2;
  if (k < c->d.k0 || k >= c->d.k0 + c->d.nk) return -1;
  return g - (int64_t)c->d.k0 * (int64_t)c->plane;
}

int fdtd_add_source(fdtd_ctx* c, int n, const int64_t* idx, const int8_t* comp, const float* amp, const int32_t* delay) {
  if (!c || n < 0 || (n && (!idx || !comp || !amp || !delay))) return fail(c, FDTD_E_ARG, "bad source");
  int tot = c->nsrc + n;
  c->src_off = realloc(c->src_off, tot * sizeof(int64_t));
  c->src_comp = realloc(c->src_comp, tot * sizeof(int8_t));
  c->src_amp = realloc(c->src_amp, tot * sizeof(real));
  c->src_delay = realloc(c->src_delay, tot * sizeof(int32_t));
  for (int e = 0; e < n; ++e) {
    int64_t l = to_local(c, idx[e]);
    if (l == -2 || comp[e] < 0 || comp[e] > 2) return fail(c, FDTD_E_ARG, "source edge %d out of grid", e);
    if (l < 0) continue;
    c->src_off[c->nsrc] = l; c->src_comp[c->nsrc] = comp[e];
    c->src_amp[c->nsrc] = (real)amp[e]; c->src_delay[c->nsrc] = delay[e];
    c->nsrc++;
  }
  return FDTD_OK;
}

int fdtd_add_probe(fdtd_ctx* c, int kind, int n, const int64_t* idx, const int8_t* comp, const float* w, int* id_out) {
  if (!c || n < 0 || (n && (!idx || !comp || !w)) || (kind != 0 && kind != 1)) return fail(c, FDTD_E_ARG, "bad probe");
  if (c->nprobe >= MAX_PROBES) return fail(c, FDTD_E_NOMEM, "too many probes");
  probe_t* p = &c->probe[c->nprobe];
  memset(p, 0, sizeof(*p));
  p->kind = kind;
  p->off = malloc((n + 1) * sizeof(int64_t)); p->comp = malloc(n + 1); p->w = malloc((n + 1) * sizeof(real));
  p->series = calloc(c->d.max_steps > 0 ? c->d.max_steps : 1, sizeof(double));
  for (int e = 0; e < n; ++e) {
    int64_t l = to_local(c, idx[e]);
    if (l == -2 || comp[e] < 0 || comp[e] > 2) {
      free(p->off); free(p->comp); free(p->w); free(p->series);
      memset(p, 0, sizeof(*p));
      return fail(c, FDTD_E_ARG, "probe edge %d out of grid", e);
    }
    if (l < 0) continue;
    p->off[p->n] = l; p->comp[p->n] = comp[e]; p->w[p->n] = (real)w[e]; p->n++;
  }
  if (id_out) *id_out = c->nprobe;
  c->nprobe++;
  return FDTD_OK;
}

int fdtd_get_probe(fdtd_ctx* c, int id, double* out, int cap, int* n_out) {
  if (!c || id < 0 || id >= c->nprobe) return fail(c, FDTD_E_ARG, "bad probe id");
  int n = (int)(c->step < c->d.max_steps ? c->step : c->d.max_steps);
  if (n_out) *n_out = n;
  if (out) memcpy(out, c->probe[id].series, (size_t)(n < cap ? n : cap) * sizeof(double));
  return FDTD_OK;
}

int fdtd_set_dft(fdtd_ctx* c, int nfreq, int every, int nsamples, const double* tw_v, const double* tw_i) {
  if (!c || nfreq < 1 || every < 1 || nsamples < 1 || !tw_v || !tw_i) return fail(c, FDTD_E_ARG, "bad dft setup");
  if (c->nbox) return fail(c, FDTD_E_STATE, "set_dft must precede add_dft_box");
  size_t n = (size_t)nsamples * nfreq * 2;
  free(c->tw_v); free(c->tw_i);
  c->tw_v = malloc(n * sizeof(double)); c->tw_i = malloc(n * sizeof(double));
  memcpy(c->tw_v, tw_v, n * sizeof(double)); memcpy(c->tw_i, tw_i, n * sizeof(double));
  c->nfreq = nfreq; c->every = every; c->nsamples = nsamples;
  return FDTD_OK;
}

/* Time-domain recording of the boxes: [EXT] what openEMS's NF2FF box dumps (CreateNF2FFBox, solver_fdtd_openems_fixed.py:220)
 * for nf2ff.CalcNF2FF to transform at any frequency afterwards (:296; the S11 resonance of microstrip.py:407-433). */
int fdtd_set_recorder(fdtd_ctx* c, int every, int nsamples) {
  if (!c || every < 1 || nsamples < 1) return fail(c, FDTD_E_ARG, "bad recorder setup");
  if (c->nbox) return fail(c, FDTD_E_STATE, "set_recorder must precede add_dft_box");
  free(c->tw_v); free(c->tw_i); c->tw_v = c->tw_i = NULL;
  c->nfreq = 0; c->recorder = 1; c->every = every; c->nsamples = nsamples;
  return FDTD_OK;
}

int fdtd_add_dft_box(fdtd_ctx* c, int kind, int comp, const int32_t lo[3], const int32_t hi[3], int* id_out) {
  if (!c || !lo || !hi || comp < 0 || comp > 2 || (kind != 0 && kind != 1)) return fail(c, FDTD_E_ARG, "bad dft box");
  if (!c->nfreq && !c->recorder) return fail(c, FDTD_E_STATE, "set_dft or set_recorder first");
  if (c->nbox >= MAX_BOXES) return fail(c, FDTD_E_NOMEM, "too many dft boxes");
  const int dims[3] = {c->d.nx, c->d.ny, c->d.nz};
  for (int a = 0; a < 3; ++a)
    if (lo[a] < 0 || hi[a] >= dims[a] || hi[a] < lo[a]) return fail(c, FDTD_E_ARG, "dft box outside grid");
  dftbox_t* b = &c->box[c->nbox];
  memset(b, 0, sizeof(*b));
  b->kind = kind; b->comp = comp;
  for (int a = 0; a < 3; ++a) { b->lo[a] = lo[a]; b->hi[a] = hi[a]; b->olo[a] = lo[a]; b->ohi[a] = hi[a]; }
  if (b->olo[2] < c->d.k0) b->olo[2] = c->d.k0;
  if (b->ohi[2] > c->d.k0 + c->d.nk - 1) b->ohi[2] = c->d.k0 + c->d.nk - 1;
  b->npts = b->ohi[2] < b->olo[2] ? 0 :
      (size_t)(b->ohi[0] - b->olo[0] + 1) * (b->ohi[1] - b->olo[1] + 1) * (b->ohi[2] - b->olo[2] + 1);
  if (c->recorder) {
    b->rec = b->npts ? calloc(b->npts * (size_t)c->nsamples, sizeof(real)) : NULL;
    if (b->npts && !b->rec) return fail(c, FDTD_E_NOMEM, "recorder box: %zu samples", b->npts * (size_t)c->nsamples);
  } else {
    b->acc = b->npts ? calloc(b->npts * c->nfreq * 2, sizeof(double)) : NULL;
  }
  if (id_out) *id_out = c->nbox;
  c->nbox++;
  return FDTD_OK;
}

int fdtd_get_dft_box(fdtd_ctx* c, int id, double* out, int32_t lo_own[3], int32_t hi_own[3]) {
  if (!c || id < 0 || id >= c->nbox) return fail(c, FDTD_E_ARG, "bad dft box id");
  dftbox_t* b = &c->box[id];
  for (int a = 0; a < 3; ++a) { if (lo_own) lo_own[a] = b->olo[a]; if (hi_own) hi_own[a] = b->ohi[a]; }
  if (out && c->recorder) return fail(c, FDTD_E_STATE, "recorder mode: use fdtd_rec_transform");
  if (out && b->npts) memcpy(out, b->acc, b->npts * c->nfreq * 2 * sizeof(double));
  return FDTD_OK;
}

/* samples of `kind` taken so far: steps 0, every, 2*every, ... of the half-steps already done */
static int64_t rec_count(const fdtd_ctx* c) {
  int64_t n = (c->step + c->every - 1) / c->every;
  return n < c->nsamples ? n : c->nsamples;
}

int fdtd_rec_transform(fdtd_ctx* c, int id, int nfreq, const double* tw, double* out, int32_t lo_own[3], int32_t hi_own[3]) {
  if (!c || id < 0 || id >= c->nbox) return fail(c, FDTD_E_ARG, "bad box id");
  if (!c->recorder) return fail(c, FDTD_E_STATE, "not in recorder mode");
  dftbox_t* b = &c->box[id];
  for (int a = 0; a < 3; ++a) { if (lo_own) lo_own[a] = b->olo[a]; if (hi_own) hi_own[a] = b->ohi[a]; }
  if (!out || !b->npts) return FDTD_OK;
  if (nfreq < 1 || !tw) return fail(c, FDTD_E_ARG, "bad transform");
  const int64_t ns = rec_count(c);
  for (int f = 0; f < nfreq; ++f) {
    double* o = out + (size_t)f * b->npts * 2;
#pragma omp parallel for schedule(static)
    for (int64_t pt = 0; pt < (int64_t)b->npts; ++pt) {
      double ar = 0.0, ai = 0.0;
      for (int64_t s_ = 0; s_ < ns; ++s_) {
        const double v = (double)b->rec[(size_t)s_ * b->npts + pt];
        ar = fma(v, tw[((size_t)s_ * nfreq + f) * 2], ar);
        ai = fma(v, tw[((size_t)s_ * nfreq + f) * 2 + 1], ai);
      }
      o[2 * pt] = ar; o[2 * pt + 1] = ai;
    }
  }
  return FDTD_OK;
}

/* ------------------------------------------------------------------------------------------
 * Half-steps.  [EXT] openEMS engine UpdateVoltages/UpdateCurrents (SURVEY §2.2 N1,N2) with the
 * CPML auxiliary fields fused in (N7: the north-star mandates CPML where openEMS has UPML).
 * ---------------------------------------------------------------------------------------- */

/* CPML transform of one row of differences d[] taken along axis a (in place -> stretched term). */
static void cpml_row(const fdtd_ctx* c, int a, int eh, real* psi_arr, int j, int k, real* d) {
  const int nx = c->d.nx, ny = c->d.ny;
  const real *B = cpml_tab(c, a, eh, 0), *C = cpml_tab(c, a, eh, 1), *K = cpml_tab(c, a, eh, 2);
  if (a == 0) {
    const int ns = c->nslot[0];
    real* psi = psi_arr + ((size_t)k * ny + j) * ns;
    for (int i = 0; i < nx; ++i) {
      int s = c->slot[0][i];
      if (s < 0) continue;
      real p = rfma(B[i], psi[s], C[i] * d[i]);
      psi[s] = p;
      d[i] = rfma(K[i], d[i], p);
    }
  } else {
    int q = a == 1 ? j : k;
    int s = c->slot[a][q];
    if (s < 0) return;
    real* psi = a == 1 ? psi_arr + ((size_t)k * c->nslot[1] + s) * nx
                       : psi_arr + ((size_t)s * ny + j) * nx;
    const real b = B[q], cc = C[q], kk = K[q];
    for (int i = 0; i < nx; ++i) {
      real p = rfma(b, psi[i], cc * d[i]);
      psi[i] = p;
      d[i] = rfma(kk, d[i], p);
    }
  }
}

static void update_E(fdtd_ctx* c) {
  const int nx = c->d.nx, ny = c->d.ny, nk = c->d.nk;
  const ptrdiff_t st[3] = {1, nx, (ptrdiff_t)c->plane};
#pragma omp parallel
  {
    real* d1 = (real*)malloc(2 * nx * sizeof(real));
    real* d2 = d1 + nx;
#pragma omp for collapse(2) schedule(static)
    for (int k = 0; k < nk; ++k)
      for (int j = 0; j < ny; ++j) {
        const size_t row = ((size_t)k * ny + j) * nx;
        for (int n = 0; n < 3; ++n) {
          const int a1 = (n + 1) % 3, a2 = (n + 2) % 3;
          const real* F2 = c->I[a2] + row; /* differenced along a1 */
          const real* F1 = c->I[a1] + row; /* differenced along a2 */
          for (int i = 0; i < nx; ++i) {
            d1[i] = F2[i] - F2[i - st[a1]];
            d2[i] = F1[i] - F1[i - st[a2]];
          }
          if (c->have_cpml) {
            cpml_row(c, a1, 0, c->psiE[n][0], j, k, d1);
            cpml_row(c, a2, 0, c->psiE[n][1], j, k, d2);
          }
          real* V = c->V[n] + row;
          const real* vv = c->vv + n * c->nloc + row;
          const real* vi = c->vi + n * c->nloc + row;
          for (int i = 0; i < nx; ++i) V[i] = rfma(vv[i], V[i], vi[i] * (d1[i] - d2[i]));
        }
      }
    free(d1);
  }
}

static void update_H(fdtd_ctx* c) {
  const int nx = c->d.nx, ny = c->d.ny, nk = c->d.nk;
  const ptrdiff_t st[3] = {1, nx, (ptrdiff_t)c->plane};
#pragma omp parallel
  {
    real* d1 = (real*)malloc(2 * nx * sizeof(real));
    real* d2 = d1 + nx;
#pragma omp for collapse(2) schedule(static)
    for (int k = 0; k < nk; ++k)
      for (int j = 0; j < ny; ++j) {
        const size_t row = ((size_t)k * ny + j) * nx;
        for (int n = 0; n < 3; ++n) {
          const int a1 = (n + 1) % 3, a2 = (n + 2) % 3;
          const real* F2 = c->V[a2] + row;
          const real* F1 = c->V[a1] + row;
          for (int i = 0; i < nx; ++i) {
            d1[i] = F2[i] - F2[i + st[a1]];
            d2[i] = F1[i] - F1[i + st[a2]];
          }
          if (c->have_cpml) {
            cpml_row(c, a1, 1, c->psiH[n][0], j, k, d1);
            cpml_row(c, a2, 1, c->psiH[n][1], j, k, d2);
          }
          real* I = c->I[n] + row;
          const real* ii = c->ii + n * c->nloc + row;
          const real* iv = c->iv + n * c->nloc + row;
          for (int i = 0; i < nx; ++i) I[i] = rfma(ii[i], I[i], iv[i] * (d1[i] - d2[i]));
        }
      }
    free(d1);
  }
}

/* First-order Mur ABC, [EXT] openEMS Engine_Ext_Mur_ABC pre/post/apply.
 * mode 0: pre (before the E update), 1: post (after it), 2: apply. */
static void mur_pass(fdtd_ctx* c, int mode) {
  const int nx = c->d.nx, ny = c->d.ny, nk = c->d.nk;
  const ptrdiff_t st[3] = {1, nx, (ptrdiff_t)c->plane};
  const int dim[3] = {nx, ny, nk};
  for (int f = 0; f < 6; ++f) {
    if (!c->mur_on[f]) continue;
    const int a = f / 2, hi = f & 1;
    /* local index of the boundary line / its inner neighbour along a */
    int b, in;
    if (a == 2) { b = hi ? c->d.nz - 1 - c->d.k0 : 0 - c->d.k0; in = hi ? b - 1 : b + 1; }
    else { b = hi ? dim[a] - 1 : 0; in = hi ? b - 1 : b + 1; }
    const int p = (a + 1) % 3, q = (a + 2) % 3; /* in-face axes */
    const real co = c->mur_c[f];
    for (int t = 0; t < 2; ++t) {
      const int comp = t == 0 ? p : q;
      real* V = c->V[comp];
      real* S = c->mur_st[f][t];
      /* storage index: the two in-face axes in (slow, fast) = (larger axis id, smaller axis id) */
      const int u = p < q ? p : q, v = p < q ? q : p; /* u fast, v slow */
      for (int iv_ = 0; iv_ < dim[v]; ++iv_)
        for (int iu = 0; iu < dim[u]; ++iu) {
          size_t s = (size_t)iv_ * dim[u] + iu;
          ptrdiff_t base = (ptrdiff_t)iu * st[u] + (ptrdiff_t)iv_ * st[v];
          ptrdiff_t ob = base + (ptrdiff_t)b * st[a], oi = base + (ptrdiff_t)in * st[a];
          if (mode == 0) S[s] = rfma(-co, V[ob], V[oi]);
          else if (mode == 1) S[s] = rfma(co, V[oi], S[s]);
          else V[ob] = S[s];
        }
    }
  }
}

static void post_E(fdtd_ctx* c) {
  mur_pass(c, 1);
  mur_pass(c, 2);
  /* soft voltage source, [EXT] Engine_Ext_Excitation::Apply2Voltages */
  for (int e = 0; e < c->nsrc; ++e) {
    int64_t t = c->step - c->src_delay[e];
    if (t < 0 || t >= c->nsig) continue;
    c->V[c->src_comp[e]][c->src_off[e]] += c->src_amp[e] * c->sig[t];
  }
}

static void sample(fdtd_ctx* c, int kind) {
  real** F = kind == FDTD_KIND_V ? c->V : c->I;
  if (c->step < c->d.max_steps)
    for (int p = 0; p < c->nprobe; ++p) {
      probe_t* pr = &c->probe[p];
      if (pr->kind != kind) continue;
      double s = 0.0;
      for (int e = 0; e < pr->n; ++e) s = fma((double)pr->w[e], (double)F[pr->comp[e]][pr->off[e]], s);
      pr->series[c->step] = s;
    }
  if (c->recorder && c->step % c->every == 0 && c->step / c->every < c->nsamples) {
    const int64_t smp = c->step / c->every;
    for (int b = 0; b < c->nbox; ++b) {
      dftbox_t* bx = &c->box[b];
      if (bx->kind != kind || !bx->npts) continue;
      const real* fld = F[bx->comp];
      const int ni = bx->ohi[0] - bx->olo[0] + 1, nj = bx->ohi[1] - bx->olo[1] + 1, nkk = bx->ohi[2] - bx->olo[2] + 1;
      real* dst = bx->rec + (size_t)smp * bx->npts;
      for (int kk = 0; kk < nkk; ++kk)
        for (int jj = 0; jj < nj; ++jj)
          memcpy(dst + ((size_t)kk * nj + jj) * ni,
                 fld + ((size_t)(bx->olo[2] - c->d.k0 + kk) * c->d.ny + bx->olo[1] + jj) * c->d.nx + bx->olo[0], (size_t)ni * sizeof(real));
    }
  }
  if (c->nfreq && c->step % c->every == 0) {
    int64_t smp = c->step / c->every;
    if (smp < c->nsamples) {
      const double* tw = (kind == FDTD_KIND_V ? c->tw_v : c->tw_i) + (size_t)smp * c->nfreq * 2;
      for (int b = 0; b < c->nbox; ++b) {
        dftbox_t* bx = &c->box[b];
        if (bx->kind != kind || !bx->npts) continue;
        const real* fld = F[bx->comp];
        const int ni = bx->ohi[0] - bx->olo[0] + 1, nj = bx->ohi[1] - bx->olo[1] + 1, nkk = bx->ohi[2] - bx->olo[2] + 1;
        for (int f = 0; f < c->nfreq; ++f) {
          const double wr = tw[2 * f], wi = tw[2 * f + 1];
          double* acc = bx->acc + (size_t)f * bx->npts * 2;
#pragma omp parallel for collapse(2) schedule(static)
          for (int kk = 0; kk < nkk; ++kk)
            for (int jj = 0; jj < nj; ++jj) {
              const real* src = fld + ((size_t)(bx->olo[2] - c->d.k0 + kk) * c->d.ny + bx->olo[1] + jj) * c->d.nx + bx->olo[0];
              double* a = acc + ((size_t)kk * nj + jj) * ni * 2;
              for (int ii_ = 0; ii_ < ni; ++ii_) {
                double v = (double)src[ii_];
                a[2 * ii_] = fma(v, wr, a[2 * ii_]);
                a[2 * ii_ + 1] = fma(v, wi, a[2 * ii_ + 1]);
              }
            }
        }
      }
    }
  }
}

int fdtd_half_step(fdtd_ctx* c, int phase) {
  if (!c) return FDTD_E_ARG;
  if (!c->have_op) return fail(c, FDTD_E_STATE, "operator not set");
  if (phase == FDTD_PHASE_E) {
    mur_pass(c, 0);
    update_E(c);
    post_E(c);
    sample(c, FDTD_KIND_V);
  } else if (phase == FDTD_PHASE_H) {
    update_H(c);
    sample(c, FDTD_KIND_I);
    c->step++;
  } else return fail(c, FDTD_E_ARG, "bad phase");
  return FDTD_OK;
}

int fdtd_run(fdtd_ctx* c, int nsteps) {
  if (!c) return FDTD_E_ARG;
  if (c->d.world > 1) return fail(c, FDTD_E_UNSUPPORTED, "oracle: use fdtd_half_step + fdtd_halo_* for world > 1");
  for (int s = 0; s < nsteps; ++s) {
    int r = fdtd_half_step(c, FDTD_PHASE_E);
    if (r) return r;
    r = fdtd_half_step(c, FDTD_PHASE_H);
    if (r) return r;
  }
  return FDTD_OK;
}

int fdtd_run_profiled(fdtd_ctx* c, int nsteps, fdtd_profile* out) {
  if (out) memset(out, 0, sizeof(*out));
  return fdtd_run(c, nsteps);
}

int fdtd_get_step(fdtd_ctx* c, int64_t* step) {
  if (!c || !step) return FDTD_E_ARG;
  *step = c->step;
  return FDTD_OK;
}

/* [EXT] openEMS energy estimate for the end criterion: sums of squares. */
int fdtd_energy(fdtd_ctx* c, double sums[2]) {
  if (!c || !sums) return FDTD_E_ARG;
  double sv = 0.0, si = 0.0;
  for (int n = 0; n < 3; ++n) {
    const real *V = c->V[n], *I = c->I[n];
#pragma omp parallel for reduction(+ : sv, si) schedule(static)
    for (size_t p = 0; p < c->nloc; ++p) { sv += (double)V[p] * V[p]; si += (double)I[p] * I[p]; }
  }
  sums[0] = sv; sums[1] = si;
  return FDTD_OK;
}

/* in-process multi-slab run: the oracle steps the slabs in lock step with direct ghost copies */
int fdtd_link(fdtd_ctx* lower, fdtd_ctx* upper) {
  if (!lower || !upper || upper->d.rank != lower->d.rank + 1 || lower->d.k0 + lower->d.nk != upper->d.k0) return fail(lower, FDTD_E_ARG, "not adjacent slabs");
  return FDTD_OK;
}

int fdtd_run_linked(fdtd_ctx** ctxs, int n, int nsteps) {
  if (!ctxs || n < 1) return FDTD_E_ARG;
  float* buf = (float*)malloc(2 * ctxs[0]->plane * sizeof(float));
  for (int s = 0; s < nsteps; ++s) {
    for (int r = 0; r < n; ++r) { int rc = fdtd_half_step(ctxs[r], FDTD_PHASE_E); if (rc) { free(buf); return rc; } }
    for (int r = 0; r + 1 < n; ++r) { fdtd_halo_get(ctxs[r + 1], FDTD_HALO_E_DOWN, buf); fdtd_halo_put(ctxs[r], FDTD_HALO_E_DOWN, buf); }
    for (int r = 0; r < n; ++r) { int rc = fdtd_half_step(ctxs[r], FDTD_PHASE_H); if (rc) { free(buf); return rc; } }
    for (int r = 0; r + 1 < n; ++r) { fdtd_halo_get(ctxs[r], FDTD_HALO_H_UP, buf); fdtd_halo_put(ctxs[r + 1], FDTD_HALO_H_UP, buf); }
  }
  free(buf);
  return FDTD_OK;
}

int fdtd_p2p_export(fdtd_ctx* c, void* out128) { (void)out128; return fail(c, FDTD_E_UNSUPPORTED, "oracle has no device transport"); }
int fdtd_p2p_attach(fdtd_ctx* c, const void* lo, const void* hi) { (void)lo; (void)hi; return fail(c, FDTD_E_UNSUPPORTED, "oracle has no device transport"); }
int fdtd_p2p_selftest(fdtd_ctx* c, unsigned token) { (void)token; return fail(c, FDTD_E_UNSUPPORTED, "oracle has no device transport"); }
int fdtd_p2p_detach(fdtd_ctx* c) { return c ? FDTD_OK : FDTD_E_ARG; }
int fdtd_p2p_link_info(fdtd_ctx* c, int which, int32_t info[8]) {
  if (!c || !info || (which != 0 && which != 1)) return FDTD_E_ARG;
  for (int q = 0; q < 8; ++q) info[q] = -1;   /* never attached */
  return FDTD_OK;
}
/* the oracle steps one half-step at a time (fdtd_run = fdtd_half_step E, H in a loop): no launches, no tiling */
int fdtd_schedule_info(fdtd_ctx* c, int32_t info[8]) {
  if (!c || !info) return FDTD_E_ARG;
  for (int q = 0; q < 8; ++q) info[q] = 0;
  info[4] = c->d.world > 1 ? 4 : 0;
  return FDTD_OK;
}
int fdtd_comm_unique_id(void* out128) { (void)out128; return fail(NULL, FDTD_E_UNSUPPORTED, "oracle has no RCCL transport"); }
int fdtd_comm_init(fdtd_ctx* c, const void* uid) { (void)uid; return fail(c, FDTD_E_UNSUPPORTED, "oracle has no RCCL transport"); }
int fdtd_comm_nranks(fdtd_ctx* c, int* nranks) { if (!c || !nranks) return FDTD_E_ARG; *nranks = 0; return FDTD_OK; }

int fdtd_halo_get(fdtd_ctx* c, int which, float* buf) {
  if (!c || !buf) return FDTD_E_ARG;
  const size_t n = c->plane;
  if (which == FDTD_HALO_H_UP) {
    real_to_float(buf, c->I[0] + (size_t)(c->d.nk - 1) * c->plane, n);
    real_to_float(buf + c->plane, c->I[1] + (size_t)(c->d.nk - 1) * c->plane, n);
  } else if (which == FDTD_HALO_E_DOWN) {
    real_to_float(buf, c->V[0], n);
    real_to_float(buf + c->plane, c->V[1], n);
  } else return fail(c, FDTD_E_ARG, "bad halo id");
  return FDTD_OK;
}

int fdtd_halo_put(fdtd_ctx* c, int which, const float* buf) {
  if (!c || !buf) return FDTD_E_ARG;
  const size_t n = c->plane;
  if (which == FDTD_HALO_H_UP) { /* ghost plane below */
    float_to_real(c->I[0] - c->plane, buf, n);
    float_to_real(c->I[1] - c->plane, buf + c->plane, n);
  } else if (which == FDTD_HALO_E_DOWN) { /* ghost plane above */
    float_to_real(c->V[0] + c->nloc, buf, n);
    float_to_real(c->V[1] + c->nloc, buf + c->plane, n);
  } else return fail(c, FDTD_E_ARG, "bad halo id");
  return FDTD_OK;
}

int fdtd_get_field(fdtd_ctx* c, int kind, int comp, float* out) {
  if (!c || !out || comp < 0 || comp > 2) return FDTD_E_ARG;
  real_to_float(out, (kind == FDTD_KIND_V ? c->V : c->I)[comp], c->nloc);
  return FDTD_OK;
}

int fdtd_set_field(fdtd_ctx* c, int kind, int comp, const float* in) {
  if (!c || !in || comp < 0 || comp > 2) return FDTD_E_ARG;
  float_to_real((kind == FDTD_KIND_V ? c->V : c->I)[comp], in, c->nloc);
  return FDTD_OK;
}

/* Radiation integral, restating what nf2ff.CalcNF2FF computes from the recorded surfaces
 * (antenna_sim/solver_fdtd_openems_fixed.py:296); Balanis (12-10)..(12-12). */
int fdtd_farfield(int device, int npts, const double* pos, const double* Js, const double* Ms,
                  double kw, int nang, const double* theta, const double* phi, double* Eth, double* Eph) {
  (void)device;
  if (npts < 0 || nang < 0 || !pos || !Js || !Ms || !theta || !phi || !Eth || !Eph) return fail(NULL, FDTD_E_ARG, "bad farfield argument");
  const double eta0 = 376.730313668;  /* sqrt(mu0/eps0) */
  const double fac = kw / (4.0 * M_PI);
#pragma omp parallel for schedule(static)
  for (int a = 0; a < nang; ++a) {
    const double st = sin(theta[a]), ct = cos(theta[a]), sp = sin(phi[a]), cp = cos(phi[a]);
    const double rx = st * cp, ry = st * sp, rz = ct;
    double N[3][2] = {{0}}, L[3][2] = {{0}};
    for (int p = 0; p < npts; ++p) {
      const double ph = kw * (rx * pos[3 * p] + ry * pos[3 * p + 1] + rz * pos[3 * p + 2]);
      const double cr = cos(ph), ci = sin(ph);
      for (int n = 0; n < 3; ++n) {
        const double jr = Js[(3 * p + n) * 2], ji = Js[(3 * p + n) * 2 + 1];
        const double mr = Ms[(3 * p + n) * 2], mi = Ms[(3 * p + n) * 2 + 1];
        N[n][0] += jr * cr - ji * ci; N[n][1] += jr * ci + ji * cr;
        L[n][0] += mr * cr - mi * ci; L[n][1] += mr * ci + mi * cr;
      }
    }
    double Nth[2], Nph[2], Lth[2], Lph[2];
    for (int z = 0; z < 2; ++z) {
      Nth[z] = N[0][z] * ct * cp + N[1][z] * ct * sp - N[2][z] * st;
      Nph[z] = -N[0][z] * sp + N[1][z] * cp;
      Lth[z] = L[0][z] * ct * cp + L[1][z] * ct * sp - L[2][z] * st;
      Lph[z] = -L[0][z] * sp + L[1][z] * cp;
    }
    /* Eth = -j*fac*(Lph + eta*Nth):  -j*(x+jy) = y - jx */
    const double ar = Lph[0] + eta0 * Nth[0], ai = Lph[1] + eta0 * Nth[1];
    Eth[2 * a] = fac * ai; Eth[2 * a + 1] = -fac * ar;
    /* Eph = +j*fac*(Lth - eta*Nph):  j*(x+jy) = -y + jx */
    const double br = Lth[0] - eta0 * Nph[0], bi = Lth[1] - eta0 * Nph[1];
    Eph[2 * a] = -fac * bi; Eph[2 * a + 1] = fac * br;
  }
  return FDTD_OK;
}
