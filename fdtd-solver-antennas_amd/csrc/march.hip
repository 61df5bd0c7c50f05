// march.hip — one-pass leapfrog (E and H half-step in ONE sweep over the fields): 16 x 16-thread xy tiles marching through
// z, fed by an LDS-DMA ring.  Opt-in (FDTD_FLAG_KERNEL_MARCH) until it beats the two passes on the workload at hand.
//
// Why: the two-pass kernels (kernels.hip) move their bytes at the chip's streaming ceiling; what is left is the bytes —
// E is written by one launch and read back by the next (72 B per cell and step; one pass: 48, 54.6 with the tile overlap).
// Round 1's z-marching kernel had the structure but loaded each plane when it needed it: one exposed memory round trip per
// plane at 2-3 waves/SIMD.  Here every plane arrives in LDS two iterations ahead by global_load_lds_dwordx4 (no register
// destination, so loads in flight cost nothing), the probe tools/streams/march_ring_probe.hip measured 98-108 Gcells/s for
// the bare data movement of this structure on the NS / C3 / C5 grids.
//
// A block owns a 15 x 15 patch of 4-cell groups (60 x 15 cells; thread column 15 / row 15 are feeders that only compute the
// E'(i+1), E'(j+1) their neighbours need) and a chunk [kb, ke) of planes.  Ring slot of plane q: Ix,Iy,Iz,Vx,Vy,Vz of the
// tile (4 KiB each, lane-linear per wave = thread-linear), the low-side halo (row j0-1: Iz, Ix; column i0-1: Iz, Iy) and
// the packed class bytes.  Iteration k:  wait for plane k  ->  barrier  ->  issue plane k+2  ->  E'(k) from slot k (+ Ix,Iy
// of slot k-1), written over V(k) in the slot and to the next-step buffer  ->  H'(k-1) from E'(k-1) (own: registers,
// j+1 / i+1 neighbours: slot k-1), E'(k) (registers) and I(k-1) (slot k-1), to the next-step buffer.  One barrier per plane.
// All loads are asm LDS-DMA with hand-counted s_waitcnt (vector-memory operations retire in order; the counts are the
// operations a wave issues after the plane it waits for), all stores plain C++.
//
// Replaces, like kernels.hip, the stepping inside FDTD.Run(...) of the reference's external engine
// (antenna_sim/solver_fdtd_openems_fixed.py:280).  Single slab, packed class operator, no Mur, (stage 1) no CPML.
#include <vector>

#include "kernel_common.hpp"

namespace {

constexpr int MX = 16, MY = 16, MT = MX * MY, MRING = 4;
constexpr unsigned M_F = 6u * 4096u;              // six field components
constexpr unsigned M_HALO = M_F;                  // 4 x 1 KiB (per wave)
constexpr unsigned M_CLS = M_F + 4096u;           // 4 x 256 B
constexpr unsigned M_SLOT = M_F + 4096u + 1024u;  // 29 696 B per plane

__device__ __forceinline__ void glds4o(const uint8_t* base, const unsigned boff, const unsigned lds_dst) {   // 4 bytes per lane
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(boff), "s"(lds_dst), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier_lds() {   // LDS writes of this wave done, then the workgroup barrier — and NO vmcnt drain
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool PML>
__global__ __launch_bounds__(MT) void k_step_march(const DevParams p, const long long step, const int extra, const int ntx,
                                                   const int nty, const int kc) {
  extern __shared__ float4 s_dyn[];   // [MRING slots of M_SLOT bytes][coefficient table]
  __shared__ SrcStage s_src;
  __shared__ double s_red[FDTD_BLOCK];
  if (extra && blockIdx.x == gridDim.x - 1) {   // probe block: both probe kinds of the step just finished
    probe_block(p, FDTD_KIND_V, step - 1, s_red);
    probe_block(p, FDTD_KIND_I, step - 1, s_red);
    return;
  }
  // ---- block decode: XCD-aware remap; x fastest, then y, then z-chunk (neighbouring tiles march together) ----
  const unsigned nb = gridDim.x - (unsigned)extra, b = blockIdx.x;
  const unsigned q8 = nb >> 3, r8 = nb & 7u, xcd = b & 7u;
  const unsigned v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int bx = (int)(v % (unsigned)ntx);
  const int by = (int)((v / (unsigned)ntx) % (unsigned)nty);
  const int bz = (int)(v / ((unsigned)ntx * (unsigned)nty));
  const int tid = (int)threadIdx.x, tx = tid & (MX - 1), ty = tid >> 4, wave = tid >> 6, lane = tid & 63;
  const int gx_raw = bx * (MX - 1) + tx, j_raw = by * (MY - 1) + ty;
  const int gx = min(gx_raw, p.P4 - 1), j = min(j_raw, p.ny - 1);   // out-of-domain threads shadow the last cell (they never store)
  const int i0 = gx * 4;
  const bool owner = tx < MX - 1 && ty < MY - 1 && gx_raw < p.P4 && j_raw < p.ny;
  const int kb = bz * kc, ke = min(kb + kc, p.nk);
  const int PL = p.plane;

  // soft sources inside this tile column (xy footprint incl. feeders, planes kb..ke): staged once, with ordinary loads,
  // BEFORE any LDS-DMA is in flight (the compiler's waits for them would drain the ring)
  int nsrc_t = 0;
  if (p.nsrc > 0) {
    const int2 srng = p.src_rng4[v];
    nsrc_t = min(srng.y - srng.x, FDTD_BLOCK);
    stage_sources(p, p.src_ids4, srng.x, nsrc_t, step, s_src);
  }
  // loop-invariant metric factors (x tables per 4-cell group, y per row); the z factors are scalar loads per plane
  const float4 ex0 = ld4(p.emet[0][0] + i0), ex1 = ld4(p.emet[1][0] + i0), ex2 = ld4(p.emet[2][0] + i0);
  const float4 hx0 = ld4(p.hmet[0][0] + i0), hx1 = ld4(p.hmet[1][0] + i0), hx2 = ld4(p.hmet[2][0] + i0);
  const float ey0 = p.emet[0][1][j], ey1 = p.emet[1][1][j], ey2 = p.emet[2][1][j];
  const float hy0 = p.hmet[0][1][j], hy1 = p.hmet[1][1][j], hy2 = p.hmet[2][1][j];
  __syncthreads();   // ordinary loads above are complete (their values are in registers / LDS) before the DMA era begins

  const unsigned ring0 = lds_off(s_dyn);
  const float2* s_lut = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(s_dyn) + MRING * M_SLOT);
  {   // coefficient table -> LDS by LDS-DMA (oldest operations of the wave: every later wait covers them)
    const unsigned w_lds = __builtin_amdgcn_readfirstlane(ring0 + MRING * M_SLOT + (unsigned)wave * 1024u);   // (also forces ring0 uniform)
    const float* lsrc = reinterpret_cast<const float*>(p.lut) + 4 * tid;
    if (2 * tid < p.lut_n) glds16(lsrc, w_lds);
    if (2 * (tid + MT) < p.lut_n) glds16(lsrc + 4 * MT, w_lds + 4096u);
  }
  const unsigned uw = __builtin_amdgcn_readfirstlane((unsigned)wave);   // wave index as a scalar (LDS-DMA bases live in M0)
  const unsigned rowoff = (unsigned)(j * p.P + i0);
  // halo lanes of this wave: 0-3 Iz, 4-7 Iy of column i0-1 (group gx0-1) for the wave's four rows; wave 0 also 8-23 Iz,
  // 24-39 Ix of row j0-1 for the tile's sixteen groups
  const int nhalo = wave == 0 ? 40 : 8;
  int hcomp = 2;
  long hoff = 0;   // element offset inside a plane (may be negative: previous row / plane, valid memory, zero coefficients)
  if (lane < 8) {
    hcomp = lane < 4 ? 2 : 1;
    const int jr = min(by * (MY - 1) + wave * 4 + (lane & 3), p.ny - 1);
    hoff = (long)jr * p.P + (long)(bx * (MX - 1)) * 4 - 4;
  } else if (lane < 40) {
    hcomp = lane < 24 ? 2 : 0;
    const int g = min(bx * (MX - 1) + ((lane - 8) & 15), p.P4 - 1);
    hoff = (long)(by * (MY - 1) - 1) * p.P + (long)g * 4;
  }
  const float* hbase = p.I[hcomp] - PL;   // one plane below plane 0
  auto issue = [&](const int q, const int pos) {   // plane q (kb-1 .. ke, clamped by the caller) -> ring position pos
    const unsigned pbase = ring0 + (unsigned)(pos & (MRING - 1)) * M_SLOT;   // wave-uniform
    const unsigned slot = pbase + uw * 1024u;
    const unsigned uo = (unsigned)((q + 1) * PL) + rowoff;
#pragma unroll
    for (int c = 0; c < 3; ++c) glds16o(p.I[c] - PL, uo, slot + (unsigned)c * 4096u);
#pragma unroll
    for (int c = 0; c < 3; ++c) glds16o(p.V[c] - PL, uo, slot + (unsigned)(3 + c) * 4096u);
    if (lane < nhalo) glds16(hbase + max((long)(q + 1) * PL + hoff, 0L), pbase + M_HALO + uw * 1024u);
    glds4o(p.ecls, (unsigned)(min(max(q, 0), p.nk - 1) * PL) + rowoff, pbase + M_CLS + uw * 256u);
  };
  // ---- prologue: planes kb-1, kb, kb+1 (3 x 8 operations) ----
  issue(kb - 1, 0);
  issue(kb, 1);
  issue(min(kb + 1, ke), 2);

  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 exk = zero4, eyk = zero4, ezk = zero4;   // E'(k-1) own, carried
  const float4* ring = s_dyn;
  for (int k = kb; k <= ke; ++k) {
    const int it = k - kb;
    // plane k is the oldest outstanding load: everything issued after it = the plane(s) behind it + the stores in between
    if (it == 0) wait_vm<8>(); else if (it == 1) wait_vm<11>(); else if (it == 2) wait_vm<17>(); else wait_vm<20>();
    barrier_lds();
    issue(min(k + 2, ke), it + 3);
    const float4* sk = ring + (size_t)((it + 1) & (MRING - 1)) * (M_SLOT / 16);   // slot of plane k
    const float4* sm = ring + (size_t)(it & (MRING - 1)) * (M_SLOT / 16);         // slot of plane k-1
    float4* skw = const_cast<float4*>(sk);
    // ---------------- E'(k) ----------------
    float4 vx = zero4, vy = zero4, vz = zero4;
    const bool plane_real = k < p.nk;   // plane nk is the ghost above the slab: its voltages stay zero
    if (plane_real) {
      const float4 ix = sk[0 * MT + tid], iy = sk[1 * MT + tid], iz = sk[2 * MT + tid];
      const float4* halo = sk + M_HALO / 16;
      const float4 iz_jm = ty > 0 ? sk[2 * MT + tid - MX] : halo[8 + tx];
      const float4 ix_jm = ty > 0 ? sk[0 * MT + tid - MX] : halo[24 + tx];
      const float iz_im = tx > 0 ? sk[2 * MT + tid - 1].w : halo[wave * 64 + (ty & 3)].w;
      const float iy_im = tx > 0 ? sk[1 * MT + tid - 1].w : halo[wave * 64 + 4 + (ty & 3)].w;
      const float4 ix_km = sm[0 * MT + tid], iy_km = sm[1 * MT + tid];
      vx = sk[3 * MT + tid]; vy = sk[4 * MT + tid]; vz = sk[5 * MT + tid];
      const uchar4 cc = reinterpret_cast<const uchar4*>(reinterpret_cast<const char*>(sk) + M_CLS)[tid];
      const float4 dx1 = sub4(iz, iz_jm), dx2 = sub4(iy, iy_km);
      const float4 dy1 = sub4(ix, ix_km);
      const float4 dy2 = make_float4(iz.x - iz_im, iz.y - iz.x, iz.z - iz.y, iz.w - iz.z);
      const float4 dz1 = make_float4(iy.x - iy_im, iy.y - iy.x, iy.z - iy.y, iy.w - iy.z);
      const float4 dz2 = sub4(ix, ix_jm);
      const float m0 = ey0 * p.emet[0][2][k], m1 = ey1 * p.emet[1][2][k], m2 = ey2 * p.emet[2][2][k];
      const int c0 = 3 * cc.x, c1 = 3 * cc.y, c2 = 3 * cc.z, c3 = 3 * cc.w;
      float2 l0 = s_lut[c0], l1 = s_lut[c1], l2 = s_lut[c2], l3 = s_lut[c3];
      vx = upd4(make_float4(l0.x, l1.x, l2.x, l3.x), vx,
                make_float4(l0.y * (ex0.x * m0), l1.y * (ex0.y * m0), l2.y * (ex0.z * m0), l3.y * (ex0.w * m0)), dx1, dx2);
      l0 = s_lut[c0 + 1]; l1 = s_lut[c1 + 1]; l2 = s_lut[c2 + 1]; l3 = s_lut[c3 + 1];
      vy = upd4(make_float4(l0.x, l1.x, l2.x, l3.x), vy,
                make_float4(l0.y * (ex1.x * m1), l1.y * (ex1.y * m1), l2.y * (ex1.z * m1), l3.y * (ex1.w * m1)), dy1, dy2);
      l0 = s_lut[c0 + 2]; l1 = s_lut[c1 + 2]; l2 = s_lut[c2 + 2]; l3 = s_lut[c3 + 2];
      vz = upd4(make_float4(l0.x, l1.x, l2.x, l3.x), vz,
                make_float4(l0.y * (ex2.x * m2), l1.y * (ex2.y * m2), l2.y * (ex2.z * m2), l3.y * (ex2.w * m2)), dz1, dz2);
      if (nsrc_t > 0) {
        const int off = k * PL + (int)rowoff;
        apply_staged(s_src, nsrc_t, 0, off, vx);
        apply_staged(s_src, nsrc_t, 1, off, vy);
        apply_staged(s_src, nsrc_t, 2, off, vz);
      }
    }
    skw[3 * MT + tid] = vx; skw[4 * MT + tid] = vy; skw[5 * MT + tid] = vz;   // E'(k) replaces V(k) in the slot
    // the three stores of E'(k): always issued by every wave that owns cells (the counted waits rely on it); plane ke belongs
    // to the chunk above (or is the ghost), so there the stores go to this thread's own cell of plane ke-1 ... no: see below
    {
      const int kst = k < ke ? k : -1;
      if (owner && kst >= 0) {
        const unsigned o = (unsigned)(kst * PL) + rowoff;
        sto4s(p.nt, p.Vn[0], o, vx); sto4s(p.nt, p.Vn[1], o, vy); sto4s(p.nt, p.Vn[2], o, vz);
      }
    }
    // ---------------- H'(k-1) ----------------
    if (it > 0) {
      const int km = k - 1;
      const float4 ix = sm[0 * MT + tid], iy = sm[1 * MT + tid], iz = sm[2 * MT + tid];
      const float4 vz_jp = sm[5 * MT + tid + MX], vx_jp = sm[3 * MT + tid + MX];
      const float vz_ip = sm[5 * MT + tid + 1].x, vy_ip = sm[4 * MT + tid + 1].x;
      const float4 hx1_ = sub4(ezk, vz_jp), hx2_ = sub4(eyk, vy);
      const float4 hy1_ = sub4(exk, vx);
      const float4 hy2_ = make_float4(ezk.x - ezk.y, ezk.y - ezk.z, ezk.z - ezk.w, ezk.w - vz_ip);
      const float4 hz1_ = make_float4(eyk.x - eyk.y, eyk.y - eyk.z, eyk.z - eyk.w, eyk.w - vy_ip);
      const float4 hz2_ = sub4(exk, vx_jp);
      const float m0 = hy0 * p.hmet[0][2][km], m1 = hy1 * p.hmet[1][2][km], m2 = hy2 * p.hmet[2][2][km];
      const float4 nx = make_float4(ix.x + (hx0.x * m0) * (hx1_.x - hx2_.x), ix.y + (hx0.y * m0) * (hx1_.y - hx2_.y),
                                    ix.z + (hx0.z * m0) * (hx1_.z - hx2_.z), ix.w + (hx0.w * m0) * (hx1_.w - hx2_.w));
      const float4 ny_ = make_float4(iy.x + (hx1.x * m1) * (hy1_.x - hy2_.x), iy.y + (hx1.y * m1) * (hy1_.y - hy2_.y),
                                     iy.z + (hx1.z * m1) * (hy1_.z - hy2_.z), iy.w + (hx1.w * m1) * (hy1_.w - hy2_.w));
      const float4 nz = make_float4(iz.x + (hx2.x * m2) * (hz1_.x - hz2_.x), iz.y + (hx2.y * m2) * (hz1_.y - hz2_.y),
                                    iz.z + (hx2.z * m2) * (hz1_.z - hz2_.z), iz.w + (hx2.w * m2) * (hz1_.w - hz2_.w));
      if (owner) {
        const unsigned o = (unsigned)(km * PL) + rowoff;
        sto4s(p.nt, p.In[0], o, nx); sto4s(p.nt, p.In[1], o, ny_); sto4s(p.nt, p.In[2], o, nz);
      }
    }
    exk = vx; eyk = vy; ezk = vz;
  }
  wait_vm<0>();
}

}  // namespace

static int march_kc(const fdtd_ctx* c) { return c->march_kc > 0 ? c->march_kc : 10; }

void march_counts(const fdtd_ctx* c, int& ntx, int& nty, int& ntz, int& kc) {
  kc = march_kc(c);
  ntx = (c->p.P4 + (MX - 1) - 1) / (MX - 1);
  nty = (c->p.ny + (MY - 1) - 1) / (MY - 1);
  ntz = (c->p.nk + kc - 1) / kc;
}

// tile columns (linear id = (bz*nty + by)*ntx + bx) whose threads compute cell group (gx, j) in plane k
void march_tiles_of_cell(const fdtd_ctx* c, int gx, int j, int k, std::vector<int>& out) {
  int ntx, nty, ntz, kc;
  march_counts(c, ntx, nty, ntz, kc);
  auto cover = [](int q, int own, int nt, int* r) {
    int n = 0;
    const int t = q / own;
    if (t < nt) r[n++] = t;
    if (q % own == 0 && t > 0) r[n++] = t - 1;   // q is the feeder of the previous tile
    return n;
  };
  int rx[2], ry[2], rz[2], nz_ = 0;
  const int nx_ = cover(gx, MX - 1, ntx, rx), ny_ = cover(j, MY - 1, nty, ry);
  rz[nz_++] = k / kc;
  if (k % kc == 0 && k > 0) rz[nz_++] = k / kc - 1;   // plane ke of the chunk below (computed redundantly there)
  for (int d = 0; d < nz_; ++d)
    for (int a = 0; a < ny_; ++a)
      for (int b = 0; b < nx_; ++b) out.push_back((rz[d] * nty + ry[a]) * ntx + rx[b]);
}

unsigned march_lds_bytes(const fdtd_ctx* c) { return MRING * M_SLOT + (unsigned)((c->p.lut_n + 1) / 2) * 16u; }

void launch_step_march(fdtd_ctx* c, long long step, bool probe_block, hipStream_t s) {
  int ntx, nty, ntz, kc;
  march_counts(c, ntx, nty, ntz, kc);
  const int extra = probe_block ? 1 : 0;
  const dim3 grid((unsigned)(ntx * nty * ntz + extra)), block(MT);
  const unsigned lds = march_lds_bytes(c);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_step_march<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4608);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_step_march<false>), grid, block, lds, s, c->p, step, extra, ntx, nty, kc);
}
