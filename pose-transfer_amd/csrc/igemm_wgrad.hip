// Weight-gradient implicit GEMM for gfx950, fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   dW[r][s][co][ci] += sum_{n,qy,qx} dY[.., co] * X[.., ci]
// where small pixel (qy,qx) pairs with large pixel (qy*stride + r - pad, qx*stride + s - pad): the same relation
// as nn.Conv2d (X large, dY small; reference models/networks.py:154,186,228,341) and
// nn.ConvTranspose2d(k4,s2)+Cropping2D(1) (X small, dY large; networks.py:156-157).
// GEMM view per tap: C[M = Cout][N = Cin tile] = A^T[K = pixels][M] * B[K = pixels][N]; both operands are
// pixel-major NHWC so a tile row is one contiguous float4 run (coalesced) and lands in the K-major LDS tile with a
// single ds_write_b128.  X carries the forward prologue (deferred per-sample norm, dropout mask, activation) and
// the virtual concat of up to 4 sources.  K is split across workgroups (float atomics into the zeroed dW).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace pg {

constexpr int WBK = 32;

struct WgradK {
  pg_src_t src[PG_MAX_SRC];
  int nsrc, Ctot;
  int cstart[PG_MAX_SRC + 1];
  int N, act;
  const float* dY;
  long yN, yC, yH, yW;
  int Cout;
  int x_is_large;
  int Hs, Ws, Hl, Wl;
  int KW, stride, pad;
  float* dW;
  int ksplit, Kpix, ntaps, cout_store;
};

// Unconditional-load helpers: a load inside a branch makes the compiler drain vmcnt at the join (full memory latency
// exposed every K tile), so invalid rows read these dummies / a clamped address and are zeroed at the LDS store.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wc99-designator"
__device__ __attribute__((aligned(16))) const float kOnesW[2048] = {[0 ... 2047] = 1.0f};
#pragma clang diagnostic pop
__device__ __attribute__((aligned(16))) const float kIdentAff[4] = {1.0f, 0.0f, 1.0f, 0.0f};

struct Pix { int n, sy, sx, ly, lx; bool lok; };
struct PixState { int n, sy, sx; };   // small-grid pixel of one tile row, advanced by 32 pixels per K tile

__device__ __forceinline__ Pix decode_pix(const WgradK& p, int k, int r, int s) {
  Pix q;
  const int hw = p.Hs * p.Ws;
  q.n = k / hw;
  const int rem = k - q.n * hw;
  q.sy = rem / p.Ws;
  q.sx = rem - q.sy * p.Ws;
  q.ly = q.sy * p.stride + r - p.pad;
  q.lx = q.sx * p.stride + s - p.pad;
  q.lok = (q.ly >= 0) & (q.ly < p.Hl) & (q.lx >= 0) & (q.lx < p.Wl);   // bitwise: no short-circuit control flow
  return q;
}

__device__ __forceinline__ PixState pix_init(const WgradK& p, int k) {
  PixState st;
  const int hw = p.Hs * p.Ws;
  st.n = k / hw;
  const int rem = k - st.n * hw;
  st.sy = rem / p.Ws;
  st.sx = rem - st.sy * p.Ws;
  return st;
}
__device__ __forceinline__ void pix_advance(const WgradK& p, PixState& st, int advy, int advx) {
  st.sx += advx;
  st.sy += advy;
  if (st.sx >= p.Ws) { st.sx -= p.Ws; st.sy += 1; }
  if (st.sy >= p.Hs) { const int q = st.sy / p.Hs; st.n += q; st.sy -= q * p.Hs; }
}
// branch-free variant: the step is pre-split into (samples, rows, columns) with rows < Hs and columns < Ws
__device__ __forceinline__ void pix_advance3(const WgradK& p, PixState& st, int advn, int advy, int advx) {
  st.sx += advx;
  const bool cx = st.sx >= p.Ws;
  st.sx -= cx ? p.Ws : 0;
  st.sy += advy + (cx ? 1 : 0);
  const bool cy = st.sy >= p.Hs;
  st.sy -= cy ? p.Hs : 0;
  st.n += advn + (cy ? 1 : 0);
}
__device__ __forceinline__ Pix pix_of(const WgradK& p, const PixState& st, int r, int s) {
  Pix q;
  q.n = st.n < p.N ? st.n : -1;
  q.sy = st.sy; q.sx = st.sx;
  q.ly = st.sy * p.stride + r - p.pad;
  q.lx = st.sx * p.stride + s - p.pad;
  q.lok = (q.ly >= 0) & (q.ly < p.Hl) & (q.lx >= 0) & (q.lx < p.Wl);   // bitwise: no short-circuit control flow
  return q;
}

// XL / HM >= 0 (pipelined vector kernels only) fix `x_is_large` / "the X source has a dropout mask" at compile time and
// read the per-sample deferred-norm affine from an LDS table (host: N <= WG_AFF_TAB): the selects between the two
// geometries, the mask offset / multiplies and 8 of the 16 global loads a thread issues per K tile disappear.
// -1 = decided at run time (generic instantiation).
constexpr int WG_AFF_TAB = 64;
template <int BM, int BN, int WGM, int WGN, int WGK, int XS, int YS, int XL = -1, int HM = -1>
__global__ __launch_bounds__(256) void wgrad_igemm_kernel(const WgradK p) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int AS = BM + (YS ? 1 : 4);
  constexpr int BS = BN + (XS ? 1 : 4);
  constexpr int A_CPR = BM / 4, A_PASS = (BM / 32 > 0) ? BM / 32 : 1;   // float4 chunks per pixel row / passes
  constexpr int B_CPR = BN / 4, B_PASS = BN / 32;
  constexpr int A_SC = BM / 8, B_SC = BN / 8;                            // scalar elements per thread
  constexpr int A_SZ = WBK * AS, B_SZ = WBK * BS;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];   // two stages: one barrier per K tile
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * A_SZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tap = blockIdx.z / p.ksplit;
  const int split = blockIdx.z - tap * p.ksplit;
  const int r = tap / p.KW, s = tap - r * p.KW;
  const int co0 = blockIdx.y * BM;
  const int ci0 = blockIdx.x * BN;
  const int nkt = (p.Kpix + WBK - 1) / WBK;
  const int kper = (nkt + p.ksplit - 1) / p.ksplit;
  const int kt0 = split * kper, kt1 = min(nkt, kt0 + kper);
  if (kt0 >= kt1) return;
  const float slope = act_slope(p.act);
  const int advy = WBK / p.Ws, advx = WBK - advy * p.Ws;
  PixState a_st[YS ? 1 : A_PASS], b_st[XS ? 1 : B_PASS];
  if (!YS) {
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) a_st[i] = pix_init(p, kt0 * WBK + tid / A_CPR + i * (256 / A_CPR));
  } else a_st[0] = pix_init(p, kt0 * WBK + (tid & 31));
  if (!XS) {
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) b_st[i] = pix_init(p, kt0 * WBK + tid / B_CPR + i * (256 / B_CPR));
  } else b_st[0] = pix_init(p, kt0 * WBK + (tid & 31));

  // source of this ci tile (vec mode: a tile never straddles sources)
  int jsrc = 0;
#pragma unroll
  for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && ci0 >= p.cstart[q]) jsrc = q;

  float4 ra[YS ? 1 : A_PASS];
  float4 rb[XS ? 1 : B_PASS];
  float4 rbm[XS ? 1 : B_PASS];
  float rba[XS ? 1 : B_PASS], rbb[XS ? 1 : B_PASS];
  unsigned b_ok = 0, a_ok = 0;
  float ras[YS ? A_SC : 1];
  float rbs[XS ? B_SC : 1];

  auto load_tile = [&](int kt) {   // tiles are visited in order: the pixel states advance by one tile per call
    // ---------------- A = dY  [pixel][co]
    if (!YS) {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        const int co = co0 + (tid % A_CPR) * 4;
        const bool cok = co < p.Cout;
        long idx;
        bool ok;
        if (p.x_is_large) {      // dY is the small tensor: dense [pixel][co], no coordinates needed
          const int k = kt * WBK + tid / A_CPR + i * (256 / A_CPR);
          ok = k < p.Kpix && cok;
          idx = ok ? ((long)k * p.Cout + co) : 0;
        } else {
          const Pix q = pix_of(p, a_st[i], r, s);
          pix_advance(p, a_st[i], advy, advx);
          ok = q.n >= 0 && q.lok && cok;
          idx = ok ? ((long)((q.n * p.Hl + q.ly) * p.Wl + q.lx) * p.Cout + co) : 0;
        }
        a_ok = (a_ok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
        ra[i] = *reinterpret_cast<const float4*>(p.dY + idx);
      }
    } else {
      const Pix q = pix_of(p, a_st[0], r, s);
      pix_advance(p, a_st[0], advy, advx);
      // strided dY (scalar mode): address and validity computed branch-free, the load itself is unconditional
      // (loads under divergent branches are serialised by a vmcnt(0) at every join)
      const bool xl = p.x_is_large != 0;
      const bool pok = (q.n >= 0) & (xl | q.lok);
      const long pbase = (long)(q.n >= 0 ? q.n : 0) * p.yN + (long)(xl ? q.sy : q.ly) * p.yH + (long)(xl ? q.sx : q.lx) * p.yW;
#pragma unroll
      for (int e = 0; e < A_SC; ++e) {
        const int co = co0 + (tid >> 5) + 8 * e;
        const bool ok = pok & (co < p.Cout);
        const float v = ldg32(reinterpret_cast<const char*>(p.dY), ok ? (pbase + (long)co * p.yC) * 4 : 0);
        ras[e] = ok ? v : 0.f;
      }
    }
    // ---------------- B = X  [pixel][ci]
    if (!XS) {
      const pg_src_t& sx = p.src[jsrc];
      const int cl = ci0 - p.cstart[jsrc] + (tid % B_CPR) * 4;
      b_ok = 0;
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) {
        const Pix q = pix_of(p, b_st[i], r, s);       // the sample index is needed for aff/mask either way
        pix_advance(p, b_st[i], advy, advx);
        const bool ok = q.n >= 0 && (p.x_is_large ? q.lok : true);
        b_ok |= (ok ? 1u : 0u) << i;
        const int nn = ok ? q.n : 0;
        const int pixidx = !ok ? 0 : (p.x_is_large ? ((q.n * p.Hl + q.ly) * p.Wl + q.lx)
                                                    : (kt * WBK + tid / B_CPR + i * (256 / B_CPR)));
        const float* ap = sx.aff ? (sx.aff + 2 * nn) : kIdentAff;
        const float* mp = sx.mask ? (sx.mask + (long)nn * sx.C + cl) : (kOnesW + (cl & 511));
        rb[i] = *reinterpret_cast<const float4*>(sx.ptr + (long)pixidx * sx.C + cl);
        rba[i] = ap[0]; rbb[i] = ap[1];
        rbm[i] = *reinterpret_cast<const float4*>(mp);
      }
    } else {
      const Pix q = pix_of(p, b_st[0], r, s);
      pix_advance(p, b_st[0], advy, advx);
      const bool xl = p.x_is_large != 0;
      const bool pok = (q.n >= 0) & (q.lok | !xl);
      const long nn = q.n >= 0 ? q.n : 0, yy = xl ? q.ly : q.sy, xx = xl ? q.lx : q.sx;
#pragma unroll
      for (int e = 0; e < B_SC; ++e) {
        const int ci = ci0 + (tid >> 5) + 8 * e;
        // per-lane source pick with constant indices only, then ONE unconditional load (see the dY note above)
        const float* sp = p.src[0].ptr;
        long sN = p.src[0].sN, sC = p.src[0].sC, sH = p.src[0].sH, sW = p.src[0].sW;
        int cs = 0;
#pragma unroll
        for (int t = 1; t < PG_MAX_SRC; ++t)
          if (t < p.nsrc && ci >= p.cstart[t]) {
            sp = p.src[t].ptr; sN = p.src[t].sN; sC = p.src[t].sC; sH = p.src[t].sH; sW = p.src[t].sW; cs = p.cstart[t];
          }
        const bool ok = pok & (ci < p.Ctot);
        const long off = ok ? (nn * sN + (long)(ci - cs) * sC + yy * sH + xx * sW) * 4 : 0;
        const float v = ldg32(reinterpret_cast<const char*>(sp), off);
        rbs[e] = ok ? v : 0.f;
      }
    }
  };

  auto store_tile = [&](int stage) {
    float* As = As0 + stage * A_SZ;
    float* Bs = Bs0 + stage * B_SZ;
    if (!YS) {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        const int pr = tid / A_CPR + i * (256 / A_CPR);
        *reinterpret_cast<float4*>(&As[pr * AS + (tid % A_CPR) * 4]) =
            ((a_ok >> i) & 1u) ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int e = 0; e < A_SC; ++e) As[(tid & 31) * AS + (tid >> 5) + 8 * e] = ras[e];
    }
    if (!XS) {
#pragma unroll
      for (int i = 0; i < B_PASS; ++i) {
        const int pr = tid / B_CPR + i * (256 / B_CPR);
        const bool ok = (b_ok >> i) & 1u;
        float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
        const float mk[4] = {rbm[i].x, rbm[i].y, rbm[i].z, rbm[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = apply_act_s((v[e] * rba[i] + rbb[i]) * mk[e], slope);
          v[e] = ok ? t : 0.f;
        }
        *reinterpret_cast<float4*>(&Bs[pr * BS + (tid % B_CPR) * 4]) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < B_SC; ++e) Bs[(tid & 31) * BS + (tid >> 5) + 8 * e] = rbs[e];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int wk = wave / (WGM * WGN);
  const int wmn = wave - wk * (WGM * WGN);
  const int wm0 = (wmn / WGN) * (TM * 32);
  const int wn0 = (wmn % WGN) * (TN * 32);
  const int l31 = lane & 31, lhi = lane >> 5;
  constexpr int KSPAN = WBK / WGK;

  // operand fetch for k-group g (8 pixels of this wave's K span), register double-buffered against the MFMAs
  constexpr int NG = KSPAN / 8;
  auto fetch = [&](int stage, int g, float (&fa)[TM][4], float (&fb)[TN][4]) {
    const float* As = As0 + stage * A_SZ;
    const float* Bs = Bs0 + stage * B_SZ;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kk = wk * KSPAN + g * 8 + lhi * 4 + e;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i][e] = As[kk * AS + wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j][e] = Bs[kk * BS + wn0 + j * 32 + l31];
    }
  };

  constexpr bool PIPE = !XS && !YS && WGK == 1 && NG == 4;
  if constexpr (PIPE) {
    // -------- software-pipelined K loop (vector loaders): pass i of the loaders (A pass i + B pass i, i = 0..3) travels
    // with k-group g == i of the MFMAs: region g = { activation math + ds_write of pass g (tile t+1) | pixel-state
    // advance + global loads of pass g (tile t+2) | LDS fetch of group g+1 | 16 MFMAs of group g }, one basic block,
    // interleaved by the sched_group_barrier pattern.  Loads and stores are unconditional: past the last tile a pass
    // stops advancing and re-reads its rows (stored into an LDS stage nobody reads again).  Explicit global-address-
    // space loads (no FLAT) off wave-uniform bases.
    static_assert(A_PASS <= 4 && B_PASS <= 4, "one loader pass per k-group");
    const pg_src_t& sx0 = p.src[0];
    const float* xp = sx0.ptr; const float* xa = sx0.aff; const float* xm = sx0.mask; int xC = sx0.C, xcs = 0;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q == jsrc) { xp = p.src[q].ptr; xa = p.src[q].aff; xm = p.src[q].mask; xC = p.src[q].C; xcs = p.cstart[q]; }
    const char* const x_base = uniform_ptr(reinterpret_cast<const char*>(xp));
    const bool has_aff = xa != nullptr, has_mask = xm != nullptr;
    const char* const aff_base = uniform_ptr(reinterpret_cast<const char*>(has_aff ? xa : kIdentAff));
    const char* const m_base = uniform_ptr(reinterpret_cast<const char*>(has_mask ? xm : kOnesW));
    const char* const y_base = uniform_ptr(reinterpret_cast<const char*>(p.dY));
    const int cl = ci0 - xcs + (tid % B_CPR) * 4;
    const int co = co0 + (tid % A_CPR) * 4;
    const bool cok = co < p.Cout;
    // launch-uniform flags held in VGPRs on purpose: selects on them compile to v_cndmask instead of scalar branches,
    // which would cut the regions below into several basic blocks (nothing is interleaved across a branch)
    int xl_v = p.x_is_large, hm_v = has_mask ? 1 : 0;
    if constexpr (XL < 0) asm volatile("" : "+v"(xl_v));
    if constexpr (HM != 0) asm volatile("" : "+v"(hm_v));      // HM = 1: SOME source has a mask; this block's may not
    const bool xl = XL >= 0 ? (XL != 0) : (xl_v != 0), hm = HM == 0 ? false : (hm_v != 0);
    const unsigned affmul = has_aff ? 8u : 0u;
    __shared__ float2 aff_tab[XL >= 0 ? WG_AFF_TAB : 1];
    float2 ab_pre = make_float2(1.f, 0.f);                      // table entry `tid`: loaded here, published below —
    if constexpr (XL >= 0) {                                    // the setup in between hides most of the latency
      const int tn = (has_aff && tid < p.N && tid < WG_AFF_TAB) ? tid : 0;
      const float2 t2 = ldg64(aff_base, (unsigned)tn * affmul);
      if (has_aff && tid < p.N) ab_pre = t2;
    }
    int kc[4] = {kt0, kt0, kt0, kt0};           // tile each pass loads next (clamped at the last tile)
    const int hw3 = p.Hs * p.Ws;
    const int advn3 = WBK / hw3, advy3 = (WBK - advn3 * hw3) / p.Ws, advx3 = WBK - advn3 * hw3 - advy3 * p.Ws;
    float2 rab2[B_PASS];
    // FAST (compile-time geometry, 128x128 tile): the byte offset of the LARGE operand's pixel row is carried along the
    // pixel walk instead of being rebuilt — off is linear in (n, sy, sx), so a step adds a tile-uniform constant plus
    // one correction per wrapped coordinate (no integer multiplies: v_mul_lo_u32 / v_mad_u64_u32 are quarter rate and
    // there were 19 of them per K tile); the dense operand's offset is (uniform tile base) + (per-thread constant);
    // an invalid X row gets the affine (0, 0) instead of a flag that is tested again at the LDS store.
    constexpr bool FAST = XL >= 0 && A_CPR == B_CPR && A_PASS == B_PASS;
    const int CL = (XL == 1) ? xC : p.Cout;                       // channels of the large operand
    const int sshift = p.stride == 2 ? 1 : 0;                    // host: stride 1 or 2 for the FAST variants
    const unsigned c_x = (unsigned)(p.stride * CL * 4), c_y = (unsigned)(p.stride * p.Wl * CL * 4),
                   c_n = (unsigned)(p.Hl * p.Wl * CL * 4);
    const unsigned Dfull = (unsigned)advx3 * c_x + (unsigned)advy3 * c_y + (unsigned)advn3 * c_n;
    const unsigned K1 = c_y - (unsigned)p.Ws * c_x, K2 = c_n - (unsigned)p.Hs * c_y;
    unsigned offL[FAST ? 4 : 1];
    unsigned dense_a = 0, dense_b = 0;                            // per-thread parts of the dense offsets (pass 0 row)
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < A_PASS; ++i) {
        const int ly = b_st[i].sy * p.stride + r - p.pad, lx = b_st[i].sx * p.stride + s - p.pad;
        offL[i] = (unsigned)(((b_st[i].n * p.Hl + ly) * p.Wl + lx) * CL + ((XL == 1) ? cl : co)) * 4u;
      }
      dense_a = (unsigned)((tid / A_CPR) * p.Cout + co) * 4u;
      dense_b = (unsigned)((tid / B_CPR) * xC + cl) * 4u;
    }

    auto load_pass = [&](auto ic, bool first) {
      constexpr int i = decltype(ic)::value;
      const bool adv = !first && kc[i] + 1 < kt1;
      if (adv) ++kc[i];
      const int an = adv ? advn3 : 0, ay = adv ? advy3 : 0, ax = adv ? advx3 : 0;
      if constexpr (FAST) {
        PixState& st = b_st[i];
        st.sx += ax;
        const bool cx = st.sx >= p.Ws;
        st.sx -= cx ? p.Ws : 0;
        st.sy += ay + (cx ? 1 : 0);
        const bool cy = st.sy >= p.Hs;
        st.sy -= cy ? p.Hs : 0;
        st.n += an + (cy ? 1 : 0);
        offL[i] += (adv ? Dfull : 0u) + (cx ? K1 : 0u) + (cy ? K2 : 0u);
        const int ly = (st.sy << sshift) + r - p.pad, lx = (st.sx << sshift) + s - p.pad;
        const bool nok = st.n < p.N;
        const bool inr = nok & ((unsigned)ly < (unsigned)p.Hl) & ((unsigned)lx < (unsigned)p.Wl);
        const unsigned rowk = (unsigned)(kc[i] * WBK + i * (256 / A_CPR));            // wave-uniform
        {   // A = dY
          const bool ok = cok & ((XL == 1) ? nok : inr);
          const unsigned off = (XL == 1) ? rowk * (unsigned)(p.Cout * 4) + dense_a : offL[i];
          a_ok = (a_ok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
          ra[i] = ldg128(y_base, ok ? off : 0u);
        }
        {   // B = X
          const bool ok = (XL == 1) ? inr : nok;
          const unsigned off = (XL == 1) ? offL[i] : rowk * (unsigned)(xC * 4) + dense_b;
          rb[i] = ldg128(x_base, ok ? off : 0u);
          const int nn = ok ? st.n : 0;
          const float2 ab = aff_tab[nn];
          rab2[i] = make_float2(ok ? ab.x : 0.f, ok ? ab.y : 0.f);
          if constexpr (HM != 0) {
            const unsigned mo1 = ((unsigned)nn * (unsigned)xC + (unsigned)cl) * 4u, mo0 = (unsigned)(cl & 511) * 4u;
            rbm[i] = ldg128(m_base, hm ? mo1 : mo0);
          }
        }
        return;
      }
      // 128x128 tile: a thread's A row and B row of pass i are the SAME pixel, so one walk (b_st) serves both operands
      // (two walks were ~30 VALU each per pass: 120 of the ~250 VALU a thread spends per K tile)
      constexpr bool SHARED_PIX = (A_CPR == B_CPR) && (i < A_PASS) && (i < B_PASS);
      Pix qb;
      if constexpr (i < B_PASS) {
        pix_advance3(p, b_st[i], an, ay, ax);
        qb = pix_of(p, b_st[i], r, s);
      }
      if constexpr (i < A_PASS) {
        // x_is_large: dY is the small tensor, dense [pixel][co]; otherwise dY is addressed through the tap geometry
        const int k = kc[i] * WBK + tid / A_CPR + i * (256 / A_CPR);
        Pix q;
        if constexpr (SHARED_PIX) q = qb;
        else { pix_advance3(p, a_st[i], an, ay, ax); q = pix_of(p, a_st[i], r, s); }
        const bool ok = cok & ((xl & (k < p.Kpix)) | (!xl & (q.n >= 0) & q.lok));   // bitwise on purpose
        const unsigned pixa = xl ? (unsigned)k : (unsigned)((q.n * p.Hl + q.ly) * p.Wl + q.lx);
        const unsigned off = ok ? (pixa * (unsigned)p.Cout + (unsigned)co) * 4u : 0u;
        a_ok = (a_ok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
        ra[i] = ldg128(y_base, off);
      }
      if constexpr (i < B_PASS) {
        const Pix q = qb;
        const bool ok = (q.n >= 0) & (q.lok | !xl);
        b_ok = (b_ok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
        const int nn = ok ? q.n : 0;
        const int pixl = (q.n * p.Hl + q.ly) * p.Wl + q.lx, pixs = kc[i] * WBK + tid / B_CPR + i * (256 / B_CPR);
        const int pixidx = ok ? (xl ? pixl : pixs) : 0;
        rb[i] = ldg128(x_base, ((unsigned)pixidx * (unsigned)xC + (unsigned)cl) * 4u);
        if constexpr (XL >= 0) rab2[i] = aff_tab[nn];
        else rab2[i] = ldg64(aff_base, (unsigned)nn * affmul);
        if constexpr (HM != 0) {
          const unsigned mo1 = ((unsigned)nn * (unsigned)xC + (unsigned)cl) * 4u, mo0 = (unsigned)(cl & 511) * 4u;
          rbm[i] = ldg128(m_base, hm ? mo1 : mo0);
        }
      }
    };
    auto store_pass = [&](int stage, auto ic) {
      constexpr int i = decltype(ic)::value;
      float* As = As0 + stage * A_SZ;
      float* Bs = Bs0 + stage * B_SZ;
      if constexpr (i < A_PASS) {
        const int pr = tid / A_CPR + i * (256 / A_CPR);
        *reinterpret_cast<float4*>(&As[pr * AS + (tid % A_CPR) * 4]) =
            ((a_ok >> i) & 1u) ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if constexpr (i < B_PASS) {
        const int pr = tid / B_CPR + i * (256 / B_CPR);
        const bool ok = (b_ok >> i) & 1u;
        float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
        float mk[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (HM != 0) { mk[0] = rbm[i].x; mk[1] = rbm[i].y; mk[2] = rbm[i].z; mk[3] = rbm[i].w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(v[e], rab2[i].x, rab2[i].y);
          if constexpr (HM != 0) t *= mk[e];
          if constexpr (FAST) v[e] = fmaxf(t, slope * t);          // invalid rows carry the affine (0, 0): t == 0
          else v[e] = ok ? fmaxf(t, slope * t) : 0.f;
        }
        *reinterpret_cast<float4*>(&Bs[pr * BS + (tid % B_CPR) * 4]) = make_float4(v[0], v[1], v[2], v[3]);
      }
    };
    constexpr int NM = 4 * TM * TN;
    constexpr int VPER = (44 + NM - 1) / NM;
    auto interleave = [&]() {
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (4 * (TM + TN) + NM - 1) / NM, 0);
        if (k == NM / 2 || k == NM / 2 + 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (k >= NM - 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto mfma16 = [&](const float (&fa)[TM][4], const float (&fb)[TN][4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    };
    if constexpr (XL >= 0) {
      if (tid < WG_AFF_TAB) aff_tab[tid] = ab_pre;
      __syncthreads();
    }
    // the pixel states were initialised AT tile kt0: the first load of every pass must not advance them
    load_pass(I0{}, true); load_pass(I1{}, true); load_pass(I2{}, true); load_pass(I3{}, true);
    store_pass(0, I0{}); store_pass(0, I1{}); store_pass(0, I2{}); store_pass(0, I3{});
    load_pass(I0{}, false); load_pass(I1{}, false); load_pass(I2{}, false); load_pass(I3{}, false);
    __syncthreads();
    int stage = 0;
    float fa[2][TM][4], fb[2][TN][4];
    for (int kt = kt0; kt < kt1; ++kt) {
      fetch(stage, 0, fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
      store_pass(stage ^ 1, I0{}); load_pass(I0{}, false); fetch(stage, 1, fa[1], fb[1]); mfma16(fa[0], fb[0]); interleave();
      __builtin_amdgcn_sched_barrier(0);
      store_pass(stage ^ 1, I1{}); load_pass(I1{}, false); fetch(stage, 2, fa[0], fb[0]); mfma16(fa[1], fb[1]); interleave();
      __builtin_amdgcn_sched_barrier(0);
      store_pass(stage ^ 1, I2{}); load_pass(I2{}, false); fetch(stage, 3, fa[1], fb[1]); mfma16(fa[0], fb[0]); interleave();
      __builtin_amdgcn_sched_barrier(0);
      store_pass(stage ^ 1, I3{}); load_pass(I3{}, false); mfma16(fa[1], fb[1]); interleave();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      stage ^= 1;
    }
  } else {
  load_tile(kt0);
    store_tile(0);
    if (kt0 + 1 < kt1) load_tile(kt0 + 1);
    __syncthreads();
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) {
        store_tile(stage ^ 1);
        if (kt + 2 < kt1) load_tile(kt + 2);
      }
      float fa[2][TM][4], fb[2][TN][4];
      fetch(stage, 0, fa[0], fb[0]);
  #pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) fetch(stage, g + 1, fa[(g + 1) & 1], fb[(g + 1) & 1]);
  #pragma unroll
        for (int e = 0; e < 4; ++e)
  #pragma unroll
          for (int i = 0; i < TM; ++i)
  #pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][e], fb[g & 1][j][e], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      stage ^= 1;
    }
  }

#ifdef PG_ABLATE_WG
  if (acc[0][0][0] != 12345.678f) return;   // diagnostic build (tools/ablate.sh): K loop only
#endif
  const bool atomic = (p.ksplit > 1) || (WGK > 1);
  if constexpr (TM == 2 && TN == 2 && WGK == 1) {
    // un-split launches (the deep layers: few pixels, 16 MB of dW each) are bound by the read-modify-write of dW:
    // row-major float4 accesses through a wave-private LDS tile instead of 4 bytes per lane in the MFMA layout
    if (!atomic && (p.Ctot & 3) == 0 && ((size_t)p.dW & 15) == 0) {
      __syncthreads();                                     // every wave is done with the operand stages
      float* T = smem + (tid >> 6) * (32 * 68);
      const int rsel = lane >> 4, c4 = (lane & 15) * 4;
      const int ci = ci0 + wn0 + c4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) T[((q & 3) + 8 * (q >> 2) + 4 * lhi) * 68 + j * 32 + l31] = acc[i][j][q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 v[8], old[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int co = co0 + wm0 + i * 32 + it * 4 + rsel;
          const bool ok = (co < p.cout_store) & (ci < p.Ctot);
          v[it] = *reinterpret_cast<const float4*>(&T[(it * 4 + rsel) * 68 + c4]);
          old[it] = *reinterpret_cast<const float4*>(p.dW + (ok ? ((long)tap * p.cout_store + co) * p.Ctot + ci : 0));
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int co = co0 + wm0 + i * 32 + it * 4 + rsel;
          if ((co < p.cout_store) & (ci < p.Ctot))
            *reinterpret_cast<float4*>(p.dW + ((long)tap * p.cout_store + co) * p.Ctot + ci) =
                make_float4(old[it].x + v[it].x, old[it].y + v[it].y, old[it].z + v[it].z, old[it].w + v[it].w);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = co0 + wm0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
      if (co >= p.cout_store) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + wn0 + j * 32 + l31;
        if (ci >= p.Ctot) continue;
        float* o = p.dW + ((long)tap * p.cout_store + co) * p.Ctot + ci;
        if (atomic) atomicAdd(o, acc[i][j][q]); else *o += acc[i][j][q];
      }
    }
}

}  // namespace pg

using namespace pg;

extern "C" int pg_conv_wgrad(const pg_wgrad_t* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  PG_REQUIRE(d != nullptr, "pg_conv_wgrad: null descriptor");
  PG_REQUIRE(d->nsrc >= 1 && d->nsrc <= PG_MAX_SRC, "pg_conv_wgrad: nsrc=%d", d->nsrc);
  WgradK k;
  memset(&k, 0, sizeof(k));
  k.nsrc = d->nsrc;
  int ctot = 0;
  for (int j = 0; j < d->nsrc; ++j) { k.src[j] = d->src[j]; k.cstart[j] = ctot; ctot += d->src[j].C; }
  for (int j = d->nsrc; j <= PG_MAX_SRC; ++j) k.cstart[j] = ctot;
  k.Ctot = ctot;
  PG_REQUIRE(ctot == d->Cin, "pg_conv_wgrad: sources have %d channels, Cin=%d", ctot, d->Cin);
  k.N = d->N; k.act = d->act; k.dY = d->dY;
  k.yN = d->yN; k.yC = d->yC; k.yH = d->yH; k.yW = d->yW;
  k.Cout = d->Cout; k.x_is_large = d->x_is_large;
  k.Hs = d->Hs; k.Ws = d->Ws; k.Hl = d->Hl; k.Wl = d->Wl;
  k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.dW = d->dW;
  k.ntaps = d->KH * d->KW;
  k.cout_store = d->cout_store > 0 ? d->cout_store : d->Cout;
  const long kp = (long)d->N * d->Hs * d->Ws;
  PG_REQUIRE(kp > 0 && kp < (1L << 31), "pg_conv_wgrad: pixel count out of range");
  k.Kpix = (int)kp;
  const int xs = d->scalar_x ? 1 : 0, ys = d->scalar_y ? 1 : 0;
  int cfg;  // 0: 128x64, 1: 64x64, 2: 32x64 (intra-block split-K), 3: 128x128
  bool all128 = !xs;
  for (int j = 0; j < d->nsrc; ++j) all128 = all128 && (d->src[j].C % 128 == 0);
  if (ys || d->Cout <= 32) cfg = 2;
  else if (d->Cout % 128 == 0) cfg = all128 ? 3 : 0;
  else cfg = 1;
  if (!xs)
    for (int j = 0; j < d->nsrc; ++j)
      PG_REQUIRE(d->src[j].C % 64 == 0, "pg_conv_wgrad: vec X needs C%%64==0 (src %d has %d)", j, d->src[j].C);
  if (!ys) PG_REQUIRE(d->Cout % 4 == 0, "pg_conv_wgrad: vec dY needs Cout%%4==0");
  // scalar X with <= 32 input channels (the generator's first convolutions, 21 / 18 channels): a 32-wide N tile, so
  // that two thirds of the MFMA work is not spent on padding columns
  const bool narrow = cfg == 1 && xs && ctot <= 32;
  const int BMs[4] = {128, 64, 32, 128};
  const int BNw = cfg == 3 ? 128 : (narrow ? 32 : 64);
  const int mt = cdiv(d->Cout, BMs[cfg]), nt = cdiv(ctot, BNw);
  const int nkt = cdiv(kp, WBK);
  int ks = d->ksplit;
  if (ks <= 0) {
    const long tiles = (long)mt * nt * k.ntaps;
    static const int target = getenv("PG_WG_TARGET") ? atoi(getenv("PG_WG_TARGET")) : 1024;
    ks = (int)((target + tiles - 1) / tiles);
    const int kmax = nkt / 2 > 0 ? nkt / 2 : 1;
    if (ks > kmax) ks = kmax;
    if (ks > 512) ks = 512;
    if (ks < 1) ks = 1;
  }
  // PG_DETERMINISTIC: no K split across workgroups (their float atomics on dW would round in arrival order).  The tiles that split
  // K inside the workgroup (WGK = 2) add exactly two values into a zero-initialised element: commutative, order-free.
  if (deterministic()) ks = 1;
  k.ksplit = ks;
  dim3 grid(nt, mt, k.ntaps * ks);
#define PG_WG(BM, WGM, WGN, WGK, XS, YS) \
  PG_KLAUNCH((wgrad_igemm_kernel<BM, 64, WGM, WGN, WGK, XS, YS>), grid, dim3(256), 0, st, k)
  // compile-time geometry / mask variants of the pipelined kernels (per-sample affines from an LDS table)
  bool any_mask = false;
  for (int j = 0; j < d->nsrc; ++j) any_mask |= d->src[j].mask != nullptr;
  // (workgroups with only a few K tiles keep the generic kernel: the table fill + barrier in front of the pipeline
  //  prologue costs them ~2 us each — measured +10-16 us on launches with 16 tiles per workgroup, -40 us with 32)
  const bool spec = d->N <= WG_AFF_TAB && !xs && !ys && (d->stride == 1 || d->stride == 2) && nkt / ks >= 8 && !env().wg_generic;
#define PG_WG_SPEC(BM, BN)                                                                                             \
  do {                                                                                                                 \
    if (d->x_is_large) {                                                                                               \
      if (any_mask) PG_KLAUNCH((wgrad_igemm_kernel<BM, BN, 2, 2, 1, 0, 0, 1, 1>), grid, dim3(256), 0, st, k);  \
      else PG_KLAUNCH((wgrad_igemm_kernel<BM, BN, 2, 2, 1, 0, 0, 1, 0>), grid, dim3(256), 0, st, k);           \
    } else {                                                                                                           \
      if (any_mask) PG_KLAUNCH((wgrad_igemm_kernel<BM, BN, 2, 2, 1, 0, 0, 0, 1>), grid, dim3(256), 0, st, k);  \
      else PG_KLAUNCH((wgrad_igemm_kernel<BM, BN, 2, 2, 1, 0, 0, 0, 0>), grid, dim3(256), 0, st, k);           \
    }                                                                                                                  \
  } while (0)
  if (cfg == 3) {
    if (spec) PG_WG_SPEC(128, 128);
    else PG_KLAUNCH((wgrad_igemm_kernel<128, 128, 2, 2, 1, 0, 0>), grid, dim3(256), 0, st, k);
  } else if (cfg == 0 && spec) {
    PG_WG_SPEC(128, 64);
  } else if (cfg == 0) {
    PG_REQUIRE(!xs && !ys, "pg_conv_wgrad: scalar operands need Cout<=64");
    PG_WG(128, 2, 2, 1, 0, 0);
  } else if (cfg == 1) {
    PG_REQUIRE(!ys, "pg_conv_wgrad: scalar dY needs Cout<=32");
    if (narrow) PG_KLAUNCH((wgrad_igemm_kernel<64, 32, 2, 1, 2, 1, 0>), grid, dim3(256), 0, st, k);
    else if (xs) PG_WG(64, 2, 2, 1, 1, 0);
    else PG_WG(64, 2, 2, 1, 0, 0);
  } else {
    PG_REQUIRE(!xs, "pg_conv_wgrad: scalar X with Cout<=32 unsupported");
    if (ys) PG_WG(32, 1, 2, 2, 0, 1); else PG_WG(32, 1, 2, 2, 0, 0);
  }
#undef PG_WG
#undef PG_WG_SPEC
  PG_LAUNCH_OK("pg_conv_wgrad");
  last_info() = (narrow ? 4 : cfg) | (xs << 4) | (ys << 8) | (ks << 16) | (1 << 30);
  return 0;
}

// ------------------------------------------------------------------------------------------- bias gradient
namespace pg {
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dY, long rows_outer, long rows_inner, int C,
                                                        long s_outer, long s_inner, long sC, float* db) {
  // grid.x = channel, grid.y = row slices; rows = rows_outer x rows_inner
  __shared__ float red[4];
  const int c = blockIdx.x;
  const long rows = rows_outer * rows_inner;
  float acc = 0.f;
  for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < rows; i += (long)gridDim.y * 256) {
    const long o = i / rows_inner, in = i - o * rows_inner;
    acc += dY[o * s_outer + in * s_inner + (long)c * sC];
  }
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(db + c, t);
}
// planar (NCHW) gradient: rows_inner contiguous floats per (outer row, channel) plane (s_inner == 1, 16-byte aligned, rows_inner %
// 4 == 0).  grid.x = plane (outer * C + c), grid.y = slice of the plane; 16-byte loads, four in flight per lane.  The generic
// kernel above pays a 64-bit division per element (67 us for the 25 MB gradient of the output image at batch 32; this: ~10 us).
__global__ __launch_bounds__(256) void bias_grad_planar_kernel(const float* dY, long rows_inner, int C, long s_outer, long sC, float* db) {
  __shared__ float red[4];
  const long o = blockIdx.x / C;
  const int c = blockIdx.x - (int)o * C;
  const float4* p4 = reinterpret_cast<const float4*>(dY + o * s_outer + (long)c * sC);
  const long n4 = rows_inner >> 2;
  const long per = (n4 + gridDim.y - 1) / gridDim.y;
  const long i0 = (long)blockIdx.y * per, i1 = min(n4, i0 + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long i = i0 + threadIdx.x;
  for (; i + 768 < i1; i += 1024) {
    const float4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
    a0 += (v0.x + v0.y) + (v0.z + v0.w); a1 += (v1.x + v1.y) + (v1.z + v1.w);
    a2 += (v2.x + v2.y) + (v2.z + v2.w); a3 += (v3.x + v3.y) + (v3.z + v3.w);
  }
  for (; i < i1; i += 256) { const float4 v = p4[i]; a0 += (v.x + v.y) + (v.z + v.w); }
  const float t = block_sum_256((a0 + a1) + (a2 + a3), red);
  if (threadIdx.x == 0) atomicAdd(db + c, t);
}
// channel-contiguous (NHWC) gradient with C % 4 == 0 and 256 % (C/4) == 0: every lane streams float4s of its channel
// quad over the pixels (fully coalesced, each byte read once); the generic kernel above reads 4 useful bytes per
// 64-byte sector and every channel block re-reads the same lines
template <bool BF>
__device__ __forceinline__ float4 bg_ld4(const float* base, long idx) {      // idx = element index; BF: bf16 STORAGE (round 3)
  if constexpr (BF) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  } else {
    return *reinterpret_cast<const float4*>(base + idx);
  }
}
template <bool BF>
__global__ __launch_bounds__(256) void bias_grad_nhwc_kernel(const float* dY, long npix, int C, float* db) {
  __shared__ float4 red[256];
  const int cq = C >> 2;                       // float4 chunks per pixel
  const int chunk = threadIdx.x % cq, prow = threadIdx.x / cq, ppb = 256 / cq;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long step = (long)gridDim.x * ppb;
  long pix = (long)blockIdx.x * ppb + prow;
  for (; pix + 7 * step < npix; pix += 8 * step) {      // eight independent loads in flight per lane
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = bg_ld4<BF>(dY, (pix + u * step) * C + chunk * 4);
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc.x += v[u].x + v[u + 1].x; acc.y += v[u].y + v[u + 1].y;
      acc.z += v[u].z + v[u + 1].z; acc.w += v[u].w + v[u + 1].w;
    }
  }
  for (; pix < npix; pix += step) {
    const float4 v = bg_ld4<BF>(dY, pix * C + chunk * 4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < cq) {
    float4 t = red[threadIdx.x];
    for (int r = 1; r < ppb; ++r) {
      const float4 u = red[r * cq + threadIdx.x];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    atomicAdd(db + threadIdx.x * 4 + 0, t.x); atomicAdd(db + threadIdx.x * 4 + 1, t.y);
    atomicAdd(db + threadIdx.x * 4 + 2, t.z); atomicAdd(db + threadIdx.x * 4 + 3, t.w);
  }
}
// bf16 tensors, 8 channels (16 bytes) per lane: twice the bytes in flight per lane of the 4-channel form above
__global__ __launch_bounds__(256) void bias_grad_bf16x8_kernel(const unsigned short* dY, long npix, int C, float* db) {
  __shared__ float red[256][8];
  const int cq = C >> 3;                       // 16-byte chunks per pixel
  const int chunk = threadIdx.x % cq, prow = threadIdx.x / cq, ppb = 256 / cq;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const long step = (long)gridDim.x * ppb;
  long pix = (long)blockIdx.x * ppb + prow;
  auto add = [&](const uint4 u) {
    acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xffff0000u);
    acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xffff0000u);
    acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xffff0000u);
    acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xffff0000u);
  };
  for (; pix + 7 * step < npix; pix += 8 * step) {      // eight independent 16-byte loads in flight per lane
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4*>(dY + (pix + u * step) * C + chunk * 8);
#pragma unroll
    for (int u = 0; u < 8; ++u) add(v[u]);
  }
  for (; pix < npix; pix += step) add(*reinterpret_cast<const uint4*>(dY + pix * C + chunk * 8));
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (threadIdx.x < cq * 8) {
    const int ch = threadIdx.x;                // channel; chunk = ch / 8, element = ch % 8
    float t = 0.f;
    for (int r = 0; r < ppb; ++r) t += red[r * cq + (ch >> 3)][ch & 7];
    atomicAdd(db + ch, t);
  }
}
}  // namespace pg

extern "C" int pg_bias_grad(const float* dY, int64_t rows_outer, int64_t rows_inner, int32_t C, int64_t s_outer,
                            int64_t s_inner, int64_t sC, float* db, void* stream) {
  PG_REQUIRE(C > 0 && rows_outer > 0 && rows_inner > 0, "pg_bias_grad: empty");
  const long rows = rows_outer * rows_inner;
  const bool dense_rows = (rows_inner == 1 && s_outer == C) || (s_inner == C && s_outer == rows_inner * (long)C);
  if (sC == 1 && C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && dense_rows && ((size_t)dY & 15) == 0) {                 // dense NHWC: the streaming kernel
    const int ppb = 256 / (C / 4);
    long blocks = (rows + (long)ppb * 16 - 1) / ((long)ppb * 16);
    // every workgroup ends with C float atomics on the SAME C addresses: measured ~90 ns per workgroup, serialised
    // (2048 workgroups: 106 us for a 67 MB tensor; 96: 21 us = 3.2 TB/s with eight 16-byte loads in flight per lane)
    if (blocks > 96) blocks = 96;
    if (deterministic()) blocks = 1;          // one workgroup: no float atomics in arrival order
    PG_KLAUNCH(pg::bias_grad_nhwc_kernel<false>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dY, rows, C, db);
    PG_LAUNCH_OK("pg_bias_grad");
    return 0;
  }
  if (s_inner == 1 && rows_inner % 4 == 0 && rows_inner >= 4096 && s_outer % 4 == 0 && sC % 4 == 0 && ((size_t)dY & 15) == 0 &&
      rows_outer * C <= 65535 && !deterministic()) {
    int sl = (int)((rows_inner / 4 + 4095) / 4096);       // >= 16 K floats per workgroup
    if (sl > 64) sl = 64;
    PG_KLAUNCH(pg::bias_grad_planar_kernel, dim3((unsigned)(rows_outer * C), sl), dim3(256), 0, (hipStream_t)stream, dY, (long)rows_inner, C,
               (long)s_outer, (long)sC, db);
    PG_LAUNCH_OK("pg_bias_grad");
    return 0;
  }
  int slices = (int)((rows + 256 * 16 - 1) / (256 * 16));
  if (slices > 64) slices = 64;
  if (slices < 1 || deterministic()) slices = 1;          // PG_DETERMINISTIC: one workgroup per channel
  PG_KLAUNCH(pg::bias_grad_kernel, dim3(C, slices), dim3(256), 0, (hipStream_t)stream, dY, (long)rows_outer,
                     (long)rows_inner, C, (long)s_outer, (long)s_inner, (long)sC, db);
  PG_LAUNCH_OK("pg_bias_grad");
  return 0;
}

// db[c] += sum over pixels of a dense NHWC gradient stored as bf16 (bf16 STORAGE on the bf16 data path, round 3)
extern "C" int pg_bias_grad_bf16(const void* dY_bf16, int64_t npix, int32_t C, float* db, void* stream) {
  PG_REQUIRE(dY_bf16 && db && npix > 0 && C > 0 && C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && ((size_t)dY_bf16 & 7) == 0,
             "pg_bias_grad_bf16: dense NHWC bf16 tensor with C %% 4 == 0 and 256 %% (C / 4) == 0 required");
  if (C % 8 == 0 && 256 % (C / 8) == 0 && ((size_t)dY_bf16 & 15) == 0 && !env().bias_grad_x4) {
    const int ppb8 = 256 / (C / 8);
    long blocks8 = (npix + (long)ppb8 * 16 - 1) / ((long)ppb8 * 16);
    static const long bcap8 = getenv("PG_BIAS_GRAD_WGS") ? atol(getenv("PG_BIAS_GRAD_WGS")) : 384;
    const long cap8 = (double)npix * C * 2 >= 64e6 ? bcap8 : 96;
    if (blocks8 > cap8) blocks8 = cap8;
    if (deterministic()) blocks8 = 1;
    PG_KLAUNCH(pg::bias_grad_bf16x8_kernel, dim3((int)blocks8), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const unsigned short*>(dY_bf16), (long)npix, C, db);
    PG_LAUNCH_OK("pg_bias_grad_bf16");
    return 0;
  }
  const int ppb = 256 / (C / 4);
  long blocks = (npix + (long)ppb * 16 - 1) / ((long)ppb * 16);
  // 96 workgroups keep ~1.5 MB in flight (8-byte loads): enough for the batch-4 tensors, a third of HBM rate on the 268 MB ones
  // of batch 32 (0.118 ms each).  Large tensors take up to 384 (the trailing atomics stay under the streaming time).
  static const long bcap = getenv("PG_BIAS_GRAD_WGS") ? atol(getenv("PG_BIAS_GRAD_WGS")) : 384;
  const long cap = (double)npix * C * 2 >= 64e6 ? bcap : 96;
  if (blocks > cap) blocks = cap;
  if (deterministic()) blocks = 1;
  PG_KLAUNCH(pg::bias_grad_nhwc_kernel<true>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float*>(dY_bf16), (long)npix, C, db);
  PG_LAUNCH_OK("pg_bias_grad_bf16");
  return 0;
}
