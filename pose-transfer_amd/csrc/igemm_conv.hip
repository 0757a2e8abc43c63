// Implicit-GEMM convolution for gfx950, fp32 MFMA (v_mfma_f32_32x32x2_f32: exact f32, 157 TF peak).
//
// One kernel family covers every data-path contraction of the Deformable-GAN step:
//   "down" : nn.Conv2d forward (reference models/networks.py:154,186,228,341) and the data-gradient of
//            nn.ConvTranspose2d(k4,s2)+Cropping2D(1) (networks.py:156-157)
//   "up"   : nn.ConvTranspose2d(k4,s2)+crop forward, as stride^2 sub-pixel phases (each a 2x2 conv), and the
//            data-gradient of nn.Conv2d.
// GEMM view: C[M = N*Gy*Gx pixels][Ngemm = channels out] = A[M][K = taps*channels in] * B[K][Ngemm].
//   A is gathered on the fly from up to 4 NHWC sources (the reference's torch.cat is never materialised) with the
//   producer's deferred per-sample norm (a_n*x+b_n), channel-dropout mask and activation fused into the load;
//   B is the packed weight [KH][KW][Cout][Cin], read k-contiguous (forward) or n-contiguous (data-gradient).
// Tiling: 256 threads = 4 waves; block tile BMxBNx32; each wave owns (BM/WGM)x(BN/WGN) as 32x32 MFMA tiles, two
//   workgroups per CU.  LDS tiles are [m][k] / [n][k] rows of 32 k's + 4 pad floats (K-contiguous global float4 -> ONE
//   ds_write_b128; an MFMA operand fetch = ONE conflict-free ds_read_b128 per lane per four k-steps); N-contiguous weights
//   (data-gradient) use [k][n].  The K loop is software-pipelined: two LDS stages, two operand register sets, ONE barrier
//   per K tile, the loader work of tile t+1 / t+2 cut into four chunks that travel with the four 16-MFMA groups of tile t
//   (DESIGN.md section 3).  PREC = 3 (bf16 data path) and DMA = 1 (fp32 data-gradient) have no register loaders at all:
//   operands go global -> LDS with global_load_lds_dwordx4.
// Split-K fills the 256 CUs on the deep, small-M layers: partial tiles through a caller workspace + one fix-up kernel
//   (float atomics without a workspace); the split count comes from a time model (conv_impl below).
// Launches of the bf16 data path that can fill the chip with 256-row tiles are routed to igemm_bf16.hip.
#include "common.h"
#include <cstdlib>
#include <type_traits>

// -DPG_ABLATE=n builds diagnostic variants of the K loop (tools/ablate.sh): 1 = no global loads, 2 = no LDS stores,
// 4 = no MFMA (operands kept live), 8 = no per-tile barrier (WRONG results, timing only), 128 = one workgroup per CU (LDS padding), 256 = all global loads hit one 4 KB block, 512 = relu-only, mask-free loader.  0 = the product kernel.
#ifndef PG_ABLATE
#define PG_ABLATE 0
#endif

#include "igemm_common.h"

namespace pg {

// NOMASK = 1 (pipelined fp32 vector kernels): no source carries a dropout mask — no mask load / multiply in the loader
// (otherwise rows without a mask read a table of ones: 4 of the 12 global loads per K tile and thread).
template <int BM, int BN, int WGM, int WGN, int AMODE, int BMODE, int PREC = 0, int DMA = 0, int NOMASK = 0>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvK p) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  // LDS layouts: A is [m][k] (row = 32 k's + 4 pad floats): the K-contiguous global float4 lands with ONE ds_write_b128
  // and an MFMA operand fetch is ONE ds_read_b128 per lane per 4 k-steps (lanes<32 take k..k+3, lanes>=32 k+4..k+7;
  // row stride 36 floats makes both conflict-free).  B is [n][k] likewise for K-contiguous weights (forward) and
  // [k][n] for N-contiguous weights (data-gradient: float4 along n, ds_read_b32 per k-step).
  constexpr int AS = BK + 4;
  constexpr int BSK = BK + 4;               // [n][k] row stride
  constexpr int BS = BN + 4;                // [k][n] row stride (B_NN)
  constexpr bool B_KN = (BMODE == B_NN);
  constexpr int A_ROWS = BM / 32;           // rows per thread in vec mode
  constexpr int B_ROWS = BN / 32;           // NT: rows per thread
  constexpr int NN_CPR = BN / 4;            // NN: float4 chunks per k-row
  constexpr int NN_PASS = 32 / (256 / NN_CPR);
  constexpr int AS_CNT = 32 / (256 / BM);   // scalar-A elements per thread
  constexpr int BS_CNT = BN / 8;            // scalar-B elements per thread

  // two LDS stages: tile t+1 is written while tile t is being multiplied -> ONE barrier per K tile, and a wave's
  // loader work is followed directly by its own MFMAs (the co-resident workgroup fills the MFMA pipe meanwhile)
  constexpr bool LP = (PREC == 1 || PREC == 2);  // bf16 operand tiles converted in the loaders (fp32 storage)
  constexpr int BKE = (PREC == 3) ? 64 : BK;     // K elements per tile: PREC 3 moves 64 bf16 (128 bytes) per row
  constexpr int NPART = (PREC == 2) ? 2 : 1;     // hi (+ lo) parts
  constexpr int ASB = 40;                        // bf16 row stride: 32 k's + 8 pad = 80 bytes (conflict-free b128 reads)
  static_assert(!LP || (AMODE == A_VEC && BMODE != B_SCALAR), "low-precision modes exist for the vector loaders only");
  constexpr int A_SZ = LP ? BM * ASB * NPART / 2 : BM * AS;     // in 4-byte units
  constexpr int B_SZ = LP ? BN * ASB * NPART / 2 : BN * BSK;
  constexpr int AFF_SZ = (AMODE == A_VEC) ? BM * PG_MAX_SRC * 2 : 0;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ) + BM * (sizeof(RowInfo) / 4) + AFF_SZ];
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * A_SZ;
  RowInfo* rows = reinterpret_cast<RowInfo*>(smem + 2 * (A_SZ + B_SZ));
  float* const affs = smem + 2 * (A_SZ + B_SZ) + BM * (sizeof(RowInfo) / 4);   // [row][src](a,b): deferred-norm affine
#if (PG_ABLATE & 128)
  __shared__ float pad_l[6000];              // diagnostic: push LDS past 80 KB -> one workgroup per CU
  if (p.ksplit == 12345) pad_l[threadIdx.x] = 1.f;
#endif
  __shared__ int taps_l[MAXTAP];             // this phase's (dy, dx, weight tap) packed: LDS, not kernarg vector loads

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int zphase = blockIdx.z / p.ksplit;
  const int split = blockIdx.z - zphase * p.ksplit;
  const int phase = p.gtaps ? 0 : zphase;                         // sub-pixel phase (tap-table row)
  const long a_off_g = p.gtaps ? p.a_off[zphase] : 0, w_off_g = p.gtaps ? p.w_off[zphase] : 0;
  float* const out_g = p.out + (p.gtaps ? p.o_off[zphase] : 0);
  // Workgroups are dealt round-robin to the 8 XCDs (one L2 each) in dispatch order (x fastest).  Remapped, the N
  // tiles that share an M tile's activation rows are 8 dispatches apart on the SAME XCD instead of gridDim.x apart on
  // any XCD: the bf16 data path is bound by operand traffic, the fp32 kernels are not (measured: no effect there).
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.xcd_swizzle) {
    const int L = bx + by * (int)gridDim.x, nt = (int)gridDim.y;
    const int g = L / (8 * nt), r = L - g * 8 * nt;
    by = r >> 3;
    bx = g * 8 + (r & 7);
  }
  const int m0 = bx * BM;
  const int nb0 = by * BN;
  const int ntap = p.ntap[phase];
  const bool b_edge = nb0 + BN > p.n_cnt;     // uniform: only edge tiles pay for zeroing rows beyond N

  if (tid < MAXTAP)
    taps_l[tid] = (p.dy[phase][tid] & 0xff) | ((p.dx[phase][tid] & 0xff) << 8) | ((int)p.wtap[phase][tid] << 16);
  if (tid < BM) {
    RowInfo ri;
    const int m = m0 + tid;
    ri.n = -1; ri.iy = 0; ri.ix = 0; ri.oy = 0; ri.ox = 0;
    if (m < p.M) {
      const int gg = p.Gy * p.Gx;
      const int n = m / gg;
      const int rem = m - n * gg;
      const int qy = rem / p.Gx;
      const int qx = rem - qy * p.Gx;
      const int oy = qy * p.so + p.phy[phase];
      const int ox = qx * p.so + p.phx[phase];
      if (oy < p.Ho && ox < p.Wo) {
        ri.n = n;
        ri.iy = (short)(qy * p.si);
        ri.ix = (short)(qx * p.si);
        ri.oy = (short)oy;
        ri.ox = (short)ox;
      }
    }
    rows[tid] = ri;
    if (AMODE == A_VEC) {
#pragma unroll
      for (int j = 0; j < PG_MAX_SRC; ++j) {
        float a = 1.f, b = 0.f;
        if (j < p.nsrc && p.src[j].aff && ri.n >= 0) { a = p.src[j].aff[2 * ri.n]; b = p.src[j].aff[2 * ri.n + 1]; }
        affs[(tid * PG_MAX_SRC + j) * 2] = a;
        affs[(tid * PG_MAX_SRC + j) * 2 + 1] = b;
      }
    }
  }
  __syncthreads();

  // K range of this block
  const int ktot = (AMODE == A_VEC) ? ntap * (p.Ctot / BKE) : (ntap * p.Ctot + BK - 1) / BK;
  const int kper = (ktot + p.ksplit - 1) / p.ksplit;
  const int kt0 = split * kper;
  const int kt1 = min(ktot, kt0 + kper);
  const int cpt = p.Ctot / BKE;  // chunks per tap (vec mode)

  // ---- per-thread loader state
  int a_n[A_ROWS], a_iy[A_ROWS], a_ix[A_ROWS];
  if (AMODE == A_VEC) {
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const RowInfo& r = rows[(tid >> 3) + 32 * i];
      a_n[i] = r.n; a_iy[i] = r.iy; a_ix[i] = r.ix;
    }
  }
  const int s_row = tid % BM;       // scalar-A mapping
  const int s_ksub = tid / BM;
  int s_n = 0, s_iy = 0, s_ix = 0;
  if (AMODE == A_SCALAR) { s_n = rows[s_row].n; s_iy = rows[s_row].iy; s_ix = rows[s_row].ix; }

  // prefetch registers
  float4 ra[A_ROWS];
  float4 rmask[A_ROWS];
  float raa[A_ROWS], rab[A_ROWS];
  float rs[AS_CNT];
  float4 rb[(BMODE == B_NT) ? B_ROWS : (BMODE == B_NN ? NN_PASS : 1)];
  float rbs[BS_CNT];
  const char* a_base = nullptr;     // wave-uniform bases; per-row 32-bit BYTE offsets
  const char* m_base = nullptr;
  unsigned aoff[A_ROWS], moff[A_ROWS];
  unsigned boff[(BMODE == B_NT) ? B_ROWS : (BMODE == B_NN ? NN_PASS : 1)];
  bool a_has_mask = false;
  int a_tap = -1, a_src = -1, b_tap = -1;

  auto load_tile = [&](int kt) {
    // ------------------------------------------------ A operand
    if (AMODE == A_VEC) {
      const int tap = kt / cpt;
      const int cc = (kt - tap * cpt) * BK;
      int j = 0;
#pragma unroll
      for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && cc >= p.cstart[q]) j = q;
      if (tap != a_tap || j != a_src) {
        // (tap, source) changed: rebuild the per-row offsets / bounds (address arithmetic + LDS reads only);
        // otherwise the offsets just advance by one K tile
        const pg_src_t& s = p.src[j];
        const int cl = cc - p.cstart[j] + (tid & 7) * 4;
        const int tp = taps_l[tap];
        const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
        a_base = reinterpret_cast<const char*>(s.ptr);
        a_has_mask = s.mask != nullptr;
        m_base = reinterpret_cast<const char*>(a_has_mask ? s.mask : kOnes);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
          const int iy = a_iy[i] + dyv, ix = a_ix[i] + dxv;
          const bool ok = a_n[i] >= 0 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
          const int row = (tid >> 3) + 32 * i;
          // zero padding = affine (0,0) on a valid dummy element (first pixel of the source): no per-element select
          raa[i] = ok ? affs[(row * PG_MAX_SRC + j) * 2] : 0.f;
          rab[i] = ok ? affs[(row * PG_MAX_SRC + j) * 2 + 1] : 0.f;
          const int nn = ok ? a_n[i] : 0;
          const unsigned pix = ok ? (unsigned)((nn * p.Hi + iy) * p.Wi + ix) : 0u;
          aoff[i] = (pix * (unsigned)s.C + (unsigned)cl) * 4u;
          moff[i] = a_has_mask ? ((unsigned)(nn * s.C + cl) * 4u) : ((unsigned)(cl & 511) * 4u);
        }
        a_tap = tap; a_src = j;
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): no scalar load stays in flight past this rare path
      } else {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) { aoff[i] += BK * 4; moff[i] += BK * 4; }
      }
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {     // unconditional straight-line loads
        ra[i] = *reinterpret_cast<const float4*>(a_base + aoff[i]);
        rmask[i] = *reinterpret_cast<const float4*>(m_base + moff[i]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < AS_CNT; ++e) {
        const int kk = s_ksub + e * (256 / BM);
        const int k = kt * BK + kk;
        float v = 0.f;
        if (k < ntap * p.Ctot && s_n >= 0) {
          const int tap = k / p.Ctot;
          const int c = k - tap * p.Ctot;
          int j = 0;
#pragma unroll
          for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && c >= p.cstart[q]) j = q;
          const pg_src_t& s = p.src[j];
          const int iy = s_iy + p.dy[phase][tap], ix = s_ix + p.dx[phase][tap];
          if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
            v = s.ptr[(long)s_n * s.sN + (long)(c - p.cstart[j]) * s.sC + (long)iy * s.sH + (long)ix * s.sW];
        }
        rs[e] = v;
      }
    }
    // ------------------------------------------------ B operand
    if (BMODE == B_NT) {
      const int tap = kt / cpt;
      if (tap != b_tap) {
        const int cc = (kt - tap * cpt) * BK + (tid & 7) * 4;
        const int base = (taps_l[tap] >> 16) * p.wCout;
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
          const int n = nb0 + (tid >> 3) + 32 * i;
          boff[i] = (unsigned)((base + p.n_off + (n < p.n_cnt ? n : 0)) * p.wCin + cc) * 4u;
        }
        b_tap = tap;
        __builtin_amdgcn_s_waitcnt(0xC07F);
      } else {
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) boff[i] += BK * 4;
      }
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i)      // row clamped; zeroed at store
        rb[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.W) + boff[i]);
    } else if (BMODE == B_NN) {
      const int tap = kt / cpt;
      if (tap != b_tap) {
        const int cc = (kt - tap * cpt) * BK;
        const int base = (taps_l[tap] >> 16) * p.wCout;
#pragma unroll
        for (int i = 0; i < NN_PASS; ++i) {
          // fp32: lanes along n (coalesced float4 rows, [k][n] LDS image).  Low precision: lanes along k so that the
          // transposing scatter into the [n][k] bf16 image is conflict-free
          const int kr = LP ? (tid & 31) : (tid / NN_CPR + i * (256 / NN_CPR));
          const int n = nb0 + (LP ? ((tid >> 5) + 8 * i) * 4 : (tid % NN_CPR) * 4);
          boff[i] = (unsigned)((base + cc + kr) * p.wCin + p.n_off + (n < p.n_cnt ? n : 0)) * 4u;
        }
        b_tap = tap;
        __builtin_amdgcn_s_waitcnt(0xC07F);
      } else {
#pragma unroll
        for (int i = 0; i < NN_PASS; ++i) boff[i] += (unsigned)(BK * p.wCin) * 4u;
      }
#pragma unroll
      for (int i = 0; i < NN_PASS; ++i)     // column clamped; zeroed at store
        rb[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.W) + boff[i]);
    } else {
#pragma unroll
      for (int e = 0; e < BS_CNT; ++e) {
        int kk, nl;
        if (p.w_transposed) { nl = tid % BN; kk = tid / BN + e * (256 / BN); }
        else { kk = tid & 31; nl = (tid >> 5) + 8 * e; }
        int tap, c; bool kval;
        if (AMODE == A_VEC) { tap = kt / cpt; c = (kt - tap * cpt) * BK + kk; kval = true; }
        else { const int k = kt * BK + kk; tap = k / p.Ctot; c = k - tap * p.Ctot; kval = k < ntap * p.Ctot; }
        float v = 0.f;
        const int n = nb0 + nl;
        if (kval && n < p.n_cnt) {
          const long base = (long)p.wtap[phase][tap] * p.wCout;
          v = p.w_transposed ? p.W[(base + c) * p.wCin + p.n_off + n] : p.W[(base + p.n_off + n) * p.wCin + c];
        }
        rbs[e] = v;
      }
    }
  };

  // low-precision store: 4 consecutive k's of one row -> bf16 hi (and lo) parts, one ds_write_b64 each
  auto store4_lp = [&](unsigned short* tile, int rows_in_tile, int row, int kq, const float (&t)[4]) {
    const unsigned h01 = pack_bf16(t[0], t[1]), h23 = pack_bf16(t[2], t[3]);
    *reinterpret_cast<uint2*>(tile + row * ASB + kq) = make_uint2(h01, h23);
    if constexpr (PREC == 2) {
      const unsigned l01 = pack_bf16(t[0] - bf16_lo_f32(h01), t[1] - bf16_hi_f32(h01));
      const unsigned l23 = pack_bf16(t[2] - bf16_lo_f32(h23), t[3] - bf16_hi_f32(h23));
      *reinterpret_cast<uint2*>(tile + rows_in_tile * ASB + row * ASB + kq) = make_uint2(l01, l23);
    }
  };
  auto store_tile = [&](int stage) {
    float* As = As0 + stage * A_SZ;
    float* Bs = Bs0 + stage * B_SZ;
    if constexpr (LP) {
      unsigned short* Ah = reinterpret_cast<unsigned short*>(As);
      unsigned short* Bh = reinterpret_cast<unsigned short*>(Bs);
      auto emit = [&](auto has_mask, auto act_c) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
          float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
          const float mk[4] = {rmask[i].x, rmask[i].y, rmask[i].z, rmask[i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = fmaf(v[e], raa[i], rab[i]);
            if constexpr (decltype(has_mask)::value) t *= mk[e];
            if constexpr (decltype(act_c)::value == PG_ACT_RELU) t = fmaxf(t, 0.f);
            if constexpr (decltype(act_c)::value == PG_ACT_LEAKY) t = fmaxf(t, 0.2f * t);
            v[e] = t;
          }
          store4_lp(Ah, BM, (tid >> 3) + 32 * i, (tid & 7) * 4, v);
        }
      };
      using T1 = std::integral_constant<bool, true>;
      using T0 = std::integral_constant<bool, false>;
      using AN = std::integral_constant<int, PG_ACT_NONE>;
      using AR = std::integral_constant<int, PG_ACT_RELU>;
      using AL = std::integral_constant<int, PG_ACT_LEAKY>;
      if (a_has_mask) {
        if (p.act == PG_ACT_RELU) emit(T1{}, AR{}); else if (p.act == PG_ACT_LEAKY) emit(T1{}, AL{}); else emit(T1{}, AN{});
      } else {
        if (p.act == PG_ACT_RELU) emit(T0{}, AR{}); else if (p.act == PG_ACT_LEAKY) emit(T0{}, AL{}); else emit(T0{}, AN{});
      }
      if constexpr (BMODE == B_NT) {
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
          float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
          if (b_edge && !(nb0 + (tid >> 3) + 32 * i < p.n_cnt)) { v[0] = v[1] = v[2] = v[3] = 0.f; }
          store4_lp(Bh, BN, (tid >> 3) + 32 * i, (tid & 7) * 4, v);
        }
      } else {   // B_NN: this thread holds 4 consecutive n at k = tid&31 -> transposing scatter of single bf16
        const int kr = tid & 31;
#pragma unroll
        for (int i = 0; i < NN_PASS; ++i) {
          const int nl = ((tid >> 5) + 8 * i) * 4;
          float v[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
          if (b_edge && !(nb0 + nl < p.n_cnt)) { v[0] = v[1] = v[2] = v[3] = 0.f; }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned h = pack_bf16(v[e], 0.f);
            Bh[(nl + e) * ASB + kr] = (unsigned short)(h & 0xffffu);
            if constexpr (PREC == 2) {
              const unsigned l = pack_bf16(v[e] - bf16_lo_f32(h), 0.f);
              Bh[BN * ASB + (nl + e) * ASB + kr] = (unsigned short)(l & 0xffffu);
            }
          }
        }
      }
      return;
    }
    if (AMODE == A_VEC) {
      // act((a*x+b)*mask): the activation / mask variant is picked once per tile (uniform), 2-4 VALU per element
      auto emit = [&](auto has_mask, auto act_c) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
          float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
          const float mk[4] = {rmask[i].x, rmask[i].y, rmask[i].z, rmask[i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = fmaf(v[e], raa[i], rab[i]);
            if constexpr (decltype(has_mask)::value) t *= mk[e];
            if constexpr (decltype(act_c)::value == PG_ACT_RELU) t = fmaxf(t, 0.f);
            if constexpr (decltype(act_c)::value == PG_ACT_LEAKY) t = fmaxf(t, 0.2f * t);
            v[e] = t;
          }
          *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * i) * AS + (tid & 7) * 4]) = make_float4(v[0], v[1], v[2], v[3]);
        }
      };
      using T1 = std::integral_constant<bool, true>;
      using T0 = std::integral_constant<bool, false>;
      using AN = std::integral_constant<int, PG_ACT_NONE>;
      using AR = std::integral_constant<int, PG_ACT_RELU>;
      using AL = std::integral_constant<int, PG_ACT_LEAKY>;
      if (a_has_mask) {
        if (p.act == PG_ACT_RELU) emit(T1{}, AR{}); else if (p.act == PG_ACT_LEAKY) emit(T1{}, AL{}); else emit(T1{}, AN{});
      } else {
        if (p.act == PG_ACT_RELU) emit(T0{}, AR{}); else if (p.act == PG_ACT_LEAKY) emit(T0{}, AL{}); else emit(T0{}, AN{});
      }
    } else {
#pragma unroll
      for (int e = 0; e < AS_CNT; ++e) As[s_row * AS + s_ksub + e * (256 / BM)] = rs[e];
    }
    if (BMODE == B_NT) {
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) {
        float4 v = rb[i];
        if (b_edge && !(nb0 + (tid >> 3) + 32 * i < p.n_cnt)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 32 * i) * BSK + (tid & 7) * 4]) = v;
      }
    } else if (BMODE == B_NN) {
      const bool zero = b_edge && !(nb0 + (tid % NN_CPR) * 4 < p.n_cnt);
#pragma unroll
      for (int i = 0; i < NN_PASS; ++i) {
        const int kr = tid / NN_CPR + i * (256 / NN_CPR);
        *reinterpret_cast<float4*>(&Bs[kr * BS + (tid % NN_CPR) * 4]) = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : rb[i];
      }
    } else {
#pragma unroll
      for (int e = 0; e < BS_CNT; ++e) {
        int kk, nl;
        if (p.w_transposed) { nl = tid % BN; kk = tid / BN + e * (256 / BN); }
        else { kk = tid & 31; nl = (tid >> 5) + 8 * e; }
        Bs[nl * BSK + kk] = rbs[e];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm0 = (wave / WGN) * (TM * 32);
  const int wn0 = (wave % WGN) * (TN * 32);
  const int l31 = lane & 31, lhi = lane >> 5;

  // operand fetch for k-group g (8 k's): element e pairs k = 8g+e (lanes<32) with k = 8g+4+e (lanes>=32).
  // The ds_reads are inline asm on purpose: hipcc's own waitcnt insertion drains lgkmcnt(0) before every MFMA group
  // (scalar loads in flight make its LDS counts "out of order"), which exposes the LDS latency three times per tile
  // with nothing queued on the matrix pipe (measured 129 vs 150 TFLOP/s for the MFMA-only loop).  Issued this way
  // the compiler does not track them; the counted waits below (lds_wait<N>) do, and any younger LDS/SMEM operation
  // the compiler adds only makes those waits stricter, never weaker.
  constexpr int NRD = TM + (B_KN ? 4 * TN : TN);           // LDS reads per fetch (<= 10)
  const unsigned fa_base = (unsigned)(size_t)As0 + (unsigned)((wm0 + l31) * AS + lhi * 4) * 4u;
  const unsigned fb_base = B_KN ? (unsigned)(size_t)Bs0 + (unsigned)((lhi * 4) * BS + wn0 + l31) * 4u
                               : (unsigned)(size_t)Bs0 + (unsigned)((wn0 + l31) * BSK + lhi * 4) * 4u;
  static_assert(TM <= 2 && TN <= 2, "fetch below is written out for at most 2x2 MFMA tiles per wave");
  auto fetch = [&](int stage, auto gc, float (&fa)[TM][4], float (&fb)[TN][4]) {
    constexpr int G = decltype(gc)::value;
    if constexpr ((PG_ABLATE & 16) != 0) { if (stage >= 0) return; }   // diagnostic: operands stay constant
    const unsigned aa = fa_base + (unsigned)(stage * A_SZ) * 4u;     // one v_add per operand; the rest are immediates
    const unsigned bb = fb_base + (unsigned)(stage * B_SZ) * 4u;
    f32x4 v;
    lds_read128<(G * 8) * 4>(v, aa);
    fa[0][0] = v[0]; fa[0][1] = v[1]; fa[0][2] = v[2]; fa[0][3] = v[3];
    if constexpr (TM > 1) {
      f32x4 w;
      lds_read128<(G * 8 + 32 * AS) * 4>(w, aa);
      fa[TM - 1][0] = w[0]; fa[TM - 1][1] = w[1]; fa[TM - 1][2] = w[2]; fa[TM - 1][3] = w[3];
    }
    if constexpr (B_KN) {
      lds_read32<((G * 8 + 0) * BS) * 4>(fb[0][0], bb);
      lds_read32<((G * 8 + 1) * BS) * 4>(fb[0][1], bb);
      lds_read32<((G * 8 + 2) * BS) * 4>(fb[0][2], bb);
      lds_read32<((G * 8 + 3) * BS) * 4>(fb[0][3], bb);
      if constexpr (TN > 1) {
        lds_read32<((G * 8 + 0) * BS + 32) * 4>(fb[TN - 1][0], bb);
        lds_read32<((G * 8 + 1) * BS + 32) * 4>(fb[TN - 1][1], bb);
        lds_read32<((G * 8 + 2) * BS + 32) * 4>(fb[TN - 1][2], bb);
        lds_read32<((G * 8 + 3) * BS + 32) * 4>(fb[TN - 1][3], bb);
      }
    } else {
      f32x4 u;
      lds_read128<(G * 8) * 4>(u, bb);
      fb[0][0] = u[0]; fb[0][1] = u[1]; fb[0][2] = u[2]; fb[0][3] = u[3];
      if constexpr (TN > 1) {
        f32x4 w;
        lds_read128<(G * 8 + 32 * BSK) * 4>(w, bb);
        fb[TN - 1][0] = w[0]; fb[TN - 1][1] = w[1]; fb[TN - 1][2] = w[2]; fb[TN - 1][3] = w[3];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  using G2 = std::integral_constant<int, 2>;
  using G3 = std::integral_constant<int, 3>;
  // wait until at most `n` of this wave's LDS/SMEM operations are outstanding, then fence the scheduler
#define PG_LDS_WAIT(n)                                              \
  do {                                                              \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");      \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)

  // 16 MFMAs of one k-group (8 k's) from one operand register set
  auto mfma_group = [&](const float (&fa)[TM][4], const float (&fb)[TN][4]) {
    if constexpr (!(PG_ABLATE & 4)) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i][e]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j][e]));
      }
    }
  };
  static_assert(BK == 32, "the K loop below is written for 4 k-groups per tile");

  if constexpr (PREC == 3) {
    // -------- bf16 DATA path: both operands are bf16 tensors in HBM (activations already normalised / activated /
    // masked by pg_materialise_bf16, weights converted by pg_weights_to_bf16), K-contiguous, and go global -> LDS by
    // DMA exactly like the fp32 data-gradient loaders: rows of 64 bf16 = 128 bytes, 16-byte chunks XOR-swizzled,
    // zero page for padding taps and the N tail, multi-source (virtual concat) A.  v_mfma_f32_32x32x16_bf16: a lane
    // supplies 8 consecutive k's = ONE ds_read_b128 per operand and k-step, 4 k-steps per tile, fp32 accumulators
    // and the same epilogues (raw fp32 output, fused statistics, data-gradient scatter) as the fp32 kernel.
    static_assert(AMODE == A_VEC && BMODE == B_NT && DMA == 0, "bf16 data path: K-contiguous operands only");
    if (kt0 >= kt1) return;
    constexpr int A_SZ3 = BM * 32, B_SZ3 = BN * 32;              // floats (128-byte rows)
    static_assert(A_SZ3 <= A_SZ && B_SZ3 <= B_SZ, "bf16 tiles must fit the fp32 staging area");
    float* const As3 = smem;
    float* const Bs3 = smem + 2 * A_SZ;
    const char* const zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));
    const char* const wp = uniform_ptr(reinterpret_cast<const char*>(p.W) + w_off_g);
    const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
    const char* pa[A_ROWS];
    const char* pb[B_ROWS];
    int ld_kt = kt0, ld_tap = kt0 / cpt, ld_ci = kt0 - (kt0 / cpt) * cpt;
    auto rebuild = [&]() {
      const int tp = taps_l[ld_tap];
      const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
      const int cc = ld_ci * 64;
      const char* sp = reinterpret_cast<const char*>(p.src[0].ptr);
      int sC = p.src[0].C, cs = 0;
#pragma unroll
      for (int q = 1; q < PG_MAX_SRC; ++q)
        if (q < p.nsrc && cc >= p.cstart[q]) { sp = reinterpret_cast<const char*>(p.src[q].ptr); sC = p.src[q].C; cs = p.cstart[q]; }
      sp = uniform_ptr(sp + a_off_g);
      const int cl = cc - cs + chunk * 8;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const int iy = a_iy[i] + dyv, ix = a_ix[i] + dxv;
        const bool ok = (a_n[i] >= 0) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
        const long off = ((long)((a_n[i] * p.Hi + iy) * p.Wi + ix) * sC + cl) * 2;
        pa[i] = ok ? sp + off : zero_pg + (tid & 7) * 16;
      }
      const int base = (tp >> 16) * p.wCout;
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) {
        const int n = nb0 + (tid >> 3) + 32 * i;
        const long off = ((long)(base + p.n_off + n) * p.wCin + cc + chunk * 8) * 2;
        pb[i] = (n < p.n_cnt) ? wp + off : zero_pg + (tid & 7) * 16;
      }
    };
    auto advance = [&]() {
      if (ld_kt + 1 < kt1) {
        ++ld_kt;
        if (++ld_ci == cpt) { ld_ci = 0; ++ld_tap; rebuild(); }
        else {
          bool src_edge = false;
#pragma unroll
          for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && ld_ci * 64 == p.cstart[q]) src_edge = true;
          if (src_edge) rebuild();
          else {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) pa[i] += 128;
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) pb[i] += 128;
          }
        }
      }
    };
    auto issue = [&](int stage) {
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>((PG_ABLATE & 256) ? wp + tid * 16 : pa[i]), As3 + stage * A_SZ3 + (i * 4 + wave) * 256, 16, 0, 0);
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>((PG_ABLATE & 256) ? wp + tid * 16 : pb[i]), Bs3 + stage * B_SZ3 + (i * 4 + wave) * 256, 16, 0, 0);
    };
    const int swr = (l31 >> 1) & 7;
    unsigned fa3[4], fb3[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa3[ks] = (unsigned)(size_t)As3 + (unsigned)((wm0 + l31) * 128) + (unsigned)(((2 * ks + lhi) ^ swr) * 16);
      fb3[ks] = (unsigned)(size_t)Bs3 + (unsigned)((wn0 + l31) * 128) + (unsigned)(((2 * ks + lhi) ^ swr) * 16);
    }
    rebuild();
    issue(0);
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      // this tile's operand reads go first (their latency runs under the DMA issue), then the next tile's DMA into
      // the other stage (past the end: the last tile again, into a stage nobody reads), then the MFMAs
      f32x4 va[4][TM], vb[4][TN];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const unsigned aa = fa3[ks] + (unsigned)(stage * A_SZ3) * 4u, bb = fb3[ks] + (unsigned)(stage * B_SZ3) * 4u;
        lds_read128<0>(va[ks][0], aa);
        if constexpr (TM > 1) lds_read128<32 * 128>(va[ks][TM - 1], aa);
        lds_read128<0>(vb[ks][0], bb);
        if constexpr (TN > 1) lds_read128<32 * 128>(vb[ks][TN - 1], bb);
      }
      __builtin_amdgcn_sched_barrier(0);
      issue(stage ^ 1);
      advance();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // operands of k-step ks have landed when at most the reads of the later k-steps are outstanding
        if (ks == 0) PG_LDS_WAIT(3 * (TM + TN)); else if (ks == 1) PG_LDS_WAIT(2 * (TM + TN));
        else if (ks == 2) PG_LDS_WAIT(TM + TN); else PG_LDS_WAIT(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va[ks][i]),
                                                                 __builtin_bit_cast(bf16x8, vb[ks][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's DMA has landed
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      stage ^= 1;
    }
  } else if constexpr (DMA != 0) {
    // -------- fp32 K loop with LDS-DMA loaders (global_load_lds_dwordx4: global -> LDS, no register round trip, no
    // ds_write, no prologue math).  Usable when the A operand needs no prologue — the upstream gradient of a
    // data-gradient launch (one source, no deferred affine, no mask, no activation) — and for the weights.
    // A wave instruction moves 64 x 16 B = 1 KB into CONTIGUOUS LDS, so tiles are unpadded and bank conflicts are
    // avoided by permuting which global chunk a lane moves: A [m][32 k] keeps chunk c of row r in slot c ^ ((r>>1)&7)
    // (conflict-free ds_read_b128 fragments); B [k][BN] keeps row k's 32-float halves swapped when (k>>2)&1 (the two
    // k's an MFMA lane pair reads sit 4 rows apart).  Zero rows (padding taps, N tail) are read from a zero page.
    // Tile t+1 is DMA'd into the other stage while tile t is multiplied; vmcnt(0) + one barrier per tile publish it.
    static_assert(AMODE == A_VEC && BMODE == B_NN && PREC == 0, "DMA loaders: data-gradient operands only");
    if (kt0 >= kt1) return;
    constexpr int A_SZD = BM * BK, B_SZD = BK * BN;              // floats, unpadded
    float* const AsD = smem;
    float* const BsD = smem + 2 * A_SZ;
    const char* const zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));
    const char* const srcp = uniform_ptr(reinterpret_cast<const char*>(p.src[0].ptr));
    const char* const wp = uniform_ptr(reinterpret_cast<const char*>(p.W));
    const int srcC = p.src[0].C;
    const int achunk = (tid & 7) ^ ((tid >> 4) & 7);             // swizzled 16-byte chunk of the row this lane moves
    const char* pa[A_ROWS];
    const char* pb[NN_PASS];
    int ld_kt = kt0, ld_tap = kt0 / cpt, ld_ci = kt0 - (kt0 / cpt) * cpt;
    auto rebuild = [&]() {                                       // (tap, channel tile) -> per-row global pointers
      const int tp = taps_l[ld_tap];
      const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
      const int cc = ld_ci * BK;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const int iy = a_iy[i] + dyv, ix = a_ix[i] + dxv;
        const bool ok = (a_n[i] >= 0) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
        const long off = ((long)((a_n[i] * p.Hi + iy) * p.Wi + ix) * srcC + cc + achunk * 4) * 4;
        pa[i] = ok ? srcp + off : zero_pg + (tid & 7) * 16;
      }
      const int base = (tp >> 16) * p.wCout;
#pragma unroll
      for (int i = 0; i < NN_PASS; ++i) {
        const int kr = tid / NN_CPR + i * (256 / NN_CPR);
        const int nchunk = (tid % NN_CPR) ^ (((kr >> 2) & 1) << 3);
        const int n = nb0 + nchunk * 4;
        const long off = ((long)(base + cc + kr) * p.wCin + p.n_off + n) * 4;
        pb[i] = (n < p.n_cnt) ? wp + off : zero_pg + (tid & 31) * 16;
      }
    };
    auto advance = [&]() {
      if (ld_kt + 1 < kt1) {
        ++ld_kt;
        if (++ld_ci == cpt) { ld_ci = 0; ++ld_tap; rebuild(); }
        else {
#pragma unroll
          for (int i = 0; i < A_ROWS; ++i) pa[i] += BK * 4;
#pragma unroll
          for (int i = 0; i < NN_PASS; ++i) pb[i] += (long)BK * p.wCin * 4;
        }
      }
    };
    auto issue_a = [&](int stage) {
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(pa[i]), AsD + stage * A_SZD + (i * 4 + wave) * 256, 16, 0, 0);
    };
    auto issue_b = [&](int stage) {
#pragma unroll
      for (int i = 0; i < NN_PASS; ++i)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(pb[i]), BsD + stage * B_SZD + (i * 4 + wave) * 256, 16, 0, 0);
    };
    // operand fetch addresses (per lane): A slot of chunk 2G+lhi in row wm0+l31 (+32 rows per i); B column position
    const int swr = (l31 >> 1) & 7;
    unsigned faD[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      faD[g] = (unsigned)(size_t)AsD + (unsigned)((wm0 + l31) * BK * 4) + (unsigned)((((2 * g + lhi) ^ swr)) * 16);
    unsigned fbD[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fbD[j] = (unsigned)(size_t)BsD + (unsigned)((lhi * 4) * BN + ((wn0 + j * 32 + l31) ^ (lhi << 5))) * 4u;
    auto fetch_d = [&](int stage, auto gc, float (&fa)[TM][4], float (&fb)[TN][4]) {
      constexpr int G = decltype(gc)::value;
      const unsigned aa = faD[G] + (unsigned)(stage * A_SZD) * 4u;
      f32x4 v;
      lds_read128<0>(v, aa);
      fa[0][0] = v[0]; fa[0][1] = v[1]; fa[0][2] = v[2]; fa[0][3] = v[3];
      if constexpr (TM > 1) {
        f32x4 w2;
        lds_read128<32 * BK * 4>(w2, aa);
        fa[TM - 1][0] = w2[0]; fa[TM - 1][1] = w2[1]; fa[TM - 1][2] = w2[2]; fa[TM - 1][3] = w2[3];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned bb = fbD[j] + (unsigned)(stage * B_SZD) * 4u;
        lds_read32<((G * 8 + 0) * BN) * 4>(fb[j][0], bb);
        lds_read32<((G * 8 + 1) * BN) * 4>(fb[j][1], bb);
        lds_read32<((G * 8 + 2) * BN) * 4>(fb[j][2], bb);
        lds_read32<((G * 8 + 3) * BN) * 4>(fb[j][3], bb);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    float fa[2][TM][4] = {}, fb[2][TN][4] = {};
    rebuild();
    issue_a(0); issue_b(0);
    advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    fetch_d(0, G0{}, fa[0], fb[0]);
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      // the next tile goes into the other stage (past the end: the last tile again, into a stage nobody reads)
      issue_a(stage ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      fetch_d(stage, G1{}, fa[1], fb[1]);
      PG_LDS_WAIT(NRD);
      __builtin_amdgcn_s_setprio(1);
      mfma_group(fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
      issue_b(stage ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      fetch_d(stage, G2{}, fa[0], fb[0]);
      PG_LDS_WAIT(NRD);
      mfma_group(fa[1], fb[1]);
      __builtin_amdgcn_sched_barrier(0);
      advance();
      __builtin_amdgcn_sched_barrier(0);
      fetch_d(stage, G3{}, fa[1], fb[1]);
      PG_LDS_WAIT(NRD);
      mfma_group(fa[0], fb[0]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // DMA of the next tile landed; own LDS reads done
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more) fetch_d(stage ^ 1, G0{}, fa[0], fb[0]);
      mfma_group(fa[1], fb[1]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      stage ^= 1;
    }
  } else {
  constexpr bool PIPE = !LP && AMODE == A_VEC && BMODE != B_SCALAR && (PG_ABLATE & 127) == 0;   // bits >= 1024: epilogue diagnostics
  if constexpr (PIPE) {
    // -------- software-pipelined fp32 K loop (vector loaders).
    // The loader work of a tile is cut into 4 chunks (thread rows i with i % 4 == c) and chunk c travels with one of
    // the four 16-MFMA groups of an iteration:  region = { global loads of chunk c-1 (tile t+2) | activation math +
    // ds_write of chunk c (tile t+1) | MFMA group }, all straight-line, so the scheduler interleaves the VALU / LDS /
    // VMEM instructions between the MFMAs of the SAME wave (sched_group_barrier pattern below) instead of leaving
    // them to the co-resident workgroup.  Address updates (the only branchy part: a (tap, source) change rebuilds a
    // row's offsets) sit between the regions.  Every chunk is loaded three regions (~3/4 tile) before it is stored.
    // All loads and LDS stores are unconditional: past the last tile the cursor stops advancing (rows are re-read and
    // written into an LDS stage nobody reads again).
    if (kt0 >= kt1) return;               // empty split (block-uniform): contributes nothing
    constexpr int B_CH = (BMODE == B_NT) ? B_ROWS : NN_PASS;
    const float slope = act_slope(p.act);
    const char* const w_base = uniform_ptr(reinterpret_cast<const char*>(p.W));
    int ld_kt = kt0, ld_tap = kt0 / cpt, ld_ci = kt0 - (kt0 / cpt) * cpt;
    int cur_tap = -1, cur_src = -1, cur_btap = -1;
    bool chgA = true, chgB = true, has_mask = false;
    unsigned stepA = 0, stepB = 0;
    int srcC = 0, cl = 0, dyv = 0, dxv = 0, jsrc = 0, bbase = 0, bcc = 0;
    bool bzero[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      bzero[i] = (BMODE == B_NT) ? !(nb0 + (tid >> 3) + 32 * i < p.n_cnt) : !(nb0 + (tid % NN_CPR) * 4 < p.n_cnt);

    auto set_tile = [&](bool adv) {      // scalars of the tile the cursor points at (wave-uniform)
      const int cc = ld_ci * BK;
      int j = 0;
#pragma unroll
      for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && cc >= p.cstart[q]) j = q;
      chgA = adv && (ld_tap != cur_tap || j != cur_src);
      chgB = adv && ld_tap != cur_btap;
      stepA = adv ? (unsigned)(BK * 4) : 0u;
      stepB = adv ? ((BMODE == B_NT) ? (unsigned)(BK * 4) : (unsigned)(BK * p.wCin) * 4u) : 0u;
      if (chgA) {
        const float* sp = p.src[0].ptr; const float* sm = p.src[0].mask; int sC = p.src[0].C, cs = 0;
#pragma unroll
        for (int q = 1; q < PG_MAX_SRC; ++q)
          if (q == j) { sp = p.src[q].ptr; sm = p.src[q].mask; sC = p.src[q].C; cs = p.cstart[q]; }
        a_base = uniform_ptr(reinterpret_cast<const char*>(sp));
        has_mask = sm != nullptr;
        m_base = uniform_ptr(reinterpret_cast<const char*>(has_mask ? sm : kOnes));
        srcC = sC; jsrc = j;
        cl = cc - cs + (tid & 7) * 4;
        const int tp = taps_l[ld_tap];
        dyv = (int)(signed char)(tp & 0xff); dxv = (int)(signed char)((tp >> 8) & 0xff);
        cur_tap = ld_tap; cur_src = j;
      }
      if (chgB) {
        bbase = (taps_l[ld_tap] >> 16) * p.wCout;
        bcc = cc;
        cur_btap = ld_tap;
      }
    };
    auto advance = [&]() {
      const bool adv = ld_kt + 1 < kt1;
      if (adv) { ++ld_kt; if (++ld_ci == cpt) { ld_ci = 0; ++ld_tap; } }
      set_tile(adv);
    };
    auto update_chunk = [&](auto cc_) {
      constexpr int c = decltype(cc_)::value;
#pragma unroll
      for (int i = c; i < A_ROWS; i += 4) {
        if (chgA) {
          const int iy = a_iy[i] + dyv, ix = a_ix[i] + dxv;
          const bool ok = a_n[i] >= 0 && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
          const int row = (tid >> 3) + 32 * i;
          raa[i] = ok ? affs[(row * PG_MAX_SRC + jsrc) * 2] : 0.f;      // zero padding = affine (0, 0)
          rab[i] = ok ? affs[(row * PG_MAX_SRC + jsrc) * 2 + 1] : 0.f;
          const int nn = ok ? a_n[i] : 0;
          const unsigned pix = ok ? (unsigned)((nn * p.Hi + iy) * p.Wi + ix) : 0u;
          aoff[i] = (pix * (unsigned)srcC + (unsigned)cl) * 4u;
          moff[i] = has_mask ? ((unsigned)(nn * srcC + cl) * 4u) : ((unsigned)(cl & 511) * 4u);
        } else {
          aoff[i] += stepA; moff[i] += stepA;
        }
      }
#pragma unroll
      for (int i = c; i < B_CH; i += 4) {
        if (chgB) {
          if constexpr (BMODE == B_NT) {
            const int n = nb0 + (tid >> 3) + 32 * i;
            boff[i] = (unsigned)((bbase + p.n_off + (n < p.n_cnt ? n : 0)) * p.wCin + bcc + (tid & 7) * 4) * 4u;
          } else {
            const int kr = tid / NN_CPR + i * (256 / NN_CPR);
            const int n = nb0 + (tid % NN_CPR) * 4;
            boff[i] = (unsigned)((bbase + bcc + kr) * p.wCin + p.n_off + (n < p.n_cnt ? n : 0)) * 4u;
          }
        } else {
          boff[i] += stepB;
        }
      }
    };
    auto load_chunk = [&](auto cc_) {
      constexpr int c = decltype(cc_)::value;
#pragma unroll
      for (int i = c; i < A_ROWS; i += 4) {
        if constexpr ((PG_ABLATE & 256) != 0) {   // diagnostic: every load hits the same 4 KB (L1-resident)
          ra[i] = ldg128(a_base, (unsigned)tid * 16u);
          rmask[i] = ldg128(m_base, (unsigned)(tid & 31) * 16u);
        } else {
          ra[i] = ldg128(a_base, aoff[i]);
          if constexpr (!(PG_ABLATE & 512) && !NOMASK) rmask[i] = ldg128(m_base, moff[i]);
        }
      }
#pragma unroll
      for (int i = c; i < B_CH; i += 4) {
        if constexpr ((PG_ABLATE & 256) != 0) rb[i] = ldg128(w_base, (unsigned)tid * 16u);
        else rb[i] = ldg128(w_base, boff[i]);
      }
    };
    auto store_chunk = [&](int stage, auto cc_) {
      constexpr int c = decltype(cc_)::value;
      float* As = As0 + stage * A_SZ;
      float* Bs = Bs0 + stage * B_SZ;
#pragma unroll
      for (int i = c; i < A_ROWS; i += 4) {
        float v[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
        const float mk[4] = {rmask[i].x, rmask[i].y, rmask[i].z, rmask[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr ((PG_ABLATE & 512) != 0) { v[e] = fmaxf(fmaf(v[e], raa[i], rab[i]), 0.f); continue; }
          float t = fmaf(v[e], raa[i], rab[i]);
          if constexpr (!NOMASK) t *= mk[e];
          v[e] = fmaxf(t, slope * t);          // slope 1 / 0 / 0.2 = none / relu / leaky-relu, no branch
        }
        *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * i) * AS + (tid & 7) * 4]) = make_float4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int i = c; i < B_CH; i += 4) {
        float4 v = rb[i];
        if (bzero[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (BMODE == B_NT)
          *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 32 * i) * BSK + (tid & 7) * 4]) = v;
        else
          *reinterpret_cast<float4*>(&Bs[(tid / NN_CPR + i * (256 / NN_CPR)) * BS + (tid % NN_CPR) * 4]) = v;
      }
    };
    // interleave request for one region: the chunk's VMEM reads first, then 1 MFMA : VPER VALU, the LDS writes late
    constexpr int NM = 4 * TM * TN;
    constexpr int VPER = (28 + NM - 1) / NM;
    auto interleave = [&]() {
      __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
        if (k == NM / 2 || k == NM - 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;

    float fa[2][TM][4] = {}, fb[2][TN][4] = {};
    // prologue: tile kt0 -> stage 0; tile kt0+1 in registers, its chunk 0 already in stage 1; chunk 0 -> tile kt0+2
    set_tile(true);
    update_chunk(C0{}); update_chunk(C1{}); update_chunk(C2{}); update_chunk(C3{});
    load_chunk(C0{}); load_chunk(C1{}); load_chunk(C2{}); load_chunk(C3{});
    store_chunk(0, C0{}); store_chunk(0, C1{}); store_chunk(0, C2{}); store_chunk(0, C3{});
    advance();
    update_chunk(C0{}); update_chunk(C1{}); update_chunk(C2{}); update_chunk(C3{});
    load_chunk(C0{}); load_chunk(C1{}); load_chunk(C2{}); load_chunk(C3{});
    store_chunk(1, C0{});
    advance();
    update_chunk(C0{});
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    fetch(0, G0{}, fa[0], fb[0]);
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      fetch(stage, G1{}, fa[1], fb[1]);
      PG_LDS_WAIT(NRD);
      __builtin_amdgcn_s_setprio(1);
      load_chunk(C0{}); store_chunk(stage ^ 1, C1{}); mfma_group(fa[0], fb[0]); interleave();
      __builtin_amdgcn_sched_barrier(0);
      update_chunk(C1{});
      __builtin_amdgcn_sched_barrier(0);
      fetch(stage, G2{}, fa[0], fb[0]);
      PG_LDS_WAIT(NRD);
      load_chunk(C1{}); store_chunk(stage ^ 1, C2{}); mfma_group(fa[1], fb[1]); interleave();
      __builtin_amdgcn_sched_barrier(0);
      update_chunk(C2{});
      __builtin_amdgcn_sched_barrier(0);
      fetch(stage, G3{}, fa[1], fb[1]);
      PG_LDS_WAIT(NRD);
      load_chunk(C2{}); store_chunk(stage ^ 1, C3{}); mfma_group(fa[0], fb[0]); interleave();
      __builtin_amdgcn_sched_barrier(0);
      update_chunk(C3{});
      __builtin_amdgcn_sched_barrier(0);
      PG_LDS_WAIT(0);                      // this wave's reads of `stage` and writes of `stage^1` are complete
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more) fetch(stage ^ 1, G0{}, fa[0], fb[0]);
      load_chunk(C3{}); store_chunk(stage, C0{}); mfma_group(fa[1], fb[1]); interleave();   // chunk 0 of tile kt+2
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      advance();
      update_chunk(C0{});
      __builtin_amdgcn_sched_barrier(0);
      stage ^= 1;
    }
  } else if constexpr (LP) {
    // -------- bf16 / bf16x3 K loop: 2 k-steps of 16 per tile on v_mfma_f32_32x32x16_bf16; lane (m=l31) reads its 8
    // consecutive k's (16 bytes) per operand and k-step.  The loaders (same as fp32) bound this path.
    if (kt0 < kt1) {
      load_tile(kt0);
      store_tile(0);
      if (kt0 + 1 < kt1) load_tile(kt0 + 1);
    }
    __syncthreads();
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) {
        store_tile(stage ^ 1);
        if (kt + 2 < kt1) load_tile(kt + 2);
      }
      const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As0 + stage * A_SZ);
      const unsigned short* Bh = reinterpret_cast<const unsigned short*>(Bs0 + stage * B_SZ);
      uint4 a[NPART][2][TM], b[NPART][2][TN];
#pragma unroll
      for (int q = 0; q < NPART; ++q)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            a[q][ks][i] = *reinterpret_cast<const uint4*>(Ah + q * BM * ASB + (wm0 + i * 32 + l31) * ASB + ks * 16 + lhi * 8);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            b[q][ks][j] = *reinterpret_cast<const uint4*>(Bh + q * BN * ASB + (wn0 + j * 32 + l31) * ASB + ks * 16 + lhi * 8);
        }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (PREC == 2) {   // small terms first
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1][ks][i]),
                                                                   __builtin_bit_cast(bf16x8, b[0][ks][j]), acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0][ks][i]),
                                                                   __builtin_bit_cast(bf16x8, b[1][ks][j]), acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0][ks][i]),
                                                                 __builtin_bit_cast(bf16x8, b[0][ks][j]), acc[i][j], 0, 0, 0);
          }
      __syncthreads();
      stage ^= 1;
    }
  } else {
  // K loop.  Two LDS stages, two operand register sets (A/B).  Per tile:  [write tile t+1 to the other stage, issue the
  // global loads of tile t+2]  g1<-LDS | MFMA g0 | g2<-LDS | MFMA g1 | g3<-LDS | MFMA g2 | BARRIER | next g0<-LDS |
  // MFMA g3.  Every LDS read of the current stage is complete before the barrier (so the next iteration may overwrite
  // it), the other stage is complete after it, and each operand fetch has 16 MFMAs (1024 cycles) to land behind.
  float fa[2][TM][4] = {}, fb[2][TN][4] = {};
  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile(0);
    if (kt0 + 1 < kt1) load_tile(kt0 + 1);
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  if (kt0 < kt1) fetch(0, G0{}, fa[0], fb[0]);
  int stage = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    if (more) {
      if constexpr (!(PG_ABLATE & 2)) store_tile(stage ^ 1);   // registers hold tile kt+1 (fetched one iteration ago)
      if constexpr (!(PG_ABLATE & 1)) { if (kt + 2 < kt1) load_tile(kt + 2); }
    }
    // one operand set is always a full fetch (NRD reads) ahead of the MFMAs that consume the other one
    fetch(stage, G1{}, fa[1], fb[1]);
    PG_LDS_WAIT(NRD);                      // set A (fetched behind the previous barrier) has landed
    __builtin_amdgcn_s_setprio(1);         // MFMA phase outranks the co-resident workgroup's loader phase
    mfma_group(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(stage, G2{}, fa[0], fb[0]);
    PG_LDS_WAIT(NRD);                      // set B = g1
    mfma_group(fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(stage, G3{}, fa[1], fb[1]);
    PG_LDS_WAIT(NRD);                      // set A = g2
    mfma_group(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    PG_LDS_WAIT(0);                        // every read of this stage has completed (the next store may overwrite it)
    if constexpr (!(PG_ABLATE & 8)) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(stage ^ 1, G0{}, fa[0], fb[0]);
    mfma_group(fa[1], fb[1]);              // g3 landed before the barrier
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    stage ^= 1;
  }
  }   // fp32 K loop
  }   // !DMA
  if (kt0 >= kt1) return;   // empty split: contributes nothing

  // ------------------------------------------------------------------ epilogue
  if constexpr ((PG_ABLATE & 4096) != 0) { if (acc[0][0][0] != 12345.678f) return; }   // diagnostic: K loop only
  if (p.part != nullptr) {
    // split-K with a workspace: plain stores of this split's partial sums; bias / activation derivative / masks /
    // statistics are applied once by splitk_fixup_kernel.  (Float atomics into the destination cost ~30 us per launch
    // on the small deep layers — every split hammers the same few hundred KB — plus a memset and, for layers followed
    // by a norm, a separate statistics pass.)
    float* pp = p.part + (long)split * p.part_stride;
    if constexpr (TM == 2 && (TN == 1 || TN == 2)) {
      if (p.vec_out) {
        __syncthreads();                                   // every wave is done with the operand stages
        float st0[2] = {0.f, 0.f}, st1[2] = {0.f, 0.f};
        vec_store_64x64<TN>(acc, smem + wave * (32 * (32 * TN + 4)), rows, wm0, lane, pp, p.n_cnt, p.Ho, p.Wo, nb0 + wn0 + (lane % (8 * TN)) * 4,
                        make_float4(0.f, 0.f, 0.f, 0.f), false, 0, st0, st1, nullptr);
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ng = nb0 + wn0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const RowInfo ri = rows[row];
          if (ri.n >= 0 && ng < p.n_cnt) pp[(long)((ri.n * p.Ho + ri.oy) * p.Wo + ri.ox) * p.n_cnt + ng] = acc[i][j][r];
        }
      }
    return;
  }
  const bool atomic = p.ksplit > 1;
  const bool do_stats = p.stats != nullptr && p.epilogue == 0;      // host: only with ksplit == 1
  int stat_n0 = 0;
  if (do_stats) {                                                   // first valid sample of this wave's rows
    const int nn = rows[wm0 + (lane & (TM * 32 - 1))].n;
    int nf = nn >= 0 ? nn : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nf = min(nf, __shfl_xor(nf, o));
    stat_n0 = nf;
  }
  float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};
  const int nbw = __builtin_amdgcn_readfirstlane(nb0 + wn0);   // wave-uniform first column (SGPR: uniform scatter path)
  bool vec_done = false;
  if constexpr (TM == 2 && (TN == 1 || TN == 2)) {
    if (p.vec_out && p.epilogue == 0 && !atomic && p.out_act == PG_OUT_NONE) {
      __syncthreads();                                     // every wave is done with the operand stages
      const int ngc = nb0 + wn0 + (lane % (8 * TN)) * 4;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && ngc < p.n_cnt) bv = *reinterpret_cast<const float4*>(p.bias + ngc);
      if constexpr (PREC == 3) {      // bf16 STORAGE exists on the bf16 data path only
        if (p.out_bf16)
          vec_store_64x64<TN, true>(acc, smem + wave * (32 * (32 * TN + 4)), rows, wm0, lane, out_g, p.n_cnt, p.Ho, p.Wo, ngc, bv, do_stats,
                                    stat_n0, st_s, st_q, p.stats);
        else
          vec_store_64x64<TN, false>(acc, smem + wave * (32 * (32 * TN + 4)), rows, wm0, lane, out_g, p.n_cnt, p.Ho, p.Wo, ngc, bv, do_stats,
                                     stat_n0, st_s, st_q, p.stats);
      } else {
        vec_store_64x64<TN, false>(acc, smem + wave * (32 * (32 * TN + 4)), rows, wm0, lane, out_g, p.n_cnt, p.Ho, p.Wo, ngc, bv, do_stats,
                                   stat_n0, st_s, st_q, p.stats);
      }
      vec_done = true;
    }
  }
  if constexpr (TM == 2 && (TN == 1 || TN == 2) && BMODE != B_SCALAR) {
    if (p.vec_dst && p.epilogue == 1 && !atomic) {
      __syncthreads();                                     // every wave is done with the operand stages
      const int ngc = nb0 + wn0 + (lane % (8 * TN)) * 4;   // first of this lane's 4 columns
      const bool cval = ngc < p.n_cnt;
      const int ngs = cval ? ngc : 0;
      // constant-index field picks only (a runtime index into the kernel argument would put it in scratch memory)
      float* gradp = p.dst[0].grad;
      const float *fwd0 = p.dst[0].fwd, *aff0 = p.dst[0].aff, *mask0 = p.dst[0].mask;
      int C = p.dst[0].C, dact = p.dst[0].act, dacc = p.dst[0].accumulate, cst = 0, dfl = p.dst[0].flags;
      // (round 4) fused norm-backward sums: host (conv_impl) keeps pg_dst_t.bsums only when the workgroup's column tile lies
      // in ONE destination (workgroup-uniform pointer) and a sample has >= 64 rows.  Picked in THIS loop: a second pick loop
      // over p.dst[] made the compiler keep a 1.3 KB scratch copy of the kernel argument (tools/check_scratch.sh)
      double* bsq = p.dst[0].bsums;
#pragma unroll
      for (int q = 1; q < PG_MAX_SRC; ++q)
        if (q < p.ndst && ngs >= p.dstart[q]) {
          gradp = p.dst[q].grad; fwd0 = p.dst[q].fwd; aff0 = p.dst[q].aff; mask0 = p.dst[q].mask;
          C = p.dst[q].C; dact = p.dst[q].act; dacc = p.dst[q].accumulate; cst = p.dstart[q]; dfl = p.dst[q].flags;
          bsq = p.dst[q].bsums;
        }
      LaneDst ld;
      ld.grad_bf16 = (dfl & PG_DST_GRAD_BF16) != 0; ld.fwd_bf16 = (dfl & PG_DST_FWD_BF16) != 0;
      ld.has_fwd = fwd0 != nullptr;
      const bool fa_ = aff0 != nullptr && ld.has_fwd;
      ld.has_mask = mask0 != nullptr;
      ld.gradp = gradp; ld.fwdp = ld.has_fwd ? fwd0 : gradp;
      ld.affp = fa_ ? aff0 : kIdentAff; ld.affmul = fa_ ? 2 : 0;
      ld.maskp = ld.has_mask ? mask0 : kOnes;
      ld.C = C; ld.c = ngs - cst;
      ld.dslope = ld.has_fwd ? act_slope(dact) : 1.f;
      ld.accum = dacc != 0;
      ld.bsums = bsq;
      const bool bs_on = __builtin_amdgcn_readfirstlane((int)(bsq != nullptr)) != 0;
      BsAcc ba;
      ba.s[0] = ba.s[1] = ba.q[0] = ba.q[1] = 0.f;
      ba.n0 = 0;
      if (bs_on) {                                             // first valid sample of this wave's rows
        const int nn = rows[wm0 + (lane & (TM * 32 - 1))].n;
        int nf = nn >= 0 ? nn : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nf = min(nf, __shfl_xor(nf, o));
        ba.n0 = nf;
      }
      float* const Tw = smem + wave * (32 * (32 * TN + 4));
      if constexpr (PREC == 3) {
        if (p.dst_io == 0) {
          if (bs_on) vec_scatter_64x64<TN, 0, RowInfo, true, true>(acc, Tw, rows, wm0, lane, ld, cval, p.Ho, p.Wo, &ba);
          else vec_scatter_64x64<TN, 0>(acc, Tw, rows, wm0, lane, ld, cval, p.Ho, p.Wo);
        } else if (p.dst_io == 1) {
          if (bs_on) vec_scatter_64x64<TN, 1, RowInfo, true, true>(acc, Tw, rows, wm0, lane, ld, cval, p.Ho, p.Wo, &ba);
          else vec_scatter_64x64<TN, 1>(acc, Tw, rows, wm0, lane, ld, cval, p.Ho, p.Wo);
        } else vec_scatter_64x64<TN, 2>(acc, Tw, rows, wm0, lane, ld, cval, p.Ho, p.Wo);
      } else {
        if (bs_on) vec_scatter_64x64<TN, 0, RowInfo, true, true>(acc, Tw, rows, wm0, lane, ld, cval, p.Ho, p.Wo, &ba);
        else vec_scatter_64x64<TN, 0>(acc, Tw, rows, wm0, lane, ld, cval, p.Ho, p.Wo);
      }
      if (bs_on) {
        // workgroup-level merge as for the forward statistics: (wave, k) pairs, equal samples merged, one pair of double atomics
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);           // [wave][2 samples][sum r, sum r * f]
        int* redn = reinterpret_cast<int*>(smem + 64);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double ds = wave_sum_d((double)ba.s[k]), dq = wave_sum_d((double)ba.q[k]);
          if (lane == 0) { red[(wave * 2 + k) * 2] = ds; red[(wave * 2 + k) * 2 + 1] = dq; }
        }
        if (lane == 0) redn[wave] = ba.n0;
        __syncthreads();
        if (tid < 8) {
          const int w = tid >> 1, k = tid & 1;
          const int n = redn[w] == 0x7fffffff ? -1 : redn[w] + k;
          double ds = red[tid * 2], dq = red[tid * 2 + 1];
          bool first = true;
          for (int o = 0; o < tid; ++o) {
            const int no = redn[o >> 1] == 0x7fffffff ? -1 : redn[o >> 1] + (o & 1);
            if (no == n) first = false;
          }
          if (first && n >= 0) {
            for (int o = tid + 1; o < 8; ++o) {
              const int no = redn[o >> 1] == 0x7fffffff ? -1 : redn[o >> 1] + (o & 1);
              if (no == n) { ds += red[o * 2]; dq += red[o * 2 + 1]; }
            }
            if (ds != 0.0 || dq != 0.0) {
              const int slot = (blockIdx.x + blockIdx.y * 5 + blockIdx.z * 3) % PG_STAT_SLOTS;
              atomicAdd(&bsq[((long)n * PG_STAT_SLOTS + slot) * 2], ds);
              atomicAdd(&bsq[((long)n * PG_STAT_SLOTS + slot) * 2 + 1], dq);
            }
          }
        }
      }
      vec_done = true;
    }
  }
  // NOTE: keep this free of lambdas that capture `p` and of runtime indices into p's arrays — either makes the
  // compiler keep a scratch-memory copy of the whole kernel argument (and of acc[][] if these loops stay rolled).
  if (!vec_done) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ng = nb0 + wn0 + j * 32 + l31;
      const bool nval = ng < p.n_cnt;
      if (p.epilogue == 0) {
        const float bv = (p.bias && split == 0 && nval) ? p.bias[ng] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const RowInfo ri = rows[row];
          if (ri.n < 0 || !nval) continue;
          float g = acc[i][j][r] + bv;
          if (p.out_act == PG_OUT_TANH) g = tanhf(g);
          float* o = out_g + ((long)ri.n * p.oN + (long)ri.oy * p.oH + (long)ri.ox * p.oW) + (long)ng * p.oC;
          if (atomic) atomicAdd(o, g); else *o = g;
          // fused per-sample statistics of the following norm layer: tiles are sample-major, so a tile holds a short
          // run of consecutive samples; this lane's values go to the accumulator of their offset from the first one
          if (do_stats) {
            const int dn = ri.n - stat_n0;
            if (dn == 0) { st_s[0] += g; st_q[0] = fmaf(g, g, st_q[0]); }
            else if (dn == 1) { st_s[1] += g; st_q[1] = fmaf(g, g, st_q[1]); }
            else stat_spill(p.stats, ri.n, g);
          }
        }
      } else if constexpr (BMODE != B_SCALAR) {   // host guarantees dst_uniform for the vector weight loaders
        // Data-gradient scatter, every destination's channel count a multiple of 32: the 32-column group of this
        // MFMA tile lies in ONE destination, so the descriptor is wave-uniform (SGPRs).  All global loads of the 16
        // rows are straight-line (dummy-but-valid addresses for absent rows) and issued before the first use: loads
        // under divergent or even uniform branches make the compiler drain vmcnt at every join, which serialises
        // 48 latencies per tile (measured: +25 % on the whole contraction).
        const int ng0 = nbw + j * 32;
        if (ng0 >= p.n_cnt) continue;
        // constant-index field picks only: a runtime index into the by-value kernel argument would make the compiler
        // copy the whole descriptor to scratch memory
        float* gradp = p.dst[0].grad;
        const float *fwd0 = p.dst[0].fwd, *aff0 = p.dst[0].aff, *mask0 = p.dst[0].mask;
        int C = p.dst[0].C, dact = p.dst[0].act, dacc = p.dst[0].accumulate, cst = 0;
#pragma unroll
        for (int q = 1; q < PG_MAX_SRC; ++q)
          if (q < p.ndst && ng0 >= p.dstart[q]) {
            gradp = p.dst[q].grad; fwd0 = p.dst[q].fwd; aff0 = p.dst[q].aff; mask0 = p.dst[q].mask;
            C = p.dst[q].C; dact = p.dst[q].act; dacc = p.dst[q].accumulate; cst = p.dstart[q];
          }
        const int c = ng - cst;
        const bool has_fwd = fwd0 != nullptr;
        const float* const fwdp = has_fwd ? fwd0 : gradp;              // dummy reads when there is no activation
        const float dslope = has_fwd ? act_slope(dact) : 1.f;          // slope 1: act' == 1 whatever was read
        // absent affine / mask: identity tables instead of branches (one code path, loads always issued)
        const bool fa_ = aff0 != nullptr && has_fwd, fm_ = mask0 != nullptr;
        const float* const affp = fa_ ? aff0 : kIdentAff;
        const float* const maskp = fm_ ? mask0 : kOnes;
        const int affmul = fa_ ? 2 : 0;
        const bool accum = dacc && !atomic;
        float fz[16], mk[16], old[16];
        float2 ab[16];
        unsigned idx[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const RowInfo ri = rows[row];
          ok[r] = ri.n >= 0;
          const int nn = ok[r] ? ri.n : 0;
          idx[r] = ok[r] ? (unsigned)((nn * p.Ho + ri.oy) * p.Wo + ri.ox) * (unsigned)C + (unsigned)c : (unsigned)c;
          if constexpr ((PG_ABLATE & 1024) != 0) { fz[r] = 1.f; ab[r] = make_float2(1.f, 0.f); mk[r] = 1.f; old[r] = 0.f; continue; }
          fz[r] = fwdp[idx[r]];
          ab[r] = *reinterpret_cast<const float2*>(affp + affmul * nn);
          mk[r] = maskp[fm_ ? nn * C + c : (c & 511)];
          old[r] = 0.f;
        }
        if (accum) {      // uniform; one extra batch of loads (encoder data-gradients add to the skip gradients)
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = gradp[idx[r]];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = fmaf(fz[r], ab[r].x, ab[r].y) * mk[r];
          const float g = fmaf(acc[i][j][r] * mk[r], act_grad_s(z, dslope), old[r]);
          if constexpr ((PG_ABLATE & 2048) != 0) { if (g == 12345.678f) gradp[idx[r]] = g; continue; }
          if (ok[r]) {
            if (atomic) atomicAdd(gradp + idx[r], g); else gradp[idx[r]] = g;
          }
        }
      } else {
        // generic scatter (destination channel counts not multiples of 32): per-lane destination, same arithmetic
        pg_dst_t ds = p.dst[0];
        int cst = 0;
#pragma unroll
        for (int q = 1; q < PG_MAX_SRC; ++q)
          if (q < p.ndst && ng >= p.dstart[q]) { ds = p.dst[q]; cst = p.dstart[q]; }
        const int c = ng - cst;
        const float dslope = act_slope(ds.act);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const RowInfo ri = rows[row];
          if (ri.n < 0 || !nval) continue;
          const long idx = (long)((ri.n * p.Ho + ri.oy) * p.Wo + ri.ox) * ds.C + c;
          float g = acc[i][j][r];
          const float mkv = ds.mask ? ds.mask[(long)ri.n * ds.C + c] : 1.f;
          if (ds.fwd) {
            float z = ds.fwd[idx];
            if (ds.aff) z = fmaf(z, ds.aff[2 * ri.n], ds.aff[2 * ri.n + 1]);
            g *= act_grad_s(z * mkv, dslope);
          }
          g *= mkv;
          if (atomic) atomicAdd(ds.grad + idx, g);
          else if (ds.accumulate) ds.grad[idx] += g;
          else ds.grad[idx] = g;
        }
      }
    }
  }
  }   // !vec_done
  if (do_stats) {
    // block-level reduction (the operand stages in LDS are free now), then ONE pair of double atomics per sample
    // slot and workgroup, spread over PG_STAT_SLOTS addresses per sample
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);           // [wave][2 samples][sum, sumsq]
    int* redn = reinterpret_cast<int*>(smem + 64);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double ds = wave_sum_d((double)st_s[k]), dq = wave_sum_d((double)st_q[k]);
      if (lane == 0) { red[(wave * 2 + k) * 2] = ds; red[(wave * 2 + k) * 2 + 1] = dq; }
    }
    if (lane == 0) redn[wave] = stat_n0;
    __syncthreads();
    if (tid < 8) {                                            // (wave, k) pairs: merge equal samples, then add
      const int w = tid >> 1, k = tid & 1;
      const int n = redn[w] == 0x7fffffff ? -1 : redn[w] + k;
      double ds = red[tid * 2], dq = red[tid * 2 + 1];
      bool first = true;
      for (int o = 0; o < tid; ++o) {
        const int no = redn[o >> 1] == 0x7fffffff ? -1 : redn[o >> 1] + (o & 1);
        if (no == n) first = false;
      }
      if (first && n >= 0) {
        for (int o = tid + 1; o < 8; ++o) {
          const int no = redn[o >> 1] == 0x7fffffff ? -1 : redn[o >> 1] + (o & 1);
          if (no == n) { ds += red[o * 2]; dq += red[o * 2 + 1]; }
        }
        if (ds != 0.0 || dq != 0.0) {
          const int slot = (blockIdx.x + blockIdx.y * 5 + blockIdx.z * 3) % PG_STAT_SLOTS;
          atomicAdd(&p.stats[((long)n * PG_STAT_SLOTS + slot) * 2], ds);
          atomicAdd(&p.stats[((long)n * PG_STAT_SLOTS + slot) * 2 + 1], dq);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- split-K fixup
struct FixupK {
  const float* part; long stride; int ks;
  int ppix;                 // pixels per sample (Ho*Wo)
  int n_cnt;
  int epilogue;
  const float* bias; float* out; double* stats;
  pg_dst_t dst[PG_MAX_SRC];
  int ndst;
  int dstart[PG_MAX_SRC + 1];
  int out_bf16;             // epilogue 0: `out` is bf16 (bf16 STORAGE); the destinations of epilogue 1 carry their own flags
  int has_bs;               // epilogue 1: a destination carries pg_dst_t.bsums (fused sums of the following norm backward)
};

// grid (workgroups per sample, N): every lane owns 4 consecutive columns of one pixel, sums the ks partial tiles and
// applies the epilogue of the launch: 0 = + bias, dense NHWC store, optional per-sample statistics for the following
// norm; 1 = the data-gradient scatter (act'(fwd) / masks / accumulate), same arithmetic as the in-kernel scatter.
__global__ __launch_bounds__(256) void splitk_fixup_kernel(const FixupK p) {
  const int n = blockIdx.y;
  const int q4 = p.n_cnt >> 2;
  const int items = p.ppix * q4;
  float st_s = 0.f, st_q = 0.f;
  float bs_s[PG_MAX_SRC], bs_q[PG_MAX_SRC];       // (round 4) per destination: sum r, sum r * f of this lane's final values
#pragma unroll
  for (int q = 0; q < PG_MAX_SRC; ++q) { bs_s[q] = 0.f; bs_q[q] = 0.f; }
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int pl = it / q4, col = (it - pl * q4) * 4;
    const long pixel = (long)n * p.ppix + pl;
    const long off = pixel * p.n_cnt + col;
    float4 v = *reinterpret_cast<const float4*>(p.part + off);
    int s = 1;
    for (; s + 7 < p.ks; s += 8) {                      // eight independent 16-byte loads in flight (ks is 8..32 on the deep layers)
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(p.part + (s + u) * p.stride + off);
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        v.x += t[u].x + t[u + 1].x; v.y += t[u].y + t[u + 1].y;
        v.z += t[u].z + t[u + 1].z; v.w += t[u].w + t[u + 1].w;
      }
    }
    for (; s + 3 < p.ks; s += 4) {
      const float4 a = *reinterpret_cast<const float4*>(p.part + s * p.stride + off);
      const float4 b = *reinterpret_cast<const float4*>(p.part + (s + 1) * p.stride + off);
      const float4 c = *reinterpret_cast<const float4*>(p.part + (s + 2) * p.stride + off);
      const float4 d = *reinterpret_cast<const float4*>(p.part + (s + 3) * p.stride + off);
      v.x += (a.x + b.x) + (c.x + d.x); v.y += (a.y + b.y) + (c.y + d.y);
      v.z += (a.z + b.z) + (c.z + d.z); v.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; s < p.ks; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(p.part + s * p.stride + off);
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (p.epilogue == 0) {
      if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      st4_any(p.out, (unsigned long)off, p.out_bf16 != 0, v);
      st_s += (v.x + v.y) + (v.z + v.w);
      st_q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    } else {
      float* gradp = p.dst[0].grad;
      const float *fwd0 = p.dst[0].fwd, *aff0 = p.dst[0].aff, *mask0 = p.dst[0].mask;
      int C = p.dst[0].C, dact = p.dst[0].act, dacc = p.dst[0].accumulate, cst = 0, dfl = p.dst[0].flags, qsel = 0;
#pragma unroll
      for (int q = 1; q < PG_MAX_SRC; ++q)
        if (q < p.ndst && col >= p.dstart[q]) {
          gradp = p.dst[q].grad; fwd0 = p.dst[q].fwd; aff0 = p.dst[q].aff; mask0 = p.dst[q].mask;
          C = p.dst[q].C; dact = p.dst[q].act; dacc = p.dst[q].accumulate; cst = p.dstart[q]; dfl = p.dst[q].flags; qsel = q;
        }
      const int c = col - cst;
      const long idx = pixel * C + c;
      float g4[4] = {v.x, v.y, v.z, v.w};
      float m4[4] = {1.f, 1.f, 1.f, 1.f};
      float f4[4] = {0.f, 0.f, 0.f, 0.f};
      if (mask0) { const float4 m = *reinterpret_cast<const float4*>(mask0 + (long)n * C + c); m4[0] = m.x; m4[1] = m.y; m4[2] = m.z; m4[3] = m.w; }
      if (fwd0) {
        const float4 f = ld4_any(fwd0, (unsigned)idx, (dfl & PG_DST_FWD_BF16) != 0);
        const float a = aff0 ? aff0[2 * n] : 1.f, b = aff0 ? aff0[2 * n + 1] : 0.f;
        f4[0] = f.x; f4[1] = f.y; f4[2] = f.z; f4[3] = f.w;
        const float slope = act_slope(dact);
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = (g4[e] * m4[e]) * act_grad_s(fmaf(f4[e], a, b) * m4[e], slope);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] *= m4[e];
      }
      const bool gbf = (dfl & PG_DST_GRAD_BF16) != 0;
      if (dacc) { const float4 old = ld4_any(gradp, (unsigned)idx, gbf); g4[0] += old.x; g4[1] += old.y; g4[2] += old.z; g4[3] += old.w; }
      st4_any(gradp, (unsigned long)idx, gbf, make_float4(g4[0], g4[1], g4[2], g4[3]));
      if (p.has_bs) {            // the value just stored is FINAL (host: this launch is the tensor's last writer): sums_mode 1
        const float s4 = (g4[0] + g4[1]) + (g4[2] + g4[3]);
        const float q4s = fmaf(g4[0], f4[0], fmaf(g4[1], f4[1], fmaf(g4[2], f4[2], g4[3] * f4[3])));
#pragma unroll
        for (int q = 0; q < PG_MAX_SRC; ++q)
          if (qsel == q) { bs_s[q] += s4; bs_q[q] += q4s; }
      }
    }
  }
  if (p.epilogue == 1 && p.has_bs) {
    __shared__ double redb[4][PG_MAX_SRC][2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < PG_MAX_SRC; ++q) {
      const double ds = wave_sum_d((double)bs_s[q]), dq = wave_sum_d((double)bs_q[q]);
      if (lane == 0) { redb[w][q][0] = ds; redb[w][q][1] = dq; }
    }
    __syncthreads();
    if (threadIdx.x < PG_MAX_SRC) {
      const int q = threadIdx.x;
      double* b = p.dst[0].bsums;
#pragma unroll
      for (int k = 1; k < PG_MAX_SRC; ++k)
        if (q == k) b = p.dst[k].bsums;
      if (q < p.ndst && b != nullptr) {
        const double ds = (redb[0][q][0] + redb[1][q][0]) + (redb[2][q][0] + redb[3][q][0]);
        const double dq = (redb[0][q][1] + redb[1][q][1]) + (redb[2][q][1] + redb[3][q][1]);
        double* slot = b + ((long)n * PG_STAT_SLOTS + (blockIdx.x % PG_STAT_SLOTS)) * 2;
        if (ds != 0.0 || dq != 0.0) { atomicAdd(&slot[0], ds); atomicAdd(&slot[1], dq); }
      }
    }
  }
  if (p.stats != nullptr) {
    __shared__ double red[8];
    const double ds = wave_sum_d((double)st_s), dq = wave_sum_d((double)st_q);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[w] = ds; red[4 + w] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double* slot = p.stats + ((long)n * PG_STAT_SLOTS + (blockIdx.x % PG_STAT_SLOTS)) * 2;
      atomicAdd(&slot[0], red[0] + red[1] + red[2] + red[3]);
      atomicAdd(&slot[1], red[4] + red[5] + red[6] + red[7]);
    }
  }
}

// ------------------------------------------------------------------------------------------- host side
template <int BM, int BN, int WGM, int WGN>
static void launch_cfg(const ConvK& k, int amode, int bmode, int prec, bool dma, bool nomask, dim3 grid, hipStream_t st) {
#define PG_CFG_LAUNCH(A, B) PG_KLAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, A, B>), grid, dim3(256), 0, st, k)
#define PG_LAUNCH_LP(A, B, P) \
  PG_KLAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, A, B, P>), grid, dim3(256), 0, st, k)
  if constexpr (WGN == 2) {
    if (prec == PG_PREC_BF16_DATA) {                                        // bf16 tensors, DMA loaders
      PG_KLAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, A_VEC, B_NT, 3>), grid, dim3(256), 0, st, k);
      return;
    }
    if (dma && prec == PG_PREC_F32 && amode == A_VEC && bmode == B_NN) {     // LDS-DMA loaders (pure operands)
      PG_KLAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, A_VEC, B_NN, 0, 1>), grid, dim3(256), 0, st, k);
      return;
    }
  }
  if constexpr (WGN == 2) {   // low-precision operand modes: the three big vector-loader tiles only
    if (prec != PG_PREC_F32 && amode == A_VEC && bmode != B_SCALAR) {
      if (prec == PG_PREC_BF16 && bmode == B_NT) PG_LAUNCH_LP(A_VEC, B_NT, 1);
      else if (prec == PG_PREC_BF16) PG_LAUNCH_LP(A_VEC, B_NN, 1);
      else if (bmode == B_NT) PG_LAUNCH_LP(A_VEC, B_NT, 2);
      else PG_LAUNCH_LP(A_VEC, B_NN, 2);
      return;
    }
  }
  if (nomask && prec == PG_PREC_F32 && amode == A_VEC && bmode == B_NT)
    PG_KLAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, A_VEC, B_NT, 0, 0, 1>), grid, dim3(256), 0, st, k);
  else if (nomask && prec == PG_PREC_F32 && amode == A_VEC && bmode == B_NN)
    PG_KLAUNCH((conv_igemm_kernel<BM, BN, WGM, WGN, A_VEC, B_NN, 0, 0, 1>), grid, dim3(256), 0, st, k);
  else if (amode == A_VEC && bmode == B_NT) PG_CFG_LAUNCH(A_VEC, B_NT);
  else if (amode == A_VEC && bmode == B_NN) PG_CFG_LAUNCH(A_VEC, B_NN);
  else if (amode == A_VEC && bmode == B_SCALAR) PG_CFG_LAUNCH(A_VEC, B_SCALAR);
  else PG_CFG_LAUNCH(A_SCALAR, B_SCALAR);
#undef PG_CFG_LAUNCH
#undef PG_LAUNCH_LP
}

}  // namespace pg

namespace pg { void launch_conv_bf16_big(const ConvK& k, int bn, dim3 grid, hipStream_t st); }   // igemm_bf16.hip
namespace pg { void launch_conv_bf16_pair(const ConvK& k, int bn, dim3 grid, hipStream_t st); }  // igemm_bf16_pair.hip
namespace pg { void launch_conv_bf16_quad(const ConvK& k, bool merged, int waves, dim3 grid, hipStream_t st); }  // igemm_bf16_quad.hip

using namespace pg;

extern "C" int pg_norm_stats(const float* y, int32_t N, int64_t L, double* sums, void* stream);

namespace {
struct TapBatch { int gtaps; const long* a_off; const long* w_off; const long* o_off; };
}

static int conv_impl(const pg_conv_t* d, const TapBatch* tb, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  PG_REQUIRE(d != nullptr, "pg_conv: null descriptor");
  PG_REQUIRE(d->nsrc >= 1 && d->nsrc <= PG_MAX_SRC, "pg_conv: nsrc=%d", d->nsrc);
  PG_REQUIRE(d->KH * d->KW <= MAXTAP && d->stride >= 1 && d->stride <= 2, "pg_conv: unsupported kernel %dx%d s%d",
             d->KH, d->KW, d->stride);
  PG_REQUIRE(d->precision >= PG_PREC_F32 && d->precision <= PG_PREC_BF16_DATA, "pg_conv: precision=%d", d->precision);
  const bool bf16_data = d->precision == PG_PREC_BF16_DATA;
  const int bke = bf16_data ? 64 : BK;            // K elements per tile
  ConvK k;
  memset(&k, 0, sizeof(k));
  k.nsrc = d->nsrc;
  int ctot = 0;
  for (int j = 0; j < d->nsrc; ++j) {
    k.src[j] = d->src[j];
    k.cstart[j] = ctot;
    ctot += d->src[j].C;
    PG_REQUIRE(d->src[j].ptr != nullptr, "pg_conv: src[%d] null", j);
  }
  for (int j = d->nsrc; j <= PG_MAX_SRC; ++j) k.cstart[j] = ctot;
  k.Ctot = ctot;
  k.N = d->N; k.Hi = d->Hi; k.Wi = d->Wi; k.act = d->act;
  k.Ho = d->Ho; k.Wo = d->Wo;
  k.W = d->W; k.wCout = d->wCout; k.wCin = d->wCin; k.w_transposed = d->w_transposed;
  const int nfull = d->w_transposed ? d->wCin : d->wCout;
  const int kfull = d->w_transposed ? d->wCout : d->wCin;
  PG_REQUIRE(kfull == ctot, "pg_conv: operand channels %d != weight K dim %d", ctot, kfull);
  k.n_off = d->n_off;
  k.n_cnt = d->n_cnt > 0 ? d->n_cnt : nfull;
  if (k.n_cnt % 32 != 0) k.dst_uniform = 0;
  PG_REQUIRE(k.n_off >= 0 && k.n_off + k.n_cnt <= nfull, "pg_conv: bad N sub-range");
  k.epilogue = d->epilogue; k.out_act = d->out_act; k.out = d->out; k.bias = d->bias;
  k.oN = d->oN; k.oC = d->oC; k.oH = d->oH; k.oW = d->oW;
  // ---- tap tables
  if (d->mode == 0) {
    k.nphase = 1; k.so = 1; k.si = d->stride; k.Gy = d->Ho; k.Gx = d->Wo;
    int t = 0;
    for (int r = 0; r < d->KH; ++r)
      for (int s = 0; s < d->KW; ++s) {
        k.dy[0][t] = (signed char)(r - d->pad); k.dx[0][t] = (signed char)(s - d->pad);
        k.wtap[0][t] = (unsigned char)(r * d->KW + s); ++t;
      }
    k.ntap[0] = t;
  } else {
    const int S = d->stride;
    k.nphase = S * S; k.so = S; k.si = 1;
    k.Gy = (d->Ho + S - 1) / S; k.Gx = (d->Wo + S - 1) / S;
    for (int py = 0; py < S; ++py)
      for (int px = 0; px < S; ++px) {
        const int ph = py * S + px;
        k.phy[ph] = py; k.phx[ph] = px;
        int t = 0;
        for (int r = 0; r < d->KH; ++r) {
          if ((((py + d->pad - r) % S) + S) % S != 0) continue;
          for (int s = 0; s < d->KW; ++s) {
            if ((((px + d->pad - s) % S) + S) % S != 0) continue;
            k.dy[ph][t] = (signed char)((py + d->pad - r) / S); k.dx[ph][t] = (signed char)((px + d->pad - s) / S);
            k.wtap[ph][t] = (unsigned char)(r * d->KW + s); ++t;
          }
        }
        k.ntap[ph] = t;
      }
  }
  k.M = d->N * k.Gy * k.Gx;
  PG_REQUIRE(k.M > 0, "pg_conv: empty problem");
  // ---- epilogue destinations
  k.ndst = d->ndst;
  if (d->epilogue == 1) {
    PG_REQUIRE(d->ndst >= 1 && d->ndst <= PG_MAX_SRC, "pg_conv: ndst=%d", d->ndst);
    int c = 0;
    for (int j = 0; j < d->ndst; ++j) { k.dst[j] = d->dst[j]; k.dstart[j] = c; c += d->dst[j].C; }
    for (int j = d->ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
    k.dst_uniform = 1;
    for (int j = 0; j < d->ndst; ++j)
      if (d->dst[j].C % 32 != 0 || (double)d->N * d->Ho * d->Wo * d->dst[j].C >= 4294967296.0) k.dst_uniform = 0;
    k.vec_dst = (k.dst_uniform && !env().no_vec_epilogue) ? 1 : 0;
    for (int j = 0; j < d->ndst; ++j)
      if (((size_t)d->dst[j].grad & 15) != 0 || ((size_t)d->dst[j].fwd & 15) != 0 || ((size_t)d->dst[j].mask & 15) != 0 ||
          ((size_t)d->dst[j].aff & 7) != 0)
        k.vec_dst = 0;
    PG_REQUIRE(c == k.n_cnt, "pg_conv: dst channels %d != N %d", c, k.n_cnt);
  } else {
    PG_REQUIRE(d->out != nullptr, "pg_conv: out null");
  }
  // ---- loader modes
  int amode = d->scalar_in ? A_SCALAR : A_VEC;
  if (amode == A_VEC) {
    for (int j = 0; j < d->nsrc; ++j)
      PG_REQUIRE(d->src[j].C % bke == 0, "pg_conv: vec mode needs C%%%d==0 (src %d has %d)", bke, j, d->src[j].C);
  }
  if (bf16_data) {      // operands are bf16 tensors: pre-materialised activations, K-contiguous bf16 weights
    PG_REQUIRE(amode == A_VEC && !d->w_transposed && d->act == PG_ACT_NONE, "pg_conv: bf16 data path needs vector, "
               "K-contiguous, already-activated operands");
    for (int j = 0; j < d->nsrc; ++j)
      PG_REQUIRE(d->src[j].aff == nullptr && d->src[j].mask == nullptr, "pg_conv: bf16 data path: source %d carries a "
                 "deferred affine / mask (materialise it first)", j);
  }
  int bmode;
  const bool nvec_ok = (k.n_cnt % 4 == 0) && (k.n_off % 4 == 0);
  if (amode == A_SCALAR) bmode = B_SCALAR;
  else if (!d->w_transposed) bmode = B_NT;               // k contiguous: needs wCin%4 (true: Ctot%32==0)
  else bmode = (nvec_ok && d->wCin % 4 == 0) ? B_NN : B_SCALAR;
  if (d->epilogue == 1 && !k.dst_uniform) bmode = B_SCALAR;   // the vector kernels carry the uniform scatter only
  // ---- tile config
  // bf16 STORAGE (output / destination tensors in bf16) is implemented by the row-major epilogues only, and those exist for wave
  // tiles of 64 rows (TM == 2): the 64 x 64 workgroup tile (wave tile 32 x 32) would fall through to the MFMA-layout epilogue and
  // address the bf16 tensors as fp32 (round 5: found when a split-K constant made the time model pick ks = 1 for the 4 x 4 layers at
  // batch 4 — wrong results and a memory fault; the default constants always split those launches, which go through the fix-up pass)
  bool io16 = d->epilogue == 0 && d->out_bf16 != 0;
  if (d->epilogue == 1)
    for (int j = 0; j < d->ndst; ++j) io16 = io16 || d->dst[j].flags != 0;
  int cfg;  // 0: 128x128, 1: 128x64, 2: 64x64, 3: 128x32
  if (k.n_cnt <= 32) cfg = 3;
  else if (k.M <= 64 && !io16) cfg = 2;
  else if (k.n_cnt % 128 == 0) cfg = 0;
  else cfg = 1;
  const int BMs[4] = {128, 128, 64, 128}, BNs[4] = {128, 64, 64, 32};
  const int mt = cdiv(k.M, BMs[cfg]), nt = cdiv(k.n_cnt, BNs[cfg]);
  // ---- split-K
  int ktot_min = 1 << 30;
  for (int ph = 0; ph < k.nphase; ++ph) {
    const int kt = (amode == A_VEC) ? k.ntap[ph] * (ctot / bke) : cdiv((long)k.ntap[ph] * ctot, BK);
    if (kt < ktot_min) ktot_min = kt;
  }
  int ks = d->ksplit;
  if (ks <= 0) {
    // Split K by a small time model (round 2; round 1 maximised the fill of the last round of 512 workgroup slots with a
    // flat 0.2 % cost per split, which picked 13 splits for 784-workgroup launches at 224^2 — 1.3 GB of partial-tile
    // traffic on a 1.7 ms contraction, 67 TFLOP/s instead of 120):
    //   T(c) = rounds(c) * (t_iter * ceil(ktot / c) + t_fixed) + [c > 1] * (2 c out_bytes / BW + t_launch)
    // rounds = ceil(blocks c / 512) (two co-resident workgroups x 256 CUs), t_iter = the K-tile time of one workgroup at
    // the sustained rate of the operand format, t_fixed = prologue + epilogue of a workgroup, the last term = partial
    // tiles written and read back by the fix-up kernel.  tools/sweep_splitk_model.sh sweeps the two constants (PG_SPLITK_* override): flat within 2 % over 4-30 us and 1.5-3 TB/s;
    // configs[2] (224^2, P = 32, batch 8): fp32 167 -> 210 img/s, bf16 data path 446 -> 681 img/s against the round-1 rule.
    static const double t_fixed = getenv("PG_SPLITK_FIXED_US") ? atof(getenv("PG_SPLITK_FIXED_US")) : 12.0;
    static const double bw_tbs = getenv("PG_SPLITK_BW_TBS") ? atof(getenv("PG_SPLITK_BW_TBS")) : 3.0;
    static const double t_launch = getenv("PG_SPLITK_LAUNCH_US") ? atof(getenv("PG_SPLITK_LAUNCH_US")) : 6.0;
    const double rate_tf = d->precision == PG_PREC_F32 ? 125.0 : (d->precision == PG_PREC_BF16X3 ? 250.0 : 650.0);
    const double t_iter = 2.0 * BMs[cfg] * BNs[cfg] * ((amode == A_VEC) ? bke : BK) / (rate_tf * 1e6 / 512.0);      // us
    const long blocks = (long)mt * nt * (tb ? tb->gtaps : k.nphase);
    const double out_bytes = (double)d->N * d->Ho * d->Wo * k.n_cnt * 4.0;
    const int kmax = ktot_min / 4 > 0 ? ktot_min / 4 : 1;
    ks = 1;
    double best = 1e30;
    for (int c = 1; c <= 32 && c <= kmax; ++c) {
      const long b = blocks * c;
      const double rounds = (double)((b + 511) / 512);
      double t = rounds * (t_iter * ((ktot_min + c - 1) / c) + t_fixed);
      if (c > 1) t += 2.0 * c * out_bytes / (bw_tbs * 1e6) + t_launch;
      if (t < best * 0.98) { best = t; ks = c; }             // a further split must buy 2 %
    }
    if (env().splitk_debug) fprintf(stderr, "[splitk] M=%d N=%d ktot=%d blocks=%ld cfg=%d -> ks=%d (T=%.0f us)\n", k.M, k.n_cnt, ktot_min, blocks, cfg, ks, best);
  }
  if (d->out_act != PG_OUT_NONE) ks = 1;
  if (d->epilogue == 0) {   // split-K accumulates atomically into a zeroed, dense NHWC output only
    const bool dense = d->oC == 1 && d->oW == (long)k.n_cnt && d->oH == (long)d->Wo * k.n_cnt &&
                       d->oN == (long)d->Ho * d->Wo * k.n_cnt;
    if (!dense) ks = 1;
  }
  if (ks < 1) ks = 1;
  // the workspace path needs every split of every phase to own at least one K tile (an empty split returns before it
  // stores its partial tile); the kernel gives split s the tiles [s*ceil(kt/ks), (s+1)*ceil(kt/ks))
  auto splits_nonempty = [&](int c) {
    for (int ph = 0; ph < k.nphase; ++ph) {
      const int kt = (amode == A_VEC) ? k.ntap[ph] * (ctot / bke) : cdiv((long)k.ntap[ph] * ctot, BK);
      if ((long)(c - 1) * cdiv(kt, c) >= kt) return false;
    }
    return true;
  };
  if (d->ksplit <= 0 && d->workspace != nullptr)
    while (ks > 1 && !splits_nonempty(ks)) --ks;
  // split-K through the caller's workspace (plain partial stores + splitk_fixup_kernel, summed in split order) when it is big enough
  const double out_elems = (double)d->N * d->Ho * d->Wo * k.n_cnt;
  auto part_ok = [&](int c) {
    if (!(c > 1 && d->workspace != nullptr && tb == nullptr && k.n_cnt % 4 == 0 && splits_nonempty(c) &&
          ((size_t)d->workspace & 15) == 0 && out_elems < 2147483648.0 && (double)c * out_elems * 4.0 <= (double)d->workspace_bytes &&
          !env().no_splitk_ws))
      return false;
    if (d->epilogue == 0) {
      if (((size_t)d->out & 15) != 0 || (d->bias && ((size_t)d->bias & 15) != 0)) return false;
    } else {
      for (int j = 0; j < d->ndst; ++j)
        if (d->dst[j].C % 4 != 0 || ((size_t)d->dst[j].grad & 15) != 0 || ((size_t)d->dst[j].fwd & 15) != 0 ||
            ((size_t)d->dst[j].mask & 15) != 0)
          return false;
    }
    return true;
  };
  // PG_DETERMINISTIC: a split launch that cannot take the workspace path would add its splits with float atomics (arrival
  // order = rounding order) — it runs un-split instead
  if (deterministic() && ks > 1 && !part_ok(ks)) ks = 1;
  k.ksplit = ks;
  k.stats = nullptr;
  if (d->stats != nullptr) {
    PG_REQUIRE(d->epilogue == 0 && d->out_act == PG_OUT_NONE && d->oC == 1 && d->oW == (long)k.n_cnt &&
               d->oH == (long)d->Wo * k.n_cnt && d->oN == (long)d->Ho * d->Wo * k.n_cnt,
               "pg_conv: fused statistics need a dense NHWC output without output activation");
    if (ks == 1) k.stats = d->stats;
  }
  const bool use_part = part_ok(ks);
  {
    const bool dense0 = d->epilogue == 0 && d->oC == 1 && d->oW == (long)k.n_cnt && d->oH == (long)d->Wo * k.n_cnt &&
                        d->oN == (long)d->Ho * d->Wo * k.n_cnt && ((size_t)d->out & 15) == 0 &&
                        (d->bias == nullptr || ((size_t)d->bias & 15) == 0) && tb == nullptr;
    k.vec_out = (k.n_cnt % 4 == 0 && out_elems < 2147483648.0 && (use_part || dense0) && !env().no_vec_epilogue) ? 1 : 0;
  }
  k.part = use_part ? reinterpret_cast<float*>(d->workspace) : nullptr;
  k.part_stride = (long)out_elems;
  {   // bf16 STORAGE of the output / the gradient destinations: only the row-major 8-byte epilogues implement it
    bool io_bf16 = d->epilogue == 0 && d->out_bf16 != 0;
    if (d->epilogue == 1)
      for (int j = 0; j < d->ndst; ++j) io_bf16 = io_bf16 || d->dst[j].flags != 0;
    k.out_bf16 = (d->epilogue == 0 && d->out_bf16 != 0) ? 1 : 0;
    k.dst_io = 0;
    if (d->epilogue == 1) {
      bool all = true, none = true;
      for (int j = 0; j < d->ndst; ++j) {
        const int want = PG_DST_GRAD_BF16 | (d->dst[j].fwd ? PG_DST_FWD_BF16 : 0);
        if ((d->dst[j].flags & want) != want || (d->dst[j].flags & ~want)) all = false;
        if (d->dst[j].flags != 0) none = false;
      }
      k.dst_io = none ? 0 : (all ? 1 : 2);
    }
    if (io_bf16) {
      PG_REQUIRE(bf16_data && tb == nullptr && cfg != 3 && d->out_act == PG_OUT_NONE && (ks == 1 || use_part) &&
                 (d->epilogue == 0 ? k.vec_out != 0 : k.vec_dst != 0),
                 "pg_conv: bf16 storage needs the bf16 data path, a dense 16-byte aligned NHWC output / destinations with "
                 "C %% 32 == 0, no output activation and (for split-K launches) the workspace (cfg %d ks %d)", cfg, ks);
    }
  }
  if (ks > 1 && !use_part) {   // atomic accumulation needs zero-initialised destinations
    if (d->epilogue == 0) {
      PG_REQUIRE(d->oC == 1 && d->oW == (long)k.n_cnt && d->oH == (long)d->Wo * k.n_cnt &&
                 d->oN == (long)d->Ho * d->Wo * k.n_cnt,
                 "pg_conv: split-K needs a dense NHWC output");
      PG_MEMSET_ASYNC(d->out, 0, sizeof(float) * (size_t)d->N * d->Ho * d->Wo * k.n_cnt * (tb ? tb->gtaps : 1), st);
    } else {
      for (int j = 0; j < d->ndst; ++j)
        if (!d->dst[j].accumulate)
          PG_MEMSET_ASYNC(d->dst[j].grad, 0, sizeof(float) * (size_t)d->N * d->Ho * d->Wo * d->dst[j].C, st);
    }
  }
  // LDS-DMA loaders: the A operand must need no prologue (one source, no deferred affine / mask / activation)
  const bool dma = amode == A_VEC && bmode == B_NN && d->nsrc == 1 && d->src[0].aff == nullptr && d->src[0].mask == nullptr &&
                   d->act == PG_ACT_NONE && d->precision == PG_PREC_F32 && cfg != 3 && !env().no_dma;
  if (tb) {
    PG_REQUIRE(tb->gtaps >= 1 && tb->gtaps <= MAXTAP && k.nphase == 1 && d->precision == PG_PREC_BF16_DATA && d->epilogue == 0,
               "batched-tap GEMM: bf16 data path, one phase, plain epilogue");
    k.gtaps = tb->gtaps;
    for (int t = 0; t < tb->gtaps; ++t) { k.a_off[t] = tb->a_off[t]; k.w_off[t] = tb->w_off[t]; k.o_off[t] = tb->o_off[t]; }
  }
  k.xcd_swizzle = (bf16_data && mt % 8 == 0 && nt > 1 && !env().no_xcd_swizzle) ? 1 : 0;
  bool nomask = !env().conv_mask_generic;
  for (int j = 0; j < d->nsrc; ++j) nomask = nomask && d->src[j].mask == nullptr;
  dim3 grid(mt, nt, (tb ? tb->gtaps : k.nphase) * ks);
  // 256-row bf16 kernel (igemm_bf16.hip) for launches with enough 256 x BN tiles to fill the chip (one workgroup per CU)
  if (bf16_data && tb == nullptr && ks == 1 && amode == A_VEC && bmode == B_NT && d->out_act == PG_OUT_NONE &&
      ((d->epilogue == 0 && k.vec_out) || (d->epilogue == 1 && k.vec_dst)) && !env().no_bf16_big) {
    // (n_cnt == 32: the output convolution's 27 -> 32 tap columns on the 512 x 64 tile, half of its columns masked: half the fp32
    //  bytes of the 64-column padding; plain epilogue without statistics only)
    const bool n32 = k.n_cnt == 32 && d->epilogue == 0 && d->stats == nullptr && k.nphase == 1;
    const int bn = (k.n_cnt % 256 == 0) ? 256 : (k.n_cnt % 128 == 0 ? 128 : ((k.n_cnt == 64 || n32) && !env().no_bf16_big64 ? 64 : 0));
    if (bn != 0) {
      int mtb = cdiv(k.M, bn == 64 ? 512 : 256);
      const int ntb = cdiv(k.n_cnt, bn);
      long wgs = (long)mtb * ntb * k.nphase;
      static const long big_min = getenv("PG_BF16_BIG_MIN") ? atol(getenv("PG_BF16_BIG_MIN")) : 96;   // swept (tools/sweep_bf16_big_min.sh): 448 -> 523 / 814, 192 -> 529 / 832 img/s (256^2 batch 4 / 224^2 batch 8); round 5, batch 4: 192 -> 643 / 647, 128 -> 621 (of 618), 96 -> 655.5 / 657.6, 64 -> 653; batch 32 unchanged
      // (round 4) 128-column tiles: 512 x 128 x 32 in three stages instead of 256 x 128 x 64 when the launch still fills the
      // chip with 512-row tiles (csrc/igemm_bf16.hip)
      int code = bn;
      if (bn == 128) {
        // PG_BIG_128_VARIANT = 256 | 512 pins the tile (read per launch: the test-suite flips it inside one process)
        const char* pin = getenv("PG_BIG_128_VARIANT");
        const int mt5 = cdiv(k.M, 512);
        const bool want = pin ? (pin[0] == '5' && k.M >= 512) : ((long)mt5 * ntb * k.nphase >= 2 * big_min);
        if (want) { code = 129; mtb = mt5; wgs = (long)mt5 * ntb * k.nphase; }
      }
      if (wgs >= big_min || getenv("PG_FORCE_BF16_BIG") != nullptr) {
        // (round 4) tap-pair sharing of the A tile (igemm_bf16_pair.hip): every phase's taps must pair up as (dy, dx), (dy, dx + si)
        // — reordered here so that taps 2g, 2g + 1 are pair g — and the tile's extra LDS rows (one per image row) must suffice
        bool pair = false, pair_taps = false;      // pair_taps: every phase's taps pair up (kp = the launch with pair-ordered taps)
        ConvK kp = k;
        {
          const char* pe = getenv("PG_BIG_PAIR");          // "0" / "1": read per launch (the test-suite flips it inside one process)
          const int bm_p = bn == 64 ? 512 : 256, axr = bn == 64 ? 8 : 16;
          bool want = pe ? pe[0] != '0' : true;
          want = want && k.Gx >= 2 && k.Wi * 1 < 32000 && k.Hi < 32000;
          const bool rows_ok = (bm_p + k.Gx - 2) / k.Gx + 1 <= axr;      // (the x-phase merged form below has its own row condition)
          if (want) {
            ConvK kk = k;
            bool ok = true;
            for (int ph = 0; ph < k.nphase && ok; ++ph) {
              const int nt_ = k.ntap[ph];
              if (nt_ < 2 || (nt_ & 1)) { ok = false; break; }
              int o[MAXTAP], used[MAXTAP];
              for (int a = 0; a < nt_; ++a) { o[a] = a; used[a] = 0; }
              for (int a = 1; a < nt_; ++a)           // insertion sort by (dy, dx)
                for (int b = a; b > 0; --b) {
                  const int x = o[b - 1], y = o[b];
                  if (k.dy[ph][x] > k.dy[ph][y] || (k.dy[ph][x] == k.dy[ph][y] && k.dx[ph][x] > k.dx[ph][y])) { o[b - 1] = y; o[b] = x; }
                }
              int w = 0;
              for (int a = 0; a < nt_ && ok; ++a) {
                if (used[a]) continue;
                int b = a + 1;
                while (b < nt_ && (used[b] || k.dy[ph][o[b]] != k.dy[ph][o[a]] || k.dx[ph][o[b]] != k.dx[ph][o[a]] + k.si)) ++b;
                if (b >= nt_) { ok = false; break; }
                used[a] = used[b] = 1;
                const int pr[2] = {o[a], o[b]};
                for (int e = 0; e < 2; ++e) {
                  kk.dy[ph][w] = k.dy[ph][pr[e]]; kk.dx[ph][w] = k.dx[ph][pr[e]]; kk.wtap[ph][w] = k.wtap[ph][pr[e]];
                  ++w;
                }
              }
            }
            if (ok) { kp = kk; pair_taps = true; pair = rows_ok; }
            if (pair) k = kk;
          }
        }
        if (pair && code == 129) { code = 128; mtb = cdiv(k.M, 256); }          // 128 columns: the paired 256 x 128 tile instead of 512 x 128 x 32
        k.xcd_swizzle = (mtb % 8 == 0 && (ntb > 1 || k.nphase > 1) && !env().no_xcd_swizzle) ? 1 : 0;
        k.xcd_swizzle |= (int)env().debug_bits;       // zero unless built with -DPG_TIMING_EXPERIMENTS
        if (k.dst_io == 1 && k.Gy * k.Gx < 32) k.dst_io = 2;      // the pipelined bf16 scatter assumes <= 2 samples per 32 rows
        // (round 5) x-phase merging (igemm_bf16_pair.hip, MG): a transposed k4 s2 convolution with N = 128 / 64 output columns runs
        // its phases (py, 0) and (py, 1) in ONE workgroup tile of 2 N columns — the A tile carries Gx + 2 slots per image row, the B
        // tile both phases' weights; the outputs of the two phases are neighbouring pixels, so the epilogues see an image of half
        // the width with 2 N channels (n_cnt, the destination's C and its column start doubled; row table = pixel-pair index).
        int bn_l = bn;             // the launch's tile width
        bool merged = false;
        ConvK kmerged;
        {
          const char* me = getenv("PG_BIG_MERGE");          // "0" / "1": read per launch (the test-suite flips it inside one process)
          const ConvK& k = kp;                               // (the pair-ordered tap tables, whether or not the unmerged launch pairs)
          bool want = pair_taps && (me ? me[0] != '0' : true) && d->mode == 1 && k.nphase == 4 && k.so == 2 && (bn == 128 || bn == 64) &&
                      ntb == 1 && k.n_cnt == bn && k.n_off == 0 && k.n_cnt == nfull && d->bias == nullptr && d->Wo == 2 * k.Gx &&
                      d->Wo % 2 == 0 && k.Gx >= 43 && ((long)cdiv(k.M, 256) * 2 >= big_min || getenv("PG_FORCE_BF16_BIG") != nullptr);
          if (want && d->epilogue == 1)
            want = d->ndst == 1 && d->dst[0].mask == nullptr && d->dst[0].C == k.n_cnt && k.dst_io == 1;
          if (want && d->epilogue == 0) want = k.vec_out != 0;
          for (int py = 0; py < 2 && want; ++py) {           // phase (py, 1)'s pair-ordered taps = phase (py, 0)'s one step to the right
            const int a = 2 * py, b = 2 * py + 1;
            if (k.ntap[a] != k.ntap[b] || k.phy[a] != py || k.phy[b] != py || k.phx[a] != 0 || k.phx[b] != 1) { want = false; break; }
            for (int t = 0; t < k.ntap[a]; ++t)
              if (k.dy[a][t] != k.dy[b][t] || k.dx[b][t] != k.dx[a][t] + 1) want = false;
          }
          if (want) {
            ConvK km = k;
            for (int py = 0; py < 2; ++py) {
              const int srcs[2] = {2 * py, 2 * py + 1}, dsts_[2] = {py, 2 + py};
              for (int h = 0; h < 2; ++h) {
                km.ntap[dsts_[h]] = k.ntap[srcs[h]];
                for (int t = 0; t < MAXTAP; ++t) {
                  km.dy[dsts_[h]][t] = k.dy[srcs[h]][t]; km.dx[dsts_[h]][t] = k.dx[srcs[h]][t]; km.wtap[dsts_[h]][t] = k.wtap[srcs[h]][t];
                }
              }
              km.phy[py] = py; km.phx[py] = 0;
            }
            km.nphase = 2;
            km.n_cnt = 2 * k.n_cnt;
            if (d->epilogue == 1) {
              km.dst[0].C = 2 * k.dst[0].C;
              for (int j = 1; j <= PG_MAX_SRC; ++j) km.dstart[j] = km.n_cnt;
            }
            mtb = cdiv(k.M, 256);
            km.xcd_swizzle = ((mtb % 8 == 0 && !env().no_xcd_swizzle) ? 1 : 0) | (int)env().debug_bits;
            kmerged = km;
            merged = true;
            bn_l = 2 * bn;
          }
        }
        // (round 4) fused norm-backward sums (pg_dst_t.bsums): the pipelined bf16 scatter implements them when a workgroup's
        // column tile lies inside one destination; otherwise the field is dropped and PG_INFO_BSUMS stays clear
        if (merged) k = kmerged;
        bool bs_any = false, bs_ok = d->epilogue == 1 && k.dst_io == 1 && k.n_cnt % bn_l == 0;
        if (d->epilogue == 1)
          for (int j = 0; j < d->ndst; ++j) {
            if (d->dst[j].bsums == nullptr) continue;
            bs_any = true;       // its column range must be tile-aligned: no workgroup mixes it with another destination
            if (k.dstart[j] % bn_l != 0 || k.dst[j].C % bn_l != 0 || d->dst[j].fwd == nullptr) bs_ok = false;
          }
        if (!(bs_any && bs_ok))
          for (int j = 0; j < PG_MAX_SRC; ++j) k.dst[j].bsums = nullptr;
        // (round 6) tap-QUAD sharing on 512 x 128 x 32 tiles (igemm_bf16_quad.hip) for the 128-column launches whose taps form
        // 2 x 2 quads (k4 s2 p1 forward: 4 quads; a transposed phase / an x-merged phase pair: 1 quad): every input pixel of the tile's
        // halo patch goes global -> LDS once, the weight tile once per 512 rows.  Conditions: the 512-row tile lies inside one sample,
        // one image row of extra LDS rows fits (Gx <= ~128), enough tiles to fill the chip.
        bool quad = false;
        // 512: 8 waves, one workgroup per CU (the default); 256: 4 waves, two per CU.  Alone on the chip the 4-wave form is the faster one
        // (enc.1 forward 172 against 181 us: one workgroup's prologue / epilogue beside the other's K loop), but inside the training pass,
        // next to the weight-gradient / encoder streams' one-workgroup-per-CU kernels, it LOSES: north-star pass + 0.10 ms against - 0.07 ms
        // for the 8-wave form, alternating arms in one process (tools/quad_inproc_ab.py, profiles/round6_quad_inproc_ab.txt).
        int quad_bm = 512;
        {
          const char* qe = getenv("PG_BIG_QUAD");           // "0" / "1": read per launch (the test-suite flips it inside one process)
          const char* qw = getenv("PG_QUAD_WAVES");         // "8" / "4": likewise
          // >= 4 rounds of 512-row workgroups: with 512 tiles (two rounds) the discriminator's 128 -> 256 data gradient at batch 64 ran
          // 154 us here against 134 us on the tap-pair kernel's 1024 tiles (profiles/round6_launch_table_b32_bf16_data.txt, first version)
          static const long quad_min = getenv("PG_QUAD_MIN") ? atol(getenv("PG_QUAD_MIN")) : 1024;
          quad_bm = (qw && qw[0] == '4') ? 256 : 512;
          const int xs = merged ? 2 : 1, gg_ = k.Gy * k.Gx;
          bool want = (qe ? qe[0] != '0' : true) && (merged ? bn_l == 128 : (bn == 128 && ntb == 1 && k.n_cnt == 128)) &&
                      gg_ % quad_bm == 0 && k.M % quad_bm == 0 && ctot % 32 == 0 && k.Wi < 32000 && k.Hi < 32000 &&
                      ((long)(k.M / 512) * k.nphase >= quad_min || getenv("PG_FORCE_BF16_BIG") != nullptr);
          if (want) {
            const int max_rho = quad_bm - 1 + xs * ((k.Gx - 1 + quad_bm - 1) / k.Gx) + (merged ? 2 : 1) + k.Gx + xs;
            want = max_rho <= (quad_bm == 512 ? 671 : 399);
          }
          for (int j = 0; j < d->nsrc && want; ++j)
            if ((double)d->N * d->Hi * d->Wi * d->src[j].C * 2.0 >= 4294967296.0 || k.cstart[j] % 32 != 0) want = false;
          if (want) {
            ConvK kk = k;
            bool ok = true;
            for (int ph = 0; ph < (merged ? 4 : k.nphase) && ok; ++ph) {      // (merged: slots py and 2 + py hold the two column halves' taps)
              const int nt_ = k.ntap[ph];
              if (nt_ < 4 || (nt_ & 3)) { ok = false; break; }
              int o[MAXTAP], used[MAXTAP];
              for (int a = 0; a < nt_; ++a) { o[a] = a; used[a] = 0; }
              for (int a = 1; a < nt_; ++a)           // insertion sort by (dy, dx)
                for (int b = a; b > 0; --b) {
                  const int x = o[b - 1], y = o[b];
                  if (k.dy[ph][x] > k.dy[ph][y] || (k.dy[ph][x] == k.dy[ph][y] && k.dx[ph][x] > k.dx[ph][y])) { o[b - 1] = y; o[b] = x; }
                }
              auto find = [&](int dy_, int dx_) {
                for (int b = 0; b < nt_; ++b)
                  if (!used[b] && k.dy[ph][o[b]] == dy_ && k.dx[ph][o[b]] == dx_) return b;
                return -1;
              };
              int w = 0;
              for (int a = 0; a < nt_ && ok; ++a) {
                if (used[a]) continue;
                const int dy0 = k.dy[ph][o[a]], dx0 = k.dx[ph][o[a]];
                int q4[4];
                for (int e = 0; e < 4 && ok; ++e) {
                  q4[e] = find(dy0 + (e >> 1) * k.si, dx0 + (e & 1) * k.si);
                  if (q4[e] < 0) ok = false; else used[q4[e]] = 1;
                }
                for (int e = 0; e < 4 && ok; ++e) {
                  kk.dy[ph][w] = k.dy[ph][o[q4[e]]]; kk.dx[ph][w] = k.dx[ph][o[q4[e]]]; kk.wtap[ph][w] = k.wtap[ph][o[q4[e]]];
                  ++w;
                }
              }
            }
            if (ok && merged)       // the right half's quad = the left half's one slot to the right (the kernel shares the A tile)
              for (int py = 0; py < 2; ++py)
                for (int t = 0; t < 4; ++t)
                  if (kk.dy[2 + py][t] != kk.dy[py][t] || kk.dx[2 + py][t] != kk.dx[py][t] + 1) ok = false;
            if (ok) { k = kk; quad = true; }
          }
        }
        if (quad) {
          const int mtq = k.M / quad_bm;
          k.xcd_swizzle = ((mtq % 8 == 0 && k.nphase > 1 && !env().no_xcd_swizzle) ? 1 : 0) | (int)env().debug_bits;
          launch_conv_bf16_quad(k, merged, quad_bm == 512 ? 8 : 4, dim3(mtq, 1, k.nphase), st);
        }
        else if (merged) launch_conv_bf16_pair(k, 1000 + bn_l, dim3(mtb, 1, 2), st);
        else if (pair) launch_conv_bf16_pair(k, bn, dim3(mtb, ntb, k.nphase), st);
        else launch_conv_bf16_big(k, code, dim3(mtb, ntb, k.nphase), st);
        PG_LAUNCH_OK("pg_conv (bf16 256-row kernel)");
        last_info() = (quad ? (merged ? 14 : 13) : merged ? (bn_l == 256 ? 11 : 12) : pair ? (bn == 256 ? 8 : (bn == 128 ? 9 : 10)) : (code == 129 ? 7 : (bn == 256 ? 4 : (bn == 128 ? 5 : 6)))) |
                      (amode << 4) | (bmode << 8) | (1 << 16) | ((bs_any && bs_ok) ? PG_INFO_BSUMS : 0);
        return 0;
      }
    }
  }
  // the 128 x 32 tile has no bf16-data instantiation: its launcher would fall through to the fp32 kernel and read the bf16
  // operands as floats.  32-column launches on this path must have been taken by the 512 x 64 kernel above.
  PG_REQUIRE(!(bf16_data && cfg == 3), "pg_conv: bf16 data path: a launch with %d output columns was not eligible for the 512x64 "
             "kernel (plain epilogue, no statistics, >= PG_BF16_BIG_MIN workgroups) and has no other bf16 kernel", k.n_cnt);
  // (round 4) fused norm-backward sums (pg_dst_t.bsums) on this path: the split-K fix-up pass implements them for any layout; the
  // in-kernel row-major scatter when a workgroup's column tile lies inside one destination, a sample has >= 64 rows per phase
  // (a wave's 64 rows then hold at most two samples) and the tensors are all fp32 or all bf16.  Otherwise the field is dropped
  // and PG_INFO_BSUMS stays clear: the caller runs pg_norm_bwd_reduce.
  bool gbs = false;
  if (d->epilogue == 1) {
    const int bn_g = cfg == 0 ? 128 : 64;
    bool any = false;
    bool ok = use_part || (ks == 1 && k.vec_dst && (cfg == 0 || cfg == 1) && bmode != B_SCALAR && k.dst_io != 2 &&
                           k.Gy * k.Gx >= 64 && k.n_cnt % bn_g == 0);
    for (int j = 0; j < d->ndst; ++j) {
      if (d->dst[j].bsums == nullptr) continue;
      any = true;
      if (d->dst[j].fwd == nullptr) ok = false;
      if (!use_part && (k.dstart[j] % bn_g != 0 || d->dst[j].C % bn_g != 0)) ok = false;
    }
    gbs = any && ok;
    if (!gbs)
      for (int j = 0; j < PG_MAX_SRC; ++j) k.dst[j].bsums = nullptr;
  }
  switch (cfg) {
    case 0: launch_cfg<128, 128, 2, 2>(k, amode, bmode, d->precision, dma, nomask, grid, st); break;
    case 1: launch_cfg<128, 64, 2, 2>(k, amode, bmode, d->precision, dma, nomask, grid, st); break;
    case 2: launch_cfg<64, 64, 2, 2>(k, amode, bmode, d->precision, dma, nomask, grid, st); break;
    default: launch_cfg<128, 32, 4, 1>(k, amode, bmode, d->precision, dma, nomask, grid, st); break;
  }
  PG_LAUNCH_OK("pg_conv");
  last_info() = cfg | (amode << 4) | (bmode << 8) | (ks << 16) | (dma ? (1 << 12) : 0) | (use_part ? (1 << 13) : 0) |
                (gbs ? PG_INFO_BSUMS : 0);
  if (use_part) {
    FixupK f;
    memset(&f, 0, sizeof(f));
    f.part = k.part; f.stride = k.part_stride; f.ks = ks;
    f.ppix = d->Ho * d->Wo; f.n_cnt = k.n_cnt; f.epilogue = d->epilogue;
    f.bias = d->bias; f.out = d->out; f.stats = (d->epilogue == 0) ? d->stats : nullptr;
    f.out_bf16 = k.out_bf16;
    for (int j = 0; j < PG_MAX_SRC; ++j) f.dst[j] = k.dst[j];
    f.ndst = k.ndst;
    f.has_bs = gbs ? 1 : 0;
    for (int j = 0; j <= PG_MAX_SRC; ++j) f.dstart[j] = k.dstart[j];
    const long items = (long)f.ppix * (k.n_cnt / 4);
    long bx = (items + 511) / 512;
    if (bx < 1) bx = 1;
    if (bx > 128) bx = 128;
    PG_KLAUNCH(splitk_fixup_kernel, dim3((unsigned)bx, (unsigned)d->N), dim3(256), 0, st, f);
    PG_LAUNCH_OK("pg_conv (split-K fixup)");
    return 0;
  }
  if (d->stats != nullptr && k.stats == nullptr)      // split-K (or scatter) launch: statistics from the stored tensor
    return pg_norm_stats(d->out, d->N, (int64_t)d->Ho * d->Wo * k.n_cnt, d->stats, stream);
  return 0;
}

extern "C" int pg_conv(const pg_conv_t* d, void* stream) { return conv_impl(d, nullptr, stream); }

// Batched NT GEMM on the bf16 data path: out[t][m][n] = sum_k A[m][a_off[t] + k] * B[n][b_off[t] + k], t < gtaps,
// A [M][K] and B [N][K] row-major bf16 with row pitch K (K % 64 == 0), offsets in ELEMENTS (may be negative / odd:
// the DMA loaders take 2-byte aligned sources).  The bf16 weight gradient is this product with A = the channel-major
// gradient, B = the channel-major activated input (or its stride-2 phase planes) and one offset per filter tap.
extern "C" int pg_gemm_taps_bf16(const void* A, const void* B, int32_t M, int32_t N, int32_t K, int32_t gtaps,
                                 const int64_t* a_off, const int64_t* b_off, float* out, void* stream) {
  PG_REQUIRE(A && B && out && M > 0 && N > 32 && K > 0 && K % 64 == 0 && gtaps >= 1 && gtaps <= MAXTAP && a_off && b_off,
             "pg_gemm_taps_bf16: bad arguments (K %% 64 == 0, N > 32, gtaps <= 16)");
  pg_conv_t d;
  memset(&d, 0, sizeof(d));
  d.nsrc = 1;
  d.src[0].ptr = reinterpret_cast<const float*>(A);
  d.src[0].C = K;
  d.N = 1; d.Hi = 1; d.Wi = M; d.Ho = 1; d.Wo = M;
  d.act = PG_ACT_NONE; d.mode = 0; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0;
  d.W = reinterpret_cast<const float*>(B); d.wCout = N; d.wCin = K;
  d.epilogue = 0; d.out_act = PG_OUT_NONE; d.out = out;
  d.oN = (int64_t)M * N; d.oC = 1; d.oH = (int64_t)M * N; d.oW = N;
  d.precision = PG_PREC_BF16_DATA;
  long ao[MAXTAP], wo[MAXTAP], oo[MAXTAP];
  for (int t = 0; t < gtaps; ++t) { ao[t] = a_off[t] * 2; wo[t] = b_off[t] * 2; oo[t] = (long)t * M * N; }
  TapBatch tb{gtaps, ao, wo, oo};
  return conv_impl(&d, &tb, stream);
}
