// Weight gradient of the k4 / s2 / p1 Block convolutions on the bf16 data path, straight from the PIXEL-MAJOR (NHWC) bf16
// tensors the forward and data-gradient contractions already use — no channel-major copies (round 1: eight
// `channel_major_bf16` passes per layer, 3.5 ms of a 39 ms generator forward+backward at batch 32, plus a product buffer
// and a torch add).
//
//   dW[tap = (r, s)][co][ci] += sum over small-grid pixels q = (n, qy, qx) of  dY[.][co] * X[.][ci]
//   where one operand lives on the small grid (row q) and the other on the large grid (row (n, 2 qy + r - 1, 2 qx + s - 1),
//   zero outside): Conv2d(k4,s2,p1): X large / dY small; ConvTranspose2d(k4,s2)+crop: X small / dY large
//   (reference models/networks.py:154-157; autograd of conv2d / conv_transpose2d weights).
//
// GEMM view per tap: M = Cout, N = Cin (one K-source of the virtual concat per launch), K = pixels.  Both operands are
// K-major in memory ([pixel][channel]), the MFMA wants 8 consecutive k per lane: the tiles are DMA'd pixel-major into LDS
// (global_load_lds_dwordx4, 16-byte channel chunks, fully coalesced) and the fragments are fetched with
// ds_read_b64_tr_b16 — the gfx950 transposing LDS read: inside a 16-lane group lane 4 r + c supplies the address of 4
// channels (8 B) of pixel r and receives channel (4 c' + ..) of pixels 0..3, i.e. a 4 x 16 block comes back transposed
// (semantics probed on the box: tools/microbench/tr_probe.hip).  Two such reads = one 32x32x16 operand fragment.
// LDS image: [64 pixels][BW channels], chunk c of pixel p stored at slot c ^ ((p & 3) << 2): the four pixel rows a
// 32-lane half reads then cover the 256-byte bank row exactly once (un-swizzled they alias 4-way: the row pitch is a
// multiple of 256 B).  Pipeline = igemm_bf16.hip: 8 waves, two 64 KB stages, ONE barrier per K tile placed before the last
// k-step's MFMAs, counted lgkmcnt, operand registers double-buffered per k-step.
// Work split: grid (M tiles, N tiles, 16 taps x ksplit); partial sums are added to dW with float atomics (ksplit > 1) or
// a plain read-modify-write.
#include <cstdlib>
#include <type_traits>

#include "igemm_common.h"

namespace pg {

struct WgBf16K {
  const unsigned short* sm;   // small-grid operand [N][Hs][Ws][Cs]
  const unsigned short* lg;   // large-grid operand [N][Hl][Wl][Cl]
  int Cs, Cl;
  int N, Hs, Ws, Hl, Wl;
  int a_is_small;             // 1: A (rows = Cout) is the small-grid tensor (Conv2d), 0: the large-grid tensor (ConvT)
  float* dW;                  // [16][Cout][ldw] fp32, accumulated
  int Cout, ldw, col_off;
  int ksplit, atomic;
  int xcd_remap;              // the 16 taps of one (M tile, N tile, K split) run back-to-back on ONE XCD (shared L2)
  long Q;                     // N * Hs * Ws
};

template <int OFF>
__device__ __forceinline__ void lds_tr64(unsigned long long& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

// LDS chunk swizzle of a pixel row of CPR 16-byte chunks (8 channels each): rows of >= 16 chunks XOR the pixel's low two bits
// into chunk bits 2..3; a 64-channel row has only 8 chunks (128 B = half the bank space): pixels 0 / 2 and 1 / 3 of a
// transposing read would meet in the same banks, so bit 1 of the pixel flips chunk bit 2 (the other 64-byte half).
template <int CPR>
__device__ __forceinline__ int wg_swz(int pixel) { return CPR >= 16 ? ((pixel & 3) << 2) : (((pixel >> 1) & 1) << 2); }

#ifndef PG_WG1_SPLIT_DMA
#define PG_WG1_SPLIT_DMA 1
#endif
template <int BM, int BN, int AS>
__global__ __launch_bounds__(512, 2) void wgrad_bf16_tr_kernel(const WgBf16K p) {
  constexpr int WGN = (BN == 64) ? 2 : (BM == 64 ? 4 : ((BN == 256) ? 4 : (BM == 256 ? 2 : 4)));
  constexpr int WGM = 8 / WGN;
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  static_assert(TM >= 1 && TN >= 1, "tile");
  constexpr int A_ST = 64 * BM * 2, B_ST = 64 * BN * 2, STAGE = A_ST + B_ST;
  constexpr int A_CPR = BM / 8, B_CPR = BN / 8;               // 16-byte chunks per pixel row
  constexpr int A_RPI = 64 / A_CPR, B_RPI = 64 / B_CPR;       // pixel rows per wave DMA instruction
  constexpr int A_PASS = 64 / (8 * A_RPI), B_PASS = 64 / (8 * B_RPI);
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroups are dealt round-robin to the 8 XCDs in dispatch order.  The 16 taps of one (M tile, N tile, K split) read
  // the SAME dY tiles and pixel-shifted copies of the same X tiles: remapped, they are the 16 consecutive workgroups one
  // XCD receives, so 15 of the 16 reads of every tile are hits in that XCD's 4 MB L2 instead of fabric round trips (round 2
  // PMC: the waves of this kernel were parked at vmcnt for a third of their cycles behind L2 misses).
  int bx = blockIdx.x, by = blockIdx.y, tap = blockIdx.z / p.ksplit, split = blockIdx.z - tap * p.ksplit;
  if (p.xcd_remap) {            // host: (gridDim.x * gridDim.y * ksplit) % 8 == 0
    const int mt = (int)gridDim.x, nt = (int)gridDim.y;
    const int L = bx + mt * (by + nt * (int)blockIdx.z);
    const int xcd = L & 7, j = L >> 3;
    tap = j & 15;
    const int u = (j >> 4) * 8 + xcd;          // unit = (bx, by, split)
    bx = u % mt;
    by = (u / mt) % nt;
    split = u / (mt * nt);
  }
  const int tr = tap >> 2, ts = tap & 3;
  const int m0 = bx * BM, n0 = by * BN;
  const int ktot = (int)((p.Q + 63) >> 6);
  const int kper = (ktot + p.ksplit - 1) / p.ksplit;
  const int kt0 = split * kper, kt1 = min(ktot, kt0 + kper);
  if (kt0 >= kt1) return;

  // operand roles are compile-time (AS = 1: A / rows = Cout is the small-grid tensor, i.e. Conv2d)
  const unsigned short* const a_base = AS ? p.sm : p.lg;
  const unsigned short* const b_base = AS ? p.lg : p.sm;
  const int Ca = AS ? p.Cs : p.Cl, Cb = AS ? p.Cl : p.Cs;
  const char* const zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- DMA: thread -> (pixel row within the tile, 16-byte slot); chunk = slot ^ ((pixel & 3) << 2)
  const int a_row = wave * A_RPI + lane / A_CPR, a_slot = lane % A_CPR;
  const int b_row = wave * B_RPI + lane / B_CPR, b_slot = lane % B_CPR;
  const int a_ch = m0 + ((a_slot ^ wg_swz<A_CPR>(a_row)) << 3);        // first channel of the chunk this lane moves
  const int b_ch = n0 + ((b_slot ^ wg_swz<B_CPR>(b_row)) << 3);
  const int hw = p.Hs * p.Ws;

  // Loader state = 32-bit BYTE offsets against wave-uniform bases, advanced by 64 small-grid pixels per K tile with adds
  // and selects only (round 2, first version: divisions and branchy 64-bit pointer code per load — 5000 ISA lines, the
  // address code outweighed the 32 MFMAs of a tile):
  //   small operand: offset += 64 * C * 2;  large operand: pixel (n, y, x) -> + a64x columns, + c64y rows, + d64n samples
  //   with at most one carry each, the byte offset moves by a uniform step plus a uniform correction per carry.
  constexpr int S_PASS = AS ? A_PASS : B_PASS, L_PASS = AS ? B_PASS : A_PASS;
  constexpr int S_RPI = AS ? A_RPI : B_RPI, L_RPI = AS ? B_RPI : A_RPI;
  const int s_row = AS ? a_row : b_row, l_row = AS ? b_row : a_row;
  const int Csm = AS ? Ca : Cb, Clg = AS ? Cb : Ca;
  const char* const s_base = uniform_ptr(reinterpret_cast<const char*>(AS ? a_base : b_base));
  const char* const l_base = uniform_ptr(reinterpret_cast<const char*>(AS ? b_base : a_base));
  const int s_ch = AS ? a_ch : b_ch, l_ch = AS ? b_ch : a_ch;
  const unsigned s_step = 64u * (unsigned)Csm * 2u;
  const unsigned Sx = 2u * (unsigned)Clg * 2u;                       // bytes per small-grid column step on the large grid
  const unsigned Sy = 2u * (unsigned)p.Wl * (unsigned)Clg * 2u;      // ... per small-grid row step
  const unsigned Sn = (unsigned)p.Hl * (unsigned)p.Wl * (unsigned)Clg * 2u;
  const int a64x = 64 % p.Ws, b64 = 64 / p.Ws, c64y = b64 % p.Hs, d64n = b64 / p.Hs;
  const unsigned l_step = (unsigned)a64x * Sx + (unsigned)c64y * Sy + (unsigned)d64n * Sn;
  const unsigned l_cx = Sy - (unsigned)p.Ws * Sx, l_cy = Sn - (unsigned)p.Hs * Sy;      // carry corrections (mod 2^32)
  unsigned s_off[S_PASS], l_off[L_PASS];
  int s_q[S_PASS];                                      // small rows: flattened pixel (validity only)
  int ln[L_PASS], ly0[L_PASS], lx0[L_PASS];             // large rows: sample, row, column of the row's small-grid pixel
  {
    const long q0 = (long)kt0 * 64;
#pragma unroll
    for (int i = 0; i < S_PASS; ++i) {
      const long q = q0 + i * 8 * S_RPI + s_row;
      s_q[i] = (int)q;
      s_off[i] = (unsigned)((q * Csm + s_ch) * 2);
    }
#pragma unroll
    for (int i = 0; i < L_PASS; ++i) {
      const long q = q0 + i * 8 * L_RPI + l_row;
      const int n = (int)(q / hw);
      const int rem = (int)(q - (long)n * hw);
      const int y = rem / p.Ws, x = rem - y * p.Ws;
      ln[i] = n; ly0[i] = y; lx0[i] = x;
      // offset of large pixel (n, 2y + tr - 1, 2x + ts - 1), channel l_ch — may point one row / column outside (masked below)
      l_off[i] = (unsigned)n * Sn + (unsigned)y * Sy + (unsigned)x * Sx +
                 (unsigned)(((long)(tr - 1) * p.Wl + (ts - 1)) * Clg * 2 + (long)l_ch * 2);
    }
  }
  const int Q32 = (int)p.Q;
  // large-grid pixel of (small pixel (y, x), tap (tr, ts)) = (2y + tr - 1, 2x + ts - 1): outside below 0 (tap 0 at y = 0) and
  // from y >= ylim on, ylim = ceil((Hl - tr + 1) / 2) — Hl = 2 Hs for the generator's blocks, 2 Hs + 1 on the odd maps of the
  // discriminator (127 -> 63 -> 31 ...), where tap 3 of the last row is still inside
  const bool edge_y0 = tr == 0, edge_x0 = ts == 0;
  const int ylim = (p.Hl - tr + 2) >> 1, xlim = (p.Wl - ts + 2) >> 1;
  // part: 0 = both operands, 1 = the small-grid rows, 2 = the large-grid rows (PG_WG1_SPLIT_DMA: two halves, see igemm_bf16.hip)
  auto issue = [&](int stage, int kt, int part = 0) {
    (void)kt;
    float* const As = reinterpret_cast<float*>(smem + stage * STAGE);
    float* const Bs = reinterpret_cast<float*>(smem + stage * STAGE + A_ST);
    float* const Ss = AS ? As : Bs;
    float* const Ls = AS ? Bs : As;
    const char* const zp = zero_pg + (lane & 7) * 16;
    if (part != 2) {
#pragma unroll
    for (int i = 0; i < S_PASS; ++i) {
      const bool ok = s_q[i] < Q32;
      const char* src = ok ? s_base + s_off[i] : zp;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), Ss + (i * 8 + wave) * 256, 16, 0, 0);
      s_q[i] += 64; s_off[i] += s_step;
    }
    }
    if (part == 1) return;
#pragma unroll
    for (int i = 0; i < L_PASS; ++i) {
      const bool oob = (ln[i] >= p.N) | (edge_y0 & (ly0[i] == 0)) | (ly0[i] >= ylim) | (edge_x0 & (lx0[i] == 0)) | (lx0[i] >= xlim);
      const char* src = oob ? zp : l_base + l_off[i];
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), Ls + (i * 8 + wave) * 256, 16, 0, 0);
      // advance by 64 small-grid pixels
      int x = lx0[i] + a64x, y = ly0[i] + c64y, n = ln[i] + d64n;
      unsigned off = l_off[i] + l_step;
      const bool cx = x >= p.Ws;
      x -= cx ? p.Ws : 0; y += cx ? 1 : 0; off += cx ? l_cx : 0u;
      const bool cy = y >= p.Hs;
      y -= cy ? p.Hs : 0; n += cy ? 1 : 0; off += cy ? l_cy : 0u;
      lx0[i] = x; ly0[i] = y; ln[i] = n; l_off[i] = off;
    }
  };

  // ---- operand fetch (transposing reads).  lane: i = l & 15 -> pixel sub-row r4 = i >> 2, channel quad cq = i & 3;
  // 16-channel block mb = (l >> 4) & 1; k half kh = l >> 5 (pixels + 8).
  const int wm0 = (wave / WGN) * (TM * 32), wn0 = (wave % WGN) * (TN * 32);
  const int r4 = (lane >> 2) & 3, cq = lane & 3, mb = (lane >> 4) & 1, kh = lane >> 5;
  unsigned fa[TM], fb[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int chunk = ((wm0 + 32 * t) >> 3) + 2 * mb + (cq >> 1);
    fa[t] = lds0 + (unsigned)((8 * kh + r4) * (BM * 2)) + (unsigned)(((chunk ^ wg_swz<A_CPR>(r4)) << 4) + ((cq & 1) << 3));
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int chunk = ((wn0 + 32 * t) >> 3) + 2 * mb + (cq >> 1);
    fb[t] = lds0 + A_ST + (unsigned)((8 * kh + r4) * (BN * 2)) + (unsigned)(((chunk ^ wg_swz<B_CPR>(r4)) << 4) + ((cq & 1) << 3));
  }
  typedef unsigned long long u64;
  struct Frag { u64 lo, hi; };
  auto fetch = [&](int stage, auto ksc, Frag (&va)[TM], Frag (&vb)[TN]) {
    constexpr int KS = decltype(ksc)::value;
    const unsigned so = (unsigned)(stage * STAGE);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      lds_tr64<KS * 16 * BM * 2>(va[t].lo, fa[t] + so);
      lds_tr64<KS * 16 * BM * 2 + 4 * BM * 2>(va[t].hi, fa[t] + so);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      lds_tr64<KS * 16 * BN * 2>(vb[t].lo, fb[t] + so);
      lds_tr64<KS * 16 * BN * 2 + 4 * BN * 2>(vb[t].hi, fb[t] + so);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfmas = [&](const Frag (&va)[TM], const Frag (&vb)[TN]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 a = {va[i].lo, va[i].hi}, b = {vb[j].lo, vb[j].hi};
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int NRD = 2 * (TM + TN);
#define PGW_WAIT(n)                                                 \
  do {                                                              \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");      \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  issue(0, kt0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  Frag va0[TM], vb0[TN], va1[TM], vb1[TN];
  fetch(0, K0{}, va0, vb0);
  int stage = 0;
  if (kt0 + 1 < kt1) issue(1, kt0 + 1);
  bool pend = false;
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    __builtin_amdgcn_sched_barrier(0);
    fetch(stage, K1{}, va1, vb1);
    PGW_WAIT(NRD);
    mfmas(va0, vb0);
    if constexpr (PG_WG1_SPLIT_DMA) {
      if (pend) { issue(stage ^ 1, kt + 1, 2); pend = false; }
      __builtin_amdgcn_sched_barrier(0);
    }
    fetch(stage, K2{}, va0, vb0);
    PGW_WAIT(NRD);
    mfmas(va1, vb1);
    fetch(stage, K3{}, va1, vb1);
    PGW_WAIT(NRD);
    mfmas(va0, vb0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < kt1) {                           // the stage this tile just released; a full tile of MFMA time ahead of its wait
      if constexpr (PG_WG1_SPLIT_DMA) { issue(stage, kt + 2, 1); pend = true; }
      else issue(stage, kt + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(stage ^ 1, K0{}, va0, vb0);
    mfmas(va1, vb1);
    stage ^= 1;
  }
#undef PGW_WAIT

  // ---- epilogue: dW[tap][m][col_off + n] += acc   (lanes 0..31 = 32 consecutive columns: 128-byte segments)
  const int l31 = lane & 31, lhi = lane >> 5;
  float* const out = p.dW + (long)tap * p.Cout * p.ldw + p.col_off;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn0 + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float* o = out + (long)m * p.ldw + n;
        if (p.atomic) atomicAdd(o, acc[i][j][r]);
        else *o += acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: FOUR taps per workgroup for the thin layers (<= 128 channels on both sides at 128^2 / 256^2 resolution).
// One tap per workgroup (kernel above) moves (BM + BN) x 64 operand elements per 2 x BM x BN x 64 FLOP; with 128 x 128 /
// 128 x 64 tiles a K tile is only 8 / 4 MFMAs per wave, so the kernel is bound by the global->LDS latency of its two
// stages and by L2 bandwidth (round-2 table: 165 - 400 TFLOP/s, dec.5 / enc.1 = 6.7 ms of a 26 ms north-star pass).
// The four taps (r, s = 0..3) of one filter row read the SAME small-grid tile and, on the large grid, pixels
// 2 qx + s - 1 of ONE image row: 130 consecutive large pixels cover all four taps of 64 small pixels.  So a workgroup =
// (M tile, N tile, filter row r, K split): per K tile it DMAs 64 small + 130 large pixel rows (2.6 x fewer bytes), keeps
// four accumulator sets (one per s), reads the shared operand's fragments once per k-step for all four taps, and has
// 4 x the MFMA work between two barriers.  Three LDS stages, DMA issued two tiles ahead.
// LDS image of the large patch: de-interleaved into an EVEN-column and an ODD-column plane (tap s reads plane s & 1 from
// row p + (s >> 1)), so that the four pixel rows of a transposing read are consecutive LDS rows like the small operand's
// and the same chunk swizzle makes them conflict-free.  Requires power-of-two Hs, Ws with Ws >= 64 (a K tile = 64
// consecutive pixels of one small-grid row) and Hl = 2 Hs.
struct Wg4K {
  const unsigned short* sm;   // small-grid operand [N][Hs][Ws][Cs]
  const unsigned short* lg;   // large-grid operand [N][2 Hs][2 Ws][Cl]
  int Cs, Cl;
  int N, Hs, Ws, lgWs, lgHs;
  float* dW;                  // [16][Cout][ldw] fp32, accumulated
  int Cout, ldw, col_off;
  int ksplit, atomic, xcd_remap;
  int ktot;                   // N * Hs * Ws / 64
};

// R2: the small grid has 32 columns — a K tile is TWO small-grid rows; the large patch holds the two large rows 2 (qy + i) + r - 1,
// 66 pixels each, image row i at plane rows 36 i .. (36 = 32 + 4: the four pixel rows of a transposing read stay aligned).
#ifndef PG_WG4_SPLIT_DMA
#define PG_WG4_SPLIT_DMA 1
#endif
template <int BM, int BN, int AS, bool R2 = false>
__global__ __launch_bounds__(512, 2) void wgrad_bf16_tr4_kernel(const Wg4K p) {
  constexpr int BS = AS ? BM : BN, BL = AS ? BN : BM;         // widths of the small-grid (shared) / large-grid (per-tap) operand
  constexpr int TS = (BS == 128 && BL == 128) ? 2 : 1, TL = 1;   // MFMA tiles per wave along the shared / per-tap operand
  constexpr int WS = BS / (32 * TS), WL = BL / (32 * TL);
  static_assert(WS * WL == 8, "8 waves");
  constexpr int S_CPR = BS / 8, L_CPR = BL / 8;               // 16-byte chunks per pixel row
  constexpr int S_PPI = 64 / S_CPR, L_PPI = 64 / L_CPR;       // pixel rows per wave DMA instruction (1 KB)
  constexpr int PL = 72;                                      // rows per plane (65 / 36 + 33 used), a multiple of L_PPI
  constexpr int RP = 36, PMAX = R2 ? 65 : 129;                // R2: plane rows per image row; last patch pixel of a row
  constexpr int S_ST = 64 * BS * 2, L_ST = 2 * PL * BL * 2, STAGE = S_ST + L_ST, NST = 3;
  constexpr int S_NI = 64 / S_PPI, L_NI = 2 * PL / L_PPI;     // DMA instructions per tile
  constexpr int S_PASS = (S_NI + 7) / 8, L_PASS = (L_NI + 7) / 8;
  constexpr int NDMA = S_PASS + L_PASS;                       // DMA instructions per thread and tile
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bx = blockIdx.x, by = blockIdx.y, tr = blockIdx.z / p.ksplit, split = blockIdx.z - tr * p.ksplit;
  if (p.xcd_remap) {            // the four filter rows of one (M tile, N tile, K split) on ONE XCD (shared L2)
    const int mt = (int)gridDim.x, nt = (int)gridDim.y;
    const int L = bx + mt * (by + nt * (int)blockIdx.z);
    const int xcd = L & 7, j = L >> 3;
    tr = j & 3;
    const int u = (j >> 2) * 8 + xcd;
    bx = u % mt;
    by = (u / mt) % nt;
    split = u / (mt * nt);
  }
  const int m0 = bx * BM, n0 = by * BN;
  const int kper = (p.ktot + p.ksplit - 1) / p.ksplit;
  const int kt0 = split * kper, kt1 = min(p.ktot, kt0 + kper);
  if (kt0 >= kt1) return;

  const char* const s_base = uniform_ptr(reinterpret_cast<const char*>(p.sm));
  const char* const l_base = uniform_ptr(reinterpret_cast<const char*>(p.lg));
  const char* const zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));
  const int s_c0 = AS ? m0 : n0, l_c0 = AS ? n0 : m0;        // first channel of this tile in the small / large operand

  f32x16 acc[4][TS][TL];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
      for (int j = 0; j < TL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;

  // ---- DMA: lane -> (LDS row of the instruction, 16-byte slot); the global chunk it moves is slot ^ swizzle(LDS row)
  unsigned s_cofs[S_PASS];            // small operand: byte offset of (tile pixel, chunk) against the tile's first pixel
  unsigned l_cofs[L_PASS];            // large patch: byte offset of (patch pixel P, chunk) against patch pixel 0
  int l_P[L_PASS];                    // patch pixel of this lane's LDS row (0 .. PMAX valid; beyond: padding rows)
  int l_i[L_PASS];                    // R2: image row (0 / 1) of this lane's LDS row
#pragma unroll
  for (int i = 0; i < S_PASS; ++i) {
    const int ins = min(i * 8 + wave, S_NI - 1);
    const int row = ins * S_PPI + lane / S_CPR, slot = lane % S_CPR;
    s_cofs[i] = (unsigned)((row * p.Cs + s_c0 + ((slot ^ wg_swz<S_CPR>(row)) << 3)) * 2);
  }
#pragma unroll
  for (int i = 0; i < L_PASS; ++i) {
    const int ins = min(i * 8 + wave, L_NI - 1);             // surplus instructions repeat the last one (same data, same place)
    const int row = ins * L_PPI + lane / L_CPR, slot = lane % L_CPR;
    const int pl = row >= PL ? 1 : 0, pr = row - pl * PL;   // plane (0: even patch pixels, 1: odd), row inside the plane
    const int ir = R2 ? pr / RP : 0, jj = pr - ir * RP;     // image row of the tile, column inside the plane row
    const int P = (R2 && (ir > 1 || jj > 32)) ? 1000 : 2 * jj + pl;
    l_P[i] = P; l_i[i] = ir;
    l_cofs[i] = (unsigned)((P * p.Cl + l_c0 + ((slot ^ wg_swz<L_CPR>(row)) << 3)) * 2);
  }
  const unsigned Cs2 = (unsigned)p.Cs * 2u, Cl2 = (unsigned)p.Cl * 2u;
  const int Wl = 2 * p.Ws, Hl = 2 * p.Hs;
  // part: 0 = both operands, 1 = the small-grid rows only, 2 = the large-grid patch only (PG_WG4_SPLIT_DMA)
  auto issue = [&](int stage, int kt, int part = 0) {
    // tile coordinates are wave-uniform: small pixels q0 .. q0 + 63 of row (n, qy), columns qx0 ..; large row Y = 2 qy + tr - 1,
    // patch pixel P <-> large column 2 qx0 - 1 + P
    const int q0 = kt << 6;
    const int qx0 = q0 & (p.Ws - 1);
    const int qy = (q0 >> p.lgWs) & (p.Hs - 1);
    const int n = q0 >> (p.lgWs + p.lgHs);
    const int Y = 2 * qy + tr - 1;
    const bool live = kt < kt1;                               // past the end: zero rows (keeps the vmcnt accounting uniform)
    const bool row_ok = live & (Y >= 0) & (Y < Hl);
    const bool row_ok1 = live & (Y + 2 >= 0) & (Y + 2 < Hl);  // R2: the second small row of the tile
    const bool left_ok = qx0 > 0, right_ok = qx0 + (R2 ? 32 : 64) < p.Ws;
    const unsigned s_off = (unsigned)q0 * Cs2;
    const unsigned l_off = ((unsigned)(n * Hl + Y) * (unsigned)Wl + (unsigned)(2 * qx0 - 1)) * Cl2;
    const unsigned l_off1 = l_off + 2u * (unsigned)Wl * Cl2;
    float* const st = reinterpret_cast<float*>(smem + stage * STAGE);
    const char* const zp = zero_pg + (lane & 7) * 16;
    if (part != 2) {
#pragma unroll
    for (int i = 0; i < S_PASS; ++i) {
      const char* src = live ? s_base + (s_off + s_cofs[i]) : zp;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), st + min(i * 8 + wave, S_NI - 1) * 256, 16, 0, 0);
    }
    }
    if (part == 1) return;
#pragma unroll
    for (int i = 0; i < L_PASS; ++i) {
      const bool rok = R2 ? (l_i[i] ? row_ok1 : row_ok) : row_ok;
      const bool ok = rok & (l_P[i] <= PMAX) & (left_ok | (l_P[i] != 0)) & (right_ok | (l_P[i] != PMAX));
      const char* src = ok ? l_base + ((R2 && l_i[i] ? l_off1 : l_off) + l_cofs[i]) : zp;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), st + S_ST / 4 + min(i * 8 + wave, L_NI - 1) * 256, 16, 0, 0);
    }
  };

  // ---- operand fetch (transposing reads), lane roles as in the one-tap kernel
  const int wsh0 = (wave / WL) * (TS * 32), wlg0 = (wave % WL) * (TL * 32);
  const int r4 = (lane >> 2) & 3, cq = lane & 3, mb = (lane >> 4) & 1, kh = lane >> 5;
  unsigned fs[TS], fl[4][TL];
#pragma unroll
  for (int t = 0; t < TS; ++t) {
    const int chunk = ((wsh0 + 32 * t) >> 3) + 2 * mb + (cq >> 1);
    fs[t] = lds0 + (unsigned)((8 * kh + r4) * (BS * 2)) + (unsigned)(((chunk ^ wg_swz<S_CPR>(r4)) << 4) + ((cq & 1) << 3));
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      const int chunk = ((wlg0 + 32 * t) >> 3) + 2 * mb + (cq >> 1);
      const int row = (s & 1) * PL + (s >> 1) + 8 * kh + r4;             // tile pixel p = 8 kh + r4 (+ k-step) -> plane row p + (s >> 1)
      fl[s][t] = lds0 + S_ST + (unsigned)(row * (BL * 2)) + (unsigned)(((chunk ^ wg_swz<L_CPR>(row)) << 4) + ((cq & 1) << 3));
    }
  typedef unsigned long long u64;
  struct Frag { u64 lo, hi; };
  struct FragSet { Frag sh[TS]; Frag lg[4][TL]; };
  auto fetch = [&](int stage, auto ksc, FragSet& f) {
    constexpr int KS = decltype(ksc)::value;
    const unsigned so = (unsigned)(stage * STAGE);
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      lds_tr64<KS * 16 * BS * 2>(f.sh[t].lo, fs[t] + so);
      lds_tr64<KS * 16 * BS * 2 + 4 * BS * 2>(f.sh[t].hi, fs[t] + so);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < TL; ++t) {
        constexpr int XR = R2 ? (KS >> 1) * 4 * BL * 2 : 0;       // R2: k-steps 2, 3 are the second image row (+ 4 plane rows)
        lds_tr64<KS * 16 * BL * 2 + XR>(f.lg[s][t].lo, fl[s][t] + so);
        lds_tr64<KS * 16 * BL * 2 + 4 * BL * 2 + XR>(f.lg[s][t].hi, fl[s][t] + so);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfmas = [&](const FragSet& f) {
    __builtin_amdgcn_s_setprio(1);
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int j = 0; j < TL; ++j) {
          const u64x2 sh = {f.sh[i].lo, f.sh[i].hi}, lg = {f.lg[s][j].lo, f.lg[s][j].hi};
          if (AS) acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sh), __builtin_bit_cast(bf16x8, lg), acc[s][i][j], 0, 0, 0);
          else acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, lg), __builtin_bit_cast(bf16x8, sh), acc[s][i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int NRD = 2 * (TS + 4 * TL);
#define PGW_WAIT(n)                                                 \
  do {                                                              \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");      \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  issue(0, kt0);
  issue(1, kt0 + 1);
  issue(2, kt0 + 2);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  FragSet f0, f1;
  fetch(0, K0{}, f0);
  int stage = 0;
  int pend_stage = -1, pend_kt = 0;          // PG_WG4_SPLIT_DMA: the large-grid half of the last issued tile is still to go out
  for (int kt = kt0; kt < kt1; ++kt) {
    const int nstage = stage == NST - 1 ? 0 : stage + 1;
    __builtin_amdgcn_sched_barrier(0);
    fetch(stage, K1{}, f1);
    PGW_WAIT(NRD);
    mfmas(f0);
    if constexpr (PG_WG4_SPLIT_DMA) {
      if (pend_stage >= 0) { issue(pend_stage, pend_kt, 2); pend_stage = -1; }
      __builtin_amdgcn_sched_barrier(0);
    }
    fetch(stage, K2{}, f0);
    PGW_WAIT(NRD);
    mfmas(f1);
    fetch(stage, K3{}, f1);
    PGW_WAIT(NRD);
    mfmas(f0);
    // every read of this stage has landed (lgkmcnt 0) and tile kt + 1 is in LDS (only the NDMA loads of tile kt + 2 may be
    // outstanding): after the barrier the stage is free for tile kt + 3
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PG_WG4_SPLIT_DMA) {
      if (kt + 1 < kt1) { issue(stage, kt + 3, 1); pend_stage = stage; pend_kt = kt + 3; }      // the rest behind the next tile's first k-step
      else issue(stage, kt + 3);                    // last iteration: keep the vmcnt accounting complete
    } else {
      issue(stage, kt + 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < kt1) fetch(nstage, K0{}, f0);
    mfmas(f1);
    stage = nstage;
  }
#undef PGW_WAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummy loads past the end must not outlive the workgroup's LDS

  // ---- epilogue: dW[tap = 4 tr + s][m][col_off + n] += acc[s]
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float* const out = p.dW + (long)(tr * 4 + s) * p.Cout * p.ldw + p.col_off;
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
      for (int j = 0; j < TL; ++j) {
        // AS: rows (Cout) = shared operand, columns = per-tap operand; else the other way round
        const int mrow0 = m0 + (AS ? wsh0 + 32 * i : wlg0 + 32 * j);
        const int n = n0 + (AS ? wlg0 + 32 * j : wsh0 + 32 * i) + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mrow0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          float* o = out + (long)m * p.ldw + n;
          if (p.atomic) atomicAdd(o, acc[s][i][j][r]);
          else *o += acc[s][i][j][r];
        }
      }
  }
}

}  // namespace pg

using namespace pg;

// dW[16][Cout][ldw] (+)= weight gradient of one K-source of a k4/s2/p1 convolution (x_is_large = 1: Conv2d, X on the large
// grid; 0: ConvTranspose2d + crop, dY on the large grid).  x / dy: bf16 NHWC tensors (already normalised / activated /
// masked: pg_materialise_bf16).  Cx % 128 == 0, Cout % 128 == 0, Hl == 2 Hs, Wl == 2 Ws.
extern "C" int pg_wgrad_bf16_ex(const void* x_bf16, int32_t Cx, const void* dy_bf16, int32_t Cout, int32_t x_is_large, int32_t N,
                                int32_t Hs, int32_t Ws, int32_t Hl, int32_t Wl, float* dW, int32_t ldw, int32_t col_off,
                                int32_t ksplit, void* stream);

extern "C" int pg_wgrad_bf16(const void* x_bf16, int32_t Cx, const void* dy_bf16, int32_t Cout, int32_t x_is_large, int32_t N,
                             int32_t Hs, int32_t Ws, float* dW, int32_t ldw, int32_t col_off, int32_t ksplit, void* stream) {
  return pg_wgrad_bf16_ex(x_bf16, Cx, dy_bf16, Cout, x_is_large, N, Hs, Ws, 2 * Hs, 2 * Ws, dW, ldw, col_off, ksplit, stream);
}

// Large grid Hl x Wl with Hs = (Hl - 2) / 2 + 1 (k4 s2 p1): Hl = 2 Hs or, for a Conv2d (x_is_large = 1), 2 Hs + 1.
extern "C" int pg_wgrad_bf16_ex(const void* x_bf16, int32_t Cx, const void* dy_bf16, int32_t Cout, int32_t x_is_large, int32_t N,
                                int32_t Hs, int32_t Ws, int32_t Hl, int32_t Wl, float* dW, int32_t ldw, int32_t col_off,
                                int32_t ksplit, void* stream) {
  PG_REQUIRE(x_bf16 && dy_bf16 && dW && N > 0 && Hs > 0 && Ws > 0, "pg_wgrad_bf16: bad arguments");
  PG_REQUIRE(Hs == (Hl - 2) / 2 + 1 && Ws == (Wl - 2) / 2 + 1 && Hl >= 2 && Wl >= 2 && (x_is_large || (Hl == 2 * Hs && Wl == 2 * Ws)),
             "pg_wgrad_bf16: k4 s2 p1 geometry required (small %dx%d, large %dx%d)", Hs, Ws, Hl, Wl);
  PG_REQUIRE(Cx % 64 == 0 && Cout % 64 == 0 && (Cx % 128 == 0 || Cx == 64) && (Cout % 128 == 0 || Cout == 64) &&
             !(Cx == 64 && Cout == 64) && col_off >= 0 && col_off + Cx <= ldw,
             "pg_wgrad_bf16: channel counts must be multiples of 128, or 64 on one side (Cx=%d Cout=%d)", Cx, Cout);
  WgBf16K k;
  memset(&k, 0, sizeof(k));
  k.N = N; k.Hs = Hs; k.Ws = Ws; k.Hl = Hl; k.Wl = Wl;
  k.a_is_small = x_is_large ? 1 : 0;                 // A rows = Cout = the dY tensor
  if (x_is_large) { k.sm = (const unsigned short*)dy_bf16; k.Cs = Cout; k.lg = (const unsigned short*)x_bf16; k.Cl = Cx; }
  else { k.sm = (const unsigned short*)x_bf16; k.Cs = Cx; k.lg = (const unsigned short*)dy_bf16; k.Cl = Cout; }
  k.dW = dW; k.Cout = Cout; k.ldw = ldw; k.col_off = col_off;
  k.Q = (long)N * Hs * Ws;
  PG_REQUIRE((double)N * Hl * Wl * (Cx > Cout ? Cx : Cout) * 2.0 < 4294967296.0 && (double)N * Hs * Ws < 2147483000.0,
             "pg_wgrad_bf16: operands must be < 4 GiB each (32-bit byte offsets)");
  int bm = (Cout % 256 == 0) ? 256 : (Cout % 128 == 0 ? 128 : 64), bn = (Cx % 256 == 0) ? 256 : (Cx % 128 == 0 ? 128 : 64);
  const int ktot = (int)((k.Q + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  // ---- four taps per workgroup (wgrad_bf16_tr4_kernel): the thin layers, where the one-tap kernel's tiles are too small to
  // hide the global->LDS latency — and, measured, every other layer whose small grid has >= 64 columns as well (128-wide
  // tiles x 4 taps at 1000 - 1080 TFLOP/s against 256 x 256 one-tap tiles at 580 on half the chip; north-star pass 25.5 ->
  // 24.5 ms thin layers only (PG_WGTR4=1) -> 24.3 ms all eligible layers (default, 2).  PG_WGTR4=0 disables the kernel.
  {
    static const int mode = getenv("PG_WGTR4") ? atoi(getenv("PG_WGTR4")) : 2;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const bool geo = pow2(Hs) && pow2(Ws) && (Ws >= 64 || (Ws == 32 && Hs >= 2)) && Hl == 2 * Hs && Wl == 2 * Ws;
    const bool r2 = Ws == 32;
    const bool thin = bm < 256 || bn < 256;
    if (mode > 0 && geo && (thin || mode == 2) && ksplit <= 0) {
      const int tm = Cout % 128 == 0 ? 128 : 64, tn = Cx % 128 == 0 ? 128 : 64;
      Wg4K q;
      memset(&q, 0, sizeof(q));
      q.sm = k.sm; q.lg = k.lg; q.Cs = k.Cs; q.Cl = k.Cl;
      q.N = N; q.Hs = Hs; q.Ws = Ws;
      q.lgWs = __builtin_ctz((unsigned)Ws); q.lgHs = __builtin_ctz((unsigned)Hs);
      q.dW = dW; q.Cout = Cout; q.ldw = ldw; q.col_off = col_off;
      q.ktot = ktot;
      const int mt4 = Cout / tm, nt4 = Cx / tn;
      // workgroups per launch (x 4 taps each): 256 at batch 32 (north-star pass 17.05 ms; 128: 17.34, 384: 17.34), 128 at small
      // batch — every split adds a full set of float atomics on dW next to main-stream launches that are latency-bound themselves
      // (round 5, bf16 data path, 256 / 128 / 64: batch 4 608 / 634 / 591 img/s, batch 8 829 / 845 / 756, batch 16 1021 / 1016 / -)
      PG_ENV_INT(target4_env, "PG_WGTR4_TARGET", 0);
      const int target4 = target4_env > 0 ? target4_env : (N <= 12 ? 128 : 256);
      const long base4 = (long)mt4 * nt4 * 4;
      int ks4 = (int)((target4 + base4 - 1) / base4);
      if (ks4 > ktot / 8) ks4 = ktot / 8;                       // >= 8 K tiles per workgroup
      if (ks4 < 1) ks4 = 1;
      while (ks4 > 1 && (long)(ks4 - 1) * ((ktot + ks4 - 1) / ks4) >= ktot) --ks4;
      if (deterministic()) ks4 = 1;       // PG_DETERMINISTIC: no float atomics on dW
      q.ksplit = ks4; q.atomic = ks4 > 1 ? 1 : 0;
      q.xcd_remap = (((long)mt4 * nt4 * ks4) % 8 == 0 && !env().no_xcd_swizzle) ? 1 : 0;
      dim3 grid4(mt4, nt4, 4 * ks4);
#define PGW4_LAUNCH(M_, N_)                                                                                              \
  do {                                                                                                                   \
    if (r2) {                                                                                                            \
      if (k.a_is_small) PG_KLAUNCH((wgrad_bf16_tr4_kernel<M_, N_, 1, true>), grid4, dim3(512), 0, st, q);        \
      else PG_KLAUNCH((wgrad_bf16_tr4_kernel<M_, N_, 0, true>), grid4, dim3(512), 0, st, q);                     \
    } else {                                                                                                             \
      if (k.a_is_small) PG_KLAUNCH((wgrad_bf16_tr4_kernel<M_, N_, 1, false>), grid4, dim3(512), 0, st, q);       \
      else PG_KLAUNCH((wgrad_bf16_tr4_kernel<M_, N_, 0, false>), grid4, dim3(512), 0, st, q);                    \
    }                                                                                                                    \
  } while (0)
      if (tm == 128 && tn == 128) PGW4_LAUNCH(128, 128);
      else if (tm == 128) PGW4_LAUNCH(128, 64);
      else PGW4_LAUNCH(64, 128);
#undef PGW4_LAUNCH
      PG_LAUNCH_OK("pg_wgrad_bf16 (four-tap kernel)");
      last_info() = 6 | (ks4 << 16) | (1 << 30);
      return 0;
    }
  }
  // few pixels (deep layers at small batch): K cannot be split (>= 16 K tiles per workgroup), so 256-wide tiles leave most
  // of the chip idle (64 workgroups for a 512 x 512 filter) — halve the tile sides until ~128 workgroups exist
  static const bool small_tiles = getenv("PG_WGTR_NO_SMALL_TILES") == nullptr;
  // (round 5) with few pixels these launches are not contractions any more: 1 - 16 K tiles, then a 256 KB read-modify-write of the
  // filter tile by ONE workgroup — 40 - 55 us per launch at batch 4 for 17 MB of dW (0.3 - 0.4 TB/s) with 64 - 128 workgroups.  The
  // tile sides are halved (256 -> 128, then 64 columns) until ~1024 workgroups share the filter (PG_WGTR_SMALL_WGS; the rule stopped
  // at 128 workgroups and 128-wide tiles until round 4).  bf16 batch 4, three A/B rounds on one box: 593 - 600 -> 620 - 625 img/s;
  // configs[2] 919 -> 945; the batch-32 pass unchanged (its deep layers have >= 32 K tiles).
  static const int small_wgs = getenv("PG_WGTR_SMALL_WGS") ? atoi(getenv("PG_WGTR_SMALL_WGS")) : 1024;
  static const int small_kt = getenv("PG_WGTR_SMALL_KT") ? atoi(getenv("PG_WGTR_SMALL_KT")) : 32;
  static const bool small_64 = getenv("PG_WGTR_NO_SMALL_64") == nullptr;
  if (small_tiles && ksplit <= 0 && ktot < small_kt) {
    if ((long)(Cout / bm) * (Cx / bn) * 16 < small_wgs && bn == 256) bn = 128;
    if ((long)(Cout / bm) * (Cx / bn) * 16 < small_wgs && bm == 256) bm = 128;
    if (small_64 && (long)(Cout / bm) * (Cx / bn) * 16 < small_wgs && bn == 128 && bm == 128) bn = 64;
  }
  const int mt = Cout / bm, nt = Cx / bn;
  int ks = ksplit;
  if (ks <= 0) {
    // ~128 workgroups per launch (half a round of the 256 CUs), >= 16 K tiles per workgroup.  Weight gradients run on the
    // side stream NEXT TO the data-gradient chain: a launch that wants every CU only takes them from the main stream, and
    // every extra split adds a full set of float atomics on dW.  Swept on the box (tools/sweep_wgrad_bf16_target.sh), img/s at 256^2 batch
    // 32 / batch 4 / 224^2 P=32 batch 8: target 768 -> 776 / 464 / 704, 256 -> 788 / 465 / 713, 128 -> 793 / 475 / 724,
    // 64 -> 681 / 439 / 656.  (Round 4, per layer in isolation at batch 32 — tools/layer_bench.py — 256 workgroups win on the two layers
    // with >= 128 K tiles: dec.2 418 -> 349 us, enc.4 135 -> 117 us; in the pass, next to the main stream's kernels, the same rule
    // LOSES: north-star 18.22 -> 18.33 ms, batch-32 step 1110 -> 1100 img/s.  Kept at 128.)
    const long base = (long)mt * nt * 16;
    PG_ENV_INT(target, "PG_WGTR_TARGET", 128);
    // thin tiles with a long K (the discriminator's 64 -> 128 layer on 63 x 63 maps: one 128 x 64 tile per tap, 1985 K tiles — 0.59 ms
    // per launch at batch 32 with 8 splits): these are latency-bound per workgroup, so they take 512 workgroups
    static const int thin_target = getenv("PG_WGTR_THIN_TARGET") ? atoi(getenv("PG_WGTR_THIN_TARGET")) : 512;
    const long tgt = ((long)bm * bn <= 128 * 64 && ktot >= 1024) ? thin_target : target;
    ks = (int)((tgt + base - 1) / base);
    if (ks > ktot / 16) ks = ktot / 16;
    if (ks < 1) ks = 1;
  }
  if (ks > ktot) ks = ktot;
  while (ks > 1 && (long)(ks - 1) * ((ktot + ks - 1) / ks) >= ktot) --ks;      // no empty split
  if (ks >= 8 && ks % 8 != 0 && ktot / ((ks + 7) / 8 * 8) >= 8) ks = (ks + 7) / 8 * 8;       // whole XCD groups of taps
  while (ks > 1 && (long)(ks - 1) * ((ktot + ks - 1) / ks) >= ktot) --ks;
  if (deterministic()) ks = 1;            // PG_DETERMINISTIC: no float atomics on dW
  k.ksplit = ks; k.atomic = ks > 1 ? 1 : 0;
  k.xcd_remap = (((long)mt * nt * ks) % 8 == 0 && !env().no_xcd_swizzle) ? 1 : 0;
  dim3 grid(mt, nt, 16 * ks);
#define PGW_LAUNCH(M_, N_)                                                                                   \
  do {                                                                                                       \
    if (k.a_is_small) PG_KLAUNCH((wgrad_bf16_tr_kernel<M_, N_, 1>), grid, dim3(512), 0, st, k);      \
    else PG_KLAUNCH((wgrad_bf16_tr_kernel<M_, N_, 0>), grid, dim3(512), 0, st, k);                   \
  } while (0)
  if (bm == 256 && bn == 256) PGW_LAUNCH(256, 256);
  else if (bm == 128 && bn == 256) PGW_LAUNCH(128, 256);
  else if (bm == 256 && bn == 128) PGW_LAUNCH(256, 128);
  else if (bn == 64) { if (bm == 256) PGW_LAUNCH(256, 64); else PGW_LAUNCH(128, 64); }         // 64-channel X (encoder level 1)
  else if (bm == 64) { if (bn == 256) PGW_LAUNCH(64, 256); else PGW_LAUNCH(64, 128); }         // 64-channel dY (last decoder block)
  else PGW_LAUNCH(128, 128);
#undef PGW_LAUNCH
  PG_LAUNCH_OK("pg_wgrad_bf16");
  last_info() = 6 | (ks << 16) | (1 << 30);
  return 0;
}
