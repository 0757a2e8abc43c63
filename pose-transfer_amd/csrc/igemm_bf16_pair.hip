// 256-row bf16 implicit-GEMM convolution with TAP-PAIR SHARING of the A operand (round 4).
//
// conv_bf16_big_kernel (igemm_bf16.hip) moves one A tile (256 rows x 64 channels) and one B tile global -> LDS per K tile.  Timing
// experiments on that kernel (the same instruction stream with pieces removed, random and all-zero operands)
// say where a launch's time goes on this chip: the matrix pipe is POWER-limited (a K loop of nothing but MFMAs reaches 1340 TFLOP/s
// on random data, 1640 on zeros), LDS operand reads cost 1 - 3 %, the per-tile barrier 1 - 5 %, and the global -> LDS DMA 21 - 23 %,
// in proportion to the number of DMA instructions.  So the lever is DMA volume per FLOP.
//
// Two taps of a k4 s2 convolution / transposed convolution that differ only by ONE STEP ALONG X read almost the same pixels: tap
// (dy, dx + si) of output-grid position (qy, qx) is tap (dy, dx) of position (qy, qx + 1).  This kernel orders the K loop
// (tap pair, channel chunk, tap of the pair) and DMAs ONE A tile per (pair, chunk): LDS row rho holds the pixel of "slot" k of an
// image row of the tile for the pair's first tap, with Gx + 1 slots per image row (slot Gx = what the last position's second tap
// reads), so the second tap of tile row r is simply LDS row rho(r) + 1.  rho(r) = r + (number of image-row starts in the tile
// before r): at most 16 (8) extra rows on a 256 (512) row tile.  A-tile DMA instructions per tap: 32 -> 17; all DMA instructions of
// a 256 x 256 launch - 23 %, of a 256 x 128 launch - 31 %, of a 512 x 64 launch - 43 %.  The operand stages become two rings: A
// (2 x 34 KB, switched every second K tile) and B (2 x 32 KB, every tile).
//
// Everything else is conv_bf16_big_kernel: 8 waves, wave tile 128 x 64 (64 x 64), v_mfma_f32_32x32x16_bf16, chunk-swizzled unpadded
// LDS images, one barrier per K tile in front of the last k-step's MFMAs, the shared epilogues (igemm_bf16_epi.h).  Because an
// LDS row of A is not a tile row's private copy any more (the swizzle is a function of the LDS row), an A fragment address is
// per (lane, tap of the pair, M tile): two VALU instructions per ds_read_b128 instead of an immediate offset.
// Host conditions (conv_impl): every phase's taps pair up, ksplit = 1, Gx large enough for the extra rows.
//
// MG (round 5) — X-PHASE MERGING for the transposed k4 s2 convolutions with few output channels (the last decoder block's forward,
// the data gradients of encoder levels 1 / 2: N = 128 or 64).  Sub-pixel phase (py, 0) reads taps dx = -1, 0 and phase (py, 1) reads
// dx = 0, +1 of the SAME input rows: with Gx + 2 slots per image row in the A tile, one tile serves both phases — the waves of the
// left column half (phase px = 0) read LDS rows rho, rho + 1 for the two taps of a step, the waves of the right half (px = 1) rows
// rho + 1, rho + 2 — and the B tile holds the two phases' weights side by side.  The workgroup tile becomes 256 x 2 Cout: the
// DMA volume per FLOP of the 256 x 256 kernel instead of the 256 x 128 one (A tile - 47 %, all DMA - 24 %), half the workgroups,
// and the two phases' outputs are neighbouring pixels: column n of the tile is channel n mod Cout of pixel (oy, 2 qx + n / Cout),
// i.e. offset n from the left pixel — the epilogues see an image of half the width with 2 Cout channels (host: n_cnt, dst C doubled,
// row table entries hold the pixel-PAIR index), nothing else changes.
#include "igemm_bf16_epi.h"

namespace pg {

struct ARow {      // LDS row of the A ring -> the input pixel it holds, before the tap offset
  int n;           // sample, -1 = beyond the problem
  short iy, ix;    // qy * si, slot * si
};

template <int BN, bool MG = false>
__global__ __launch_bounds__(512, 2) void conv_bf16_pair_kernel(const ConvK p) {
  static_assert(!(MG && BN == 64), "x-phase merging: 256-row tiles");
  constexpr int XS = MG ? 2 : 1;                         // extra slots per image row of the A tile
  constexpr int BM = (BN == 64) ? 512 : 256;
  constexpr int WGN = BN / 64, WGM = 8 / WGN;
  constexpr int TM = BM / WGM / 32, TN = 2;
  constexpr int ROWB = 128;
  constexpr int AXR = (BM == 512) ? 8 : 16;              // extra LDS rows of an A tile: one per image row that starts inside the tile (+1)
  constexpr int AROWS = BM + AXR;
  constexpr int A_NI = AROWS / 8;                        // wave DMA instructions per A tile (8 rows of 128 B each)
  constexpr int A_PASS = (A_NI + 7) / 8, B_PASS = BN / 64;
  constexpr int A_ST = AROWS * ROWB, B_ST = BN * ROWB;
  constexpr int B_OFF = 2 * A_ST;                        // [A ring: 2 stages][B ring: 2 stages]
  constexpr int OPS = 2 * (A_ST + B_ST);
  constexpr int ROWS_OFF = OPS, AROW_OFF = ROWS_OFF + BM * (int)sizeof(RowB), TAPS_OFF = AROW_OFF + AROWS * (int)sizeof(ARow);
  constexpr int STAT_OFF = (TAPS_OFF + MAXTAP * 8 + 7) & ~7, STAT_N = 8;      // two tap tables (MG: the right half's phase)
  static_assert(8 * (32 * (32 * TN + 4)) * 4 <= OPS && AROWS % 8 == 0, "epilogue tiles / DMA rows");
  static_assert(STAT_OFF + STAT_N * 16 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char smem[STAT_OFF + STAT_N * 2 * 8];
  RowB* rows = reinterpret_cast<RowB*>(smem + ROWS_OFF);
  int* taps_l = reinterpret_cast<int*>(smem + TAPS_OFF);
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_swizzle & 1) {          // as conv_bf16_big_kernel: (phase, N tile, M tile) order per XCD
    const int mt = (int)gridDim.x, nt = (int)gridDim.y, P = (int)gridDim.z;
    const int L = bx + mt * (by + nt * bz);
    const int xcd = L & 7, j = L >> 3;
    bz = j % P;
    const int r = j / P;
    by = r % nt;
    bx = (r / nt) * 8 + xcd;
  }
  const int phase = bz;             // host: ksplit == 1, no tap batch
  float* const out_g = p.out;
  const int m0 = bx * BM, nb0 = by * BN;
  const int ntap = p.ntap[phase];
  const int gx = p.Gx;
  const int qx0 = m0 % gx, R0 = m0 / gx;

  if (tid >= 64 && tid < 64 + STAT_N * 2) reinterpret_cast<double*>(smem + STAT_OFF)[tid - 64] = 0.0;
  if (tid < MAXTAP)
    taps_l[tid] = (p.dy[phase][tid] & 0xff) | ((p.dx[phase][tid] & 0xff) << 8) | ((int)p.wtap[phase][tid] << 16);
  if (MG && tid >= 64 && tid < 64 + MAXTAP) {           // the right half's taps (phase (py, 1)): host keeps them in slot 2 + py
    const int q = tid - 64;
    taps_l[MAXTAP + q] = (p.dy[2 + phase][q] & 0xff) | ((p.dx[2 + phase][q] & 0xff) << 8) | ((int)p.wtap[2 + phase][q] << 16);
  }
  if (tid < BM) {
    RowB ri;
    const int m = m0 + tid;
    ri.n = -1; ri.opix = 0; ri.iy = 0; ri.ix = 0; ri.oy = 0; ri.ox = 0;
    if (m < p.M) {
      const int gg = p.Gy * p.Gx;
      const int n = m / gg;
      const int rem = m - n * gg;
      const int qy = rem / p.Gx;
      const int qx = rem - qy * p.Gx;
      const int oy = qy * p.so + p.phy[phase];
      const int ox = qx * p.so + p.phx[phase];
      if (oy < p.Ho && ox < p.Wo) {
        ri.n = n; ri.iy = (short)(qy * p.si); ri.ix = (short)(qx * p.si); ri.oy = (short)oy; ri.ox = (short)ox;
        ri.opix = (n * p.Ho + oy) * p.Wo + ox;
        if (MG) ri.opix >>= 1;                            // pixel PAIR (ox = 2 qx, Wo even): the epilogues' rows are 2 Cout wide
      }
    }
    rows[tid] = ri;
  }
  for (int rho = tid; rho < AROWS; rho += 512) {          // LDS row -> (local image row j, slot k): rho + qx0 = j (Gx + XS) + k
    const int t = rho + qx0;
    const int j = t / (gx + XS), k = t - j * (gx + XS);
    const int Rg = R0 + j;
    const int n = Rg / p.Gy, qy = Rg - n * p.Gy;
    ARow a;
    a.n = n < p.N ? n : -1; a.iy = (short)(qy * p.si); a.ix = (short)(k * p.si);
    reinterpret_cast<ARow*>(smem + AROW_OFF)[rho] = a;
  }
  __syncthreads();

  const int cpt = p.Ctot / 64;                      // channel chunks = K tiles per tap
  const int npc = (ntap >> 1) * cpt;                // (tap pair, chunk) steps; two K tiles each

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

  // ---- DMA loader state
  const char* const zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));
  const char* const wp = uniform_ptr(reinterpret_cast<const char*>(p.W));
  const int chunk = (tid & 7) ^ ((tid >> 4) & 7);   // source chunk of this lane's LDS slot (swizzle by LDS row, rows 64 i + (tid >> 3))
  // per-row source: 32-bit byte offset against a wave-uniform base (the source tensor of the chunk / the weights) + a validity bit
  // (invalid rows read the zero page) — 64-bit pointers per row cost the 256 x 256 variant its registers (host: tensors < 4 GiB)
  unsigned pa[A_PASS], pb[B_PASS];
  unsigned pa_ok = 0, pb_ok = 0;
  const char* a_src = zero_pg;                       // uniform: source tensor of the A cursor's chunk
  long wdelta = 0, wdelta1 = 0;                      // uniform: byte offset from the pair's first to its second tap in W (MG: per column half)
  int a_g = 0, a_ci = 0, a_left = npc;               // A cursor: next (pair, chunk) to load; steps not yet loaded
  int b_g = 0, b_ci = 0, b_left = npc;               // B cursor: next (pair, chunk); its two tiles are issued one by one
  auto rebuild_a = [&]() __attribute__((always_inline)) {
    const int tp = lds_rd32_now(lds0 + TAPS_OFF + (2 * a_g) * 4);
    const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
    const int cc = a_ci * 64;
    const char* sp = reinterpret_cast<const char*>(p.src[0].ptr);
    int sC = p.src[0].C, cs = 0;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.nsrc && cc >= p.cstart[q]) { sp = reinterpret_cast<const char*>(p.src[q].ptr); sC = p.src[q].C; cs = p.cstart[q]; }
    a_src = uniform_ptr(sp);
    const int cl = cc - cs + chunk * 8;
    pa_ok = 0;
    int rn[A_PASS], ryx[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int rho = min((tid >> 3) + 64 * i, AROWS - 1);
      const unsigned ra = lds0 + AROW_OFF + (unsigned)(rho * (int)sizeof(ARow));
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4" : "=&v"(rn[i]), "=&v"(ryx[i]) : "v"(ra));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int iy = (int)(short)(ryx[i] & 0xffff) + dyv, ix = (ryx[i] >> 16) + dxv;
      const bool ok = (rn[i] >= 0) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      pa[i] = ok ? ((unsigned)((rn[i] * p.Hi + iy) * p.Wi + ix) * (unsigned)sC + (unsigned)cl) * 2u : 0u;
      pa_ok |= (ok ? 1u : 0u) << i;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_a = [&]() __attribute__((always_inline)) {            // after a step's A tile was issued
    if (--a_left <= 0) return;
    if (++a_ci == cpt) { a_ci = 0; ++a_g; rebuild_a(); return; }
    bool src_edge = false;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && a_ci * 64 == p.cstart[q]) src_edge = true;
    if (src_edge) { rebuild_a(); return; }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) pa[i] += ROWB;       // (invalid rows: the offset is not used)
  };
  auto rebuild_b = [&]() __attribute__((always_inline)) {
    const int tp0 = lds_rd32_now(lds0 + TAPS_OFF + (2 * b_g) * 4), tp1 = lds_rd32_now(lds0 + TAPS_OFF + (2 * b_g + 1) * 4);
    const int base = __builtin_amdgcn_readfirstlane((tp0 >> 16) * p.wCout);
    wdelta = (long)__builtin_amdgcn_readfirstlane((tp1 >> 16) - (tp0 >> 16)) * p.wCout * p.wCin * 2;
    int base1 = 0;
    if constexpr (MG) {                              // the right column half: phase (py, 1)'s pair g
      const int tq0 = lds_rd32_now(lds0 + TAPS_OFF + (MAXTAP + 2 * b_g) * 4), tq1 = lds_rd32_now(lds0 + TAPS_OFF + (MAXTAP + 2 * b_g + 1) * 4);
      base1 = __builtin_amdgcn_readfirstlane((tq0 >> 16) * p.wCout);
      wdelta1 = (long)__builtin_amdgcn_readfirstlane((tq1 >> 16) - (tq0 >> 16)) * p.wCout * p.wCin * 2;
    }
    pb_ok = 0;
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      const bool right = MG && 64 * i >= BN / 2;      // compile-time per pass: rows [BN / 2, BN) of the B tile
      const int n = nb0 + (tid >> 3) + 64 * i - (right ? BN / 2 : 0);
      const bool ok = MG ? true : n < p.n_cnt;        // (MG: host guarantees Cout == BN / 2)
      pb[i] = ok ? ((unsigned)((right ? base1 : base) + p.n_off + n) * (unsigned)p.wCin + (unsigned)(b_ci * 64 + chunk * 8)) * 2u : 0u;
      pb_ok |= (ok ? 1u : 0u) << i;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_b = [&]() __attribute__((always_inline)) {            // after BOTH tiles of a step were issued
    if (--b_left <= 0) return;
    if (++b_ci == cpt) { b_ci = 0; ++b_g; rebuild_b(); return; }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) pb[i] += ROWB;
  };
  auto issue_a = [&](int stage) __attribute__((always_inline)) {
    float* const As = reinterpret_cast<float*>(smem + stage * A_ST);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
      if (i * 8 + wave < A_NI) {      // wave-uniform: the last pass covers only the first extra rows
        const char* src = ((pa_ok >> i) & 1u) ? a_src + pa[i] : zero_pg + (tid & 7) * 16;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), As + (i * 8 + wave) * 256, 16, 0, 0);
      }
  };
  auto issue_b = [&](int stage, int second) __attribute__((always_inline)) {
    float* const Bs = reinterpret_cast<float*>(smem + B_OFF + stage * B_ST);
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      const long wd = (MG && 64 * i >= BN / 2) ? wdelta1 : wdelta;
      const char* src = ((pb_ok >> i) & 1u) ? wp + (second ? wd : 0) + pb[i] : zero_pg + (tid & 7) * 16;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), Bs + (i * 8 + wave) * 256, 16, 0, 0);
    }
  };

  // ---- operand fetch.  A: LDS row of tile row r: rho(r) = r + (qx0 + r) / Gx (+ 1 for the pair's second tap).  The 16-byte slot
  // of k-step ks and lane half lhi is (2 ks + lhi) ^ swizzle(LDS row) = slot(ks = 0) ^ (2 ks): ONE address per (tap of the pair, M
  // tile) for ks = 0, and address(ks) = address(0) ^ (ks << 5).  The xor sits in the asm statement of its read (as a separate
  // expression the compiler hoists all 24 loop-invariant results out of the K loop: 260 spilled registers); ring stages and the B
  // tile's second 32 rows are instruction offsets.
  constexpr bool BIGA = A_ST > 60000;                 // 512-row tiles: the second A stage is beyond the 16-bit instruction offset
  unsigned abase[BIGA ? 2 : 1][2][TM];                // [ring stage (512-row tiles only)][tap of the pair][M tile]
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm0 + 32 * i + l31;
    const int rho = r + XS * ((qx0 + r) / gx) + ((MG && wn0 >= BN / 2) ? 1 : 0);      // (MG, right half: taps dx = 0, +1 -> one slot further)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = rho + j, s = (row >> 1) & 7;
      abase[0][j][i] = lds0 + (unsigned)(row * ROWB) + (unsigned)((lhi ^ s) << 4);
      if constexpr (BIGA) abase[1][j][i] = abase[0][j][i] + (unsigned)A_ST;
    }
  }
  const int swr = (l31 >> 1) & 7;
  const unsigned fb0 = lds0 + B_OFF + (unsigned)((wn0 + l31) * ROWB) + (unsigned)((lhi ^ swr) * 16);
  // AS / BSG: ring stages, JT: tap of the pair, KS: k-step — all compile-time
  auto fetch = [&](auto asg, auto jt, auto bsg, auto ksc, f32x4 (&va)[TM], f32x4 (&vb)[TN]) __attribute__((always_inline)) {
    constexpr int AS = decltype(asg)::value, JT = decltype(jt)::value, BSG = decltype(bsg)::value, KS = decltype(ksc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned t;
#ifdef PG_PAIR_NOXOR
      asm volatile("ds_read_b128 %0, %2 offset:%4" : "=v"(va[i]), "=&v"(t) : "v"(abase[BIGA ? AS : 0][JT][i]), "n"(KS * 32), "n"(BIGA ? 0 : AS * A_ST));
#else
      asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                   : "=v"(va[i]), "=&v"(t) : "v"(abase[BIGA ? AS : 0][JT][i]), "n"(KS * 32), "n"(BIGA ? 0 : AS * A_ST));
#endif
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      unsigned t;
#ifdef PG_PAIR_NOXOR
      asm volatile("ds_read_b128 %0, %2 offset:%4" : "=v"(vb[j]), "=&v"(t) : "v"(fb0), "n"(KS * 32), "n"(BSG * B_ST + j * 32 * ROWB));
#else
      asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                   : "=v"(vb[j]), "=&v"(t) : "v"(fb0), "n"(KS * 32), "n"(BSG * B_ST + j * 32 * ROWB));
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfmas = [&](const f32x4 (&va)[TM], const f32x4 (&vb)[TN]) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va[i]), __builtin_bit_cast(bf16x8, vb[j]),
                                                             acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int NRD = TM + TN;
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;

  // ---- prologue: step 0's A tile -> A stage 0, its first B tile -> B stage 0
  rebuild_a();
  issue_a(0);
  advance_a();
  rebuild_b();
  issue_b(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // in flight before the loop: step 0's second B tile, step 1's A tile
  issue_b(1, 1);
  advance_b();
  if (npc > 1) { issue_a(1); advance_a(); }
  f32x4 va0[TM], vb0[TN], va1[TM], vb1[TN];
  fetch(I0{}, I0{}, I0{}, I0{}, va0, vb0);
  bool pend_a = false;               // an A tile whose ring stage was released at the last switch is still to be issued
  int pend_stage = 0;
  // one (pair, chunk) step = two K tiles; AS = its A ring stage (compile-time: the loop below alternates the two instantiations)
  // DMA placement (measured: the position right behind the tile-switch barrier, where all eight waves issue at once and the next
  // tile's first operand fetch follows, is the expensive one): a released ring stage is refilled behind the FIRST k-step of the next
  // tile (B) and behind its SECOND k-step (A, every other tile) — PG_PAIR_DMA_AT_BARRIER=1 builds the earlier placement.
#ifndef PG_PAIR_DMA_AT_BARRIER
#define PG_PAIR_DMA_AT_BARRIER 0
#endif
#ifndef PG_PAIR_A_AT_BARRIER
#define PG_PAIR_A_AT_BARRIER 1     // the A tile (every other switch) right behind the barrier, B tiles behind the first k-step
#endif
  int pend_b = -1;                   // B ring stage to refill behind the next first k-step (-1: none); its tile is `pend_b`'s tap of the B cursor
  auto step = [&](auto asg, int pc) __attribute__((always_inline)) {
    constexpr int AS = decltype(asg)::value;
    typedef std::integral_constant<int, AS> IA;
    typedef std::integral_constant<int, AS ^ 1> IN;
    const bool more = pc + 1 < npc;
    // ================= first tile of the step: tap 2g, B stage 0
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I0{}, I0{}, I1{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    if (pend_b == 1) { issue_b(1, 1); advance_b(); pend_b = -1; }          // B stage 1 was released at the last switch
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I0{}, I0{}, I2{}, va0, vb0);
    PGB_LDS_WAIT(NRD);
    mfmas(va1, vb1);
    if (!PG_PAIR_A_AT_BARRIER && pend_a) { issue_a(pend_stage); advance_a(); pend_a = false; }      // the A stage released at the last switch
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I0{}, I0{}, I3{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    // tile switch: this wave's reads of B stage 0 have landed, its share of the next tile's operands has landed; after the
    // barrier both hold for every wave, and B stage 0 is free: the next step's first B tile goes there
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if (PG_PAIR_DMA_AT_BARRIER) issue_b(0, 0); else pend_b = 0;
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I1{}, I1{}, I0{}, va0, vb0);
    mfmas(va1, vb1);
    // ================= second tile: tap 2g + 1 = the same A stage one LDS row further, B stage 1
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I1{}, I1{}, I1{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    if (pend_b == 0) { issue_b(0, 0); pend_b = -1; }                       // B stage 0 was released at the last switch
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I1{}, I1{}, I2{}, va0, vb0);
    PGB_LDS_WAIT(NRD);
    mfmas(va1, vb1);
    fetch(IA{}, I1{}, I1{}, I3{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    // tile switch: B stage 1 and A stage AS are free
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if (PG_PAIR_DMA_AT_BARRIER) { issue_b(1, 1); advance_b(); } else pend_b = 1;
    }
    if (pc + 2 < npc) {
      if (PG_PAIR_A_AT_BARRIER) { issue_a(AS); advance_a(); }
      else { pend_a = true; pend_stage = AS; }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(IN{}, I0{}, I0{}, I0{}, va0, vb0);
    mfmas(va1, vb1);
  };
  for (int pc = 0; pc < npc; pc += 2) {
    step(I0{}, pc);
    if (pc + 1 < npc) step(I1{}, pc + 1);
  }

  // ------------------------------------------------------------------ epilogue (shared; rows outside the problem are NOT zero here)
  big_epilogue<TM, TN, STAT_OFF, STAT_N>(p, acc, smem, rows, tid, m0, nb0, wm0, wn0, bx, by, bz, 0, out_g, true, false, [](int) {});
}

#ifdef PG_TIMING_EXPERIMENTS      // (round 6) the persistent form lost in every variant: it is built only into the timing library
// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form (round 5).  conv_bf16_pair_kernel above runs ONE tile per workgroup, one workgroup per CU: between two tiles a
// CU sees the old workgroup drain, the dispatch of the next one, its row / tap tables (1.1 - 1.3 us) and the full latency of its
// first operand tile (2.2 - 3.3 us) — 4 - 6 us in which the matrix pipe idles, 6 - 10 % of a dec.5-class tile (50 - 90 us) and
// 15 - 25 % of the short-K tiles of encoder levels 1 / 2 (tools/conv_timeline.py, round 3).  Here a workgroup walks tiles
// L, L + G, L + 2 G ... (G = gridDim.x = the CU count; with the XCD order the walk stays on its XCD) and the NEXT tile's
// prologue — DMA tables, tap tables, its first A tile and first B tile — is issued right behind the barrier that ends the K loop,
// i.e. UNDER the current tile's epilogue (stores, forward-value loads, statistics: VALU and memory work that leaves the LDS rings
// alone).  For that the epilogue's transposition tiles move out of ring stages 0 (the next tile's first DMA lands there) into
// stages 1 of the two rings and the LDS behind the tables; the DMA table and the tap tables are double-buffered.  The loader
// state is rebuilt after the epilogue instead of being kept in registers across it.  BN = 64 (512-row tiles) has no LDS left for
// this: it takes the persistent walk without the early prologue.
template <int BN, bool MG = false, bool EARLY = true>
__global__ __launch_bounds__(512, 2) void conv_bf16_pairp_kernel(const ConvK p, const int mt, const int nt, const int P) {
  static_assert(!(MG && BN == 64), "x-phase merging: 256-row tiles");
  constexpr int XS = MG ? 2 : 1;
  constexpr int BM = (BN == 64) ? 512 : 256;
  constexpr int WGN = BN / 64, WGM = 8 / WGN;
  constexpr int TM = BM / WGM / 32, TN = 2;
  constexpr int ROWB = 128;
  constexpr int AXR = (BM == 512) ? 8 : 16;
  constexpr int AROWS = BM + AXR;
  constexpr int A_NI = AROWS / 8;
  constexpr int A_PASS = (A_NI + 7) / 8, B_PASS = BN / 64;
  constexpr int A_ST = AROWS * ROWB, B_ST = BN * ROWB;
  constexpr int B_OFF = 2 * A_ST;
  constexpr int OPS = 2 * (A_ST + B_ST);
  constexpr bool PF = BN != 64 && EARLY;                 // next tile's prologue under the epilogue (EARLY = false: the plain persistent walk)
  constexpr int NBUF = PF ? 2 : 1;
  constexpr int ARTAB = AROWS * (int)sizeof(ARow), TPTAB = MAXTAP * 8;
  constexpr int ROWS_OFF = OPS, AROW_OFF = ROWS_OFF + BM * (int)sizeof(RowB), TAPS_OFF = AROW_OFF + NBUF * ARTAB;
  constexpr int STAT_OFF = (TAPS_OFF + NBUF * TPTAB + 7) & ~7, STAT_N = 8;
  // epilogue transposition tiles (8704 bytes per wave): ring stages 1, then the LDS behind the statistics table
  constexpr int TSZ = 32 * (32 * TN + 4) * 4;
  constexpr int T_NA = PF ? (A_ST / TSZ < 8 ? A_ST / TSZ : 8) : 8, T_NB = PF ? ((B_ST / TSZ < 8 - T_NA) ? B_ST / TSZ : 8 - T_NA) : 0;
  constexpr int T_NF = 8 - T_NA - T_NB;
  constexpr int TFREE_OFF = (STAT_OFF + STAT_N * 16 + 15) & ~15;
  constexpr int LDS_TOTAL = TFREE_OFF + T_NF * TSZ;
  static_assert(8 * TSZ <= OPS && AROWS % 8 == 0, "epilogue tiles / DMA rows");
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char smem[LDS_TOTAL];
  RowB* rows = reinterpret_cast<RowB*>(smem + ROWS_OFF);
  const unsigned lds0 = (unsigned)(size_t)smem;

  // Everything derived from the thread index is RE-derived per tile from an opaque zero (tile_begin): the compiler would otherwise
  // keep ~20 loop-invariant lane constants alive across the tile loop, and at 256 VGPRs they end up in scratch — reloaded in the K
  // loop's rare paths, where a scratch load's vmcnt wait also drains the operand DMA queue (first version: 17.05 -> 19.8 ms).
  int tid = threadIdx.x;
  int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = mt * nt * P, G = (int)gridDim.x;
  int L = (int)blockIdx.x;
  if (L >= total) return;
  float* const out_g = p.out;
  const int gx = p.Gx;
  const int cpt = p.Ctot / 64;

  // a tile's geometry
  struct Tile { int bx, by, bz, m0, nb0, qx0, R0, npc; };
  auto decode = [&](int l) __attribute__((always_inline)) {
    Tile t;
    if (p.xcd_swizzle & 1) {          // (phase, N tile, M tile) order per XCD, as conv_bf16_big_kernel
      const int xcd = l & 7, j = l >> 3;
      t.bz = j % P;
      const int r = j / P;
      t.by = r % nt;
      t.bx = (r / nt) * 8 + xcd;
    } else {
      t.bx = l % mt;
      const int r = l / mt;
      t.by = r % nt;
      t.bz = r / nt;
    }
    t.m0 = t.bx * BM; t.nb0 = t.by * BN;
    t.qx0 = t.m0 % gx; t.R0 = t.m0 / gx;
    t.npc = (p.ntap[t.bz] >> 1) * cpt;
    return t;
  };
  // tables the DMA address code reads: taps (both column halves for MG), LDS row -> input pixel of the A ring
  auto build_dma_tables = [&](const Tile& t, int buf) __attribute__((always_inline)) {
    int* taps = reinterpret_cast<int*>(smem + TAPS_OFF + buf * TPTAB);
    const int phase = t.bz;
    if (tid < MAXTAP)
      taps[tid] = (p.dy[phase][tid] & 0xff) | ((p.dx[phase][tid] & 0xff) << 8) | ((int)p.wtap[phase][tid] << 16);
    if (MG && tid >= 64 && tid < 64 + MAXTAP) {
      const int q = tid - 64;
      taps[MAXTAP + q] = (p.dy[2 + phase][q] & 0xff) | ((p.dx[2 + phase][q] & 0xff) << 8) | ((int)p.wtap[2 + phase][q] << 16);
    }
    for (int rho = tid; rho < AROWS; rho += 512) {
      const int tt = rho + t.qx0;
      const int j = tt / (gx + XS), k = tt - j * (gx + XS);
      const int Rg = t.R0 + j;
      const int n = Rg / p.Gy, qy = Rg - n * p.Gy;
      ARow a;
      a.n = n < p.N ? n : -1; a.iy = (short)(qy * p.si); a.ix = (short)(k * p.si);
      reinterpret_cast<ARow*>(smem + AROW_OFF + buf * ARTAB)[rho] = a;
    }
  };
  // tables the epilogue reads: output rows, the workgroup's statistics table
  auto build_rows = [&](const Tile& t) __attribute__((always_inline)) {
    const int phase = t.bz;
    if (tid >= 64 && tid < 64 + STAT_N * 2) reinterpret_cast<double*>(smem + STAT_OFF)[tid - 64] = 0.0;
    if (tid < BM) {
      RowB ri;
      const int m = t.m0 + tid;
      ri.n = -1; ri.opix = 0; ri.iy = 0; ri.ix = 0; ri.oy = 0; ri.ox = 0;
      if (m < p.M) {
        const int gg = p.Gy * p.Gx;
        const int n = m / gg;
        const int rem = m - n * gg;
        const int qy = rem / p.Gx;
        const int qx = rem - qy * p.Gx;
        const int oy = qy * p.so + p.phy[phase];
        const int ox = qx * p.so + p.phx[phase];
        if (oy < p.Ho && ox < p.Wo) {
          ri.n = n; ri.iy = (short)(qy * p.si); ri.ix = (short)(qx * p.si); ri.oy = (short)oy; ri.ox = (short)ox;
          ri.opix = (n * p.Ho + oy) * p.Wo + ox;
          if (MG) ri.opix >>= 1;
        }
      }
      rows[tid] = ri;
    }
  };

  int wm0 = (wave / WGN) * (TM * 32);
  int wn0 = (wave % WGN) * (TN * 32);
  int l31 = lane & 31, lhi = lane >> 5;

  // ---- DMA loader state (as conv_bf16_pair_kernel; the tile it works on is `lt`, its tables are in buffer `lbuf`)
  const char* zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));
  const char* wp = uniform_ptr(reinterpret_cast<const char*>(p.W));
  int chunk = (tid & 7) ^ ((tid >> 4) & 7);
  unsigned pa[A_PASS], pb[B_PASS];
  unsigned pa_ok = 0, pb_ok = 0;
  const char* a_src = zero_pg;
  long wdelta = 0, wdelta1 = 0;
  int a_g = 0, a_ci = 0, a_left = 0;
  int b_g = 0, b_ci = 0, b_left = 0;
  int l_nb0 = 0;
  unsigned l_taps = lds0 + TAPS_OFF, l_arow = lds0 + AROW_OFF;      // LDS addresses of the loader's tables
  auto rebuild_a = [&]() __attribute__((always_inline)) {
    const int tp = lds_rd32_now(l_taps + (2 * a_g) * 4);
    const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
    const int cc = a_ci * 64;
    const char* sp = reinterpret_cast<const char*>(p.src[0].ptr);
    int sC = p.src[0].C, cs = 0;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.nsrc && cc >= p.cstart[q]) { sp = reinterpret_cast<const char*>(p.src[q].ptr); sC = p.src[q].C; cs = p.cstart[q]; }
    a_src = uniform_ptr(sp);
    const int cl = cc - cs + chunk * 8;
    pa_ok = 0;
    int rn[A_PASS], ryx[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int rho = min((tid >> 3) + 64 * i, AROWS - 1);
      const unsigned ra = l_arow + (unsigned)(rho * (int)sizeof(ARow));
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4" : "=&v"(rn[i]), "=&v"(ryx[i]) : "v"(ra));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int iy = (int)(short)(ryx[i] & 0xffff) + dyv, ix = (ryx[i] >> 16) + dxv;
      const bool ok = (rn[i] >= 0) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      pa[i] = ok ? ((unsigned)((rn[i] * p.Hi + iy) * p.Wi + ix) * (unsigned)sC + (unsigned)cl) * 2u : 0u;
      pa_ok |= (ok ? 1u : 0u) << i;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_a = [&]() __attribute__((always_inline)) {
    if (--a_left <= 0) return;
    if (++a_ci == cpt) { a_ci = 0; ++a_g; rebuild_a(); return; }
    bool src_edge = false;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && a_ci * 64 == p.cstart[q]) src_edge = true;
    if (src_edge) { rebuild_a(); return; }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) pa[i] += ROWB;
  };
  auto rebuild_b = [&]() __attribute__((always_inline)) {
    const int tp0 = lds_rd32_now(l_taps + (2 * b_g) * 4), tp1 = lds_rd32_now(l_taps + (2 * b_g + 1) * 4);
    const int base = __builtin_amdgcn_readfirstlane((tp0 >> 16) * p.wCout);
    wdelta = (long)__builtin_amdgcn_readfirstlane((tp1 >> 16) - (tp0 >> 16)) * p.wCout * p.wCin * 2;
    int base1 = 0;
    if constexpr (MG) {
      const int tq0 = lds_rd32_now(l_taps + (MAXTAP + 2 * b_g) * 4), tq1 = lds_rd32_now(l_taps + (MAXTAP + 2 * b_g + 1) * 4);
      base1 = __builtin_amdgcn_readfirstlane((tq0 >> 16) * p.wCout);
      wdelta1 = (long)__builtin_amdgcn_readfirstlane((tq1 >> 16) - (tq0 >> 16)) * p.wCout * p.wCin * 2;
    }
    pb_ok = 0;
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      const bool right = MG && 64 * i >= BN / 2;
      const int n = l_nb0 + (tid >> 3) + 64 * i - (right ? BN / 2 : 0);
      const bool ok = MG ? true : n < p.n_cnt;
      pb[i] = ok ? ((unsigned)((right ? base1 : base) + p.n_off + n) * (unsigned)p.wCin + (unsigned)(b_ci * 64 + chunk * 8)) * 2u : 0u;
      pb_ok |= (ok ? 1u : 0u) << i;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_b = [&]() __attribute__((always_inline)) {
    if (--b_left <= 0) return;
    if (++b_ci == cpt) { b_ci = 0; ++b_g; rebuild_b(); return; }
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) pb[i] += ROWB;
  };
  auto issue_a = [&](int stage) __attribute__((always_inline)) {
    float* const As = reinterpret_cast<float*>(smem + stage * A_ST);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
      if (i * 8 + wave < A_NI) {
        const char* src = ((pa_ok >> i) & 1u) ? a_src + pa[i] : zero_pg + (tid & 7) * 16;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), As + (i * 8 + wave) * 256, 16, 0, 0);
      }
  };
  auto issue_b = [&](int stage, int second) __attribute__((always_inline)) {
    float* const Bs = reinterpret_cast<float*>(smem + B_OFF + stage * B_ST);
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      const long wd = (MG && 64 * i >= BN / 2) ? wdelta1 : wdelta;
      const char* src = ((pb_ok >> i) & 1u) ? wp + (second ? wd : 0) + pb[i] : zero_pg + (tid & 7) * 16;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), Bs + (i * 8 + wave) * 256, 16, 0, 0);
    }
  };
  // a tile's prologue: loader state at its first (pair, chunk); ISSUE: its first A tile -> A stage 0, first B tile -> B stage 0.
  // Without ISSUE the same state is reached (after the epilogue that ran over the issued form's registers).
  auto start_loader = [&](const Tile& t, int buf, bool do_issue) __attribute__((always_inline)) {
    l_nb0 = t.nb0;
    l_taps = lds0 + TAPS_OFF + buf * TPTAB; l_arow = lds0 + AROW_OFF + buf * ARTAB;
    a_g = 0; a_ci = 0; a_left = t.npc;
    b_g = 0; b_ci = 0; b_left = t.npc;
    rebuild_a();
    if (do_issue) issue_a(0);
    advance_a();
    rebuild_b();
    if (do_issue) issue_b(0, 0);
  };

  constexpr bool BIGA = A_ST > 60000;
  int swr = (l31 >> 1) & 7;
  unsigned fb0 = lds0 + B_OFF + (unsigned)((wn0 + l31) * ROWB) + (unsigned)((lhi ^ swr) * 16);
  unsigned abase[BIGA ? 2 : 1][2][TM];
  auto fetch = [&](auto asg, auto jt, auto bsg, auto ksc, f32x4 (&va)[TM], f32x4 (&vb)[TN]) __attribute__((always_inline)) {
    constexpr int AS = decltype(asg)::value, JT = decltype(jt)::value, BSG = decltype(bsg)::value, KS = decltype(ksc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned t;
      asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                   : "=v"(va[i]), "=&v"(t) : "v"(abase[BIGA ? AS : 0][JT][i]), "n"(KS * 32), "n"(BIGA ? 0 : AS * A_ST));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      unsigned t;
      asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                   : "=v"(vb[j]), "=&v"(t) : "v"(fb0), "n"(KS * 32), "n"(BSG * B_ST + j * 32 * ROWB));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x16 acc[TM][TN];
  auto mfmas = [&](const f32x4 (&va)[TM], const f32x4 (&vb)[TN]) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va[i]), __builtin_bit_cast(bf16x8, vb[j]),
                                                             acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int NRD = TM + TN;
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;

  // this wave's transposition tile in the epilogue
  float* Tw = nullptr;
  auto tile_begin = [&]() __attribute__((always_inline)) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    tid = (int)threadIdx.x + z;
    lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wm0 = (wave / WGN) * (TM * 32); wn0 = (wave % WGN) * (TN * 32);
    l31 = lane & 31; lhi = lane >> 5;
    chunk = (tid & 7) ^ ((tid >> 4) & 7);
    swr = (l31 >> 1) & 7;
    fb0 = lds0 + B_OFF + (unsigned)((wn0 + l31) * ROWB) + (unsigned)((lhi ^ swr) * 16);
    zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros) + z);
    wp = uniform_ptr(reinterpret_cast<const char*>(p.W) + z);
    Tw = reinterpret_cast<float*>(smem + (!PF ? wave * TSZ
                                              : (wave < T_NA ? A_ST + wave * TSZ
                                                             : (wave < T_NA + T_NB ? B_OFF + B_ST + (wave - T_NA) * TSZ
                                                                                   : TFREE_OFF + (wave - T_NA - T_NB) * TSZ))));
  };
  tile_begin();

  Tile cur = decode(L);
  int buf = 0;
  build_dma_tables(cur, 0);
  build_rows(cur);
  __syncthreads();
  start_loader(cur, 0, true);

  for (;;) {
    const int npc = cur.npc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm0 + 32 * i + l31;
      const int rho = r + XS * ((cur.qx0 + r) / gx) + ((MG && wn0 >= BN / 2) ? 1 : 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = rho + j, sw_ = (row >> 1) & 7;
        abase[0][j][i] = lds0 + (unsigned)(row * ROWB) + (unsigned)((lhi ^ sw_) << 4);
        if constexpr (BIGA) abase[1][j][i] = abase[0][j][i] + (unsigned)A_ST;
      }
    }
    // the tile's first A / B tiles are in flight (issued by the prologue — for every tile but the first: under the previous epilogue)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue_b(1, 1);
    advance_b();
    if (npc > 1) { issue_a(1); advance_a(); }
    f32x4 va0[TM], vb0[TN], va1[TM], vb1[TN];
    fetch(I0{}, I0{}, I0{}, I0{}, va0, vb0);
    int pend_b = -1;
    auto step = [&](auto asg, int pc) __attribute__((always_inline)) {
      constexpr int AS = decltype(asg)::value;
      typedef std::integral_constant<int, AS> IA;
      typedef std::integral_constant<int, AS ^ 1> IN;
      const bool more = pc + 1 < npc;
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I0{}, I0{}, I1{}, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      if (pend_b == 1) { issue_b(1, 1); advance_b(); pend_b = -1; }
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I0{}, I0{}, I2{}, va0, vb0);
      PGB_LDS_WAIT(NRD);
      mfmas(va1, vb1);
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I0{}, I0{}, I3{}, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (more) pend_b = 0;
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I1{}, I1{}, I0{}, va0, vb0);
      mfmas(va1, vb1);
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I1{}, I1{}, I1{}, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      if (pend_b == 0) { issue_b(0, 0); pend_b = -1; }
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I1{}, I1{}, I2{}, va0, vb0);
      PGB_LDS_WAIT(NRD);
      mfmas(va1, vb1);
      fetch(IA{}, I1{}, I1{}, I3{}, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (more) pend_b = 1;
      if (pc + 2 < npc) { issue_a(AS); advance_a(); }
      __builtin_amdgcn_sched_barrier(0);
      if (more) fetch(IN{}, I0{}, I0{}, I0{}, va0, vb0);
      mfmas(va1, vb1);
    };
    for (int pc = 0; pc < npc; pc += 2) {
      step(I0{}, pc);
      if (pc + 1 < npc) step(I1{}, pc + 1);
    }

    // ---- epilogue; behind its first barrier (every wave is done with the rings) the next tile's prologue goes out
    const int Ln = L + G;
    const bool has_next = Ln < total;
    Tile nxt = cur;
    if (has_next) nxt = decode(Ln);
    big_epilogue<TM, TN, STAT_OFF, STAT_N>(p, acc, smem, rows, tid, cur.m0, cur.nb0, wm0, wn0, cur.bx, cur.by, cur.bz, 0, out_g, true, false,
                                           [](int) {}, Tw, [&]() __attribute__((always_inline)) {
                                             if (PF && has_next) {
                                               build_dma_tables(nxt, buf ^ 1);
                                               asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);      // (LDS tables only: __syncthreads would wait for vmcnt(0) too)
                                               start_loader(nxt, buf ^ 1, true);
                                             }
                                           });
    if (!has_next) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);      // every wave is done with rows[] and the statistics table (no vmcnt wait: the epilogue's stores stay in flight)
    tile_begin();
    L = Ln; cur = nxt;
    if (PF) {
      buf ^= 1;
      build_rows(cur);
      start_loader(cur, buf, false);       // the loader state the issued prologue left (its registers did not survive the epilogue)
      // (rows[] / statistics table: the K loop's first barrier, with its lgkmcnt(0), follows)
    } else {
      build_dma_tables(cur, 0);
      build_rows(cur);
      __syncthreads();
      start_loader(cur, 0, true);
    }
  }
}

#endif  // PG_TIMING_EXPERIMENTS

void launch_conv_bf16_pair(const ConvK& k_in, int bn, dim3 grid, hipStream_t st) {
  // (round 5) the persistent walk with the next tile's prologue under the epilogue (conv_bf16_pairp_kernel; timing builds only since
  // round 6): correct (the tap-pair and merged test cases passed through it with PG_PAIR_PERSIST=31) and OFF — slower in every variant on every layer measured
  // (tools/layer_bench.py, batch 32, one box; one tile per workgroup -> persistent):
  //   first version: lane constants alive across the tile loop spilled to scratch and were reloaded in the K loop's rare paths
  //   (rebuild_a / rebuild_b), where a scratch load's vmcnt wait also drains the operand DMA queue: north-star 17.05 -> 19.8 ms;
  //   second version (constants re-derived per tile from an opaque zero: no scratch in any K loop, none at all in the 128-wide
  //   variants; raw barriers so that no wait covers the epilogue's stores): forward launches + 3 ... 7 % (enc.1 218 -> 229 us, dec.5
  //   1027 -> 1083), data gradients + 11 ... 45 % (enc.1 287 -> 318, enc.2 182 -> 265, dec.4 985 -> 1134, dec.5 1106 -> 1405 us),
  //   north-star 17.0 -> 17.3 (128-wide variants only) ... 18.1 ms (all).  The 3.3 - 4.5 us prologue it hides are paid back by the
  //   second loader rebuild, three more barriers per tile and — in the 256-wide variants — by the scatter epilogue: with the next
  //   tile's state alive its register budget overflows, and every compiler-inserted scratch reload carries a vmcnt(0) that breaks the
  //   epilogue's counted waits (loads of the next half in flight behind the stores of this one).
  //   third experiment (PG_PAIR_PERSIST_NOPF: the plain persistent walk, the next tile's prologue AFTER the epilogue as a fresh
  //   workgroup would run it): still + 4 ... 45 % per launch (enc.2 data gradient 182 -> 263 us, dec.5 data gradient 1113 -> 1312),
  //   north-star 17.2 -> 18.0 - 18.3 ms.  So it is persistence itself, and the reason is the memory counter: loads and stores share ONE
  //   in-order vmcnt on gfx950, so the first wait for an operand tile of tile t + 1 also waits for the acknowledgement of every output
  //   store of tile t (256 KB per CU: ~12 us at the HBM rate) — a workgroup that ENDS leaves its stores in flight and its successor
  //   starts with fresh counters.  One tile per workgroup is the right shape on this part for store-heavy epilogues.
#ifdef PG_TIMING_EXPERIMENTS
  // PG_PAIR_PERSIST = bit mask of the variants that take the persistent form: 1 <128>, 2 <128, merged>, 4 <256>, 8 <256, merged>, 16 <64>
  static const int persist_mask = getenv("PG_PAIR_PERSIST") ? atoi(getenv("PG_PAIR_PERSIST")) : 0;
  const int vbit = bn == 1256 ? 8 : (bn == 1128 ? 2 : (bn == 256 ? 4 : (bn == 64 ? 16 : 1)));
  const bool persist = (persist_mask & vbit) != 0;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
    if (getenv("PG_PAIR_PERSIST_WGS")) ncu = atoi(getenv("PG_PAIR_PERSIST_WGS"));
  }
  if (persist) {
    const ConvK& k = k_in;
    const int mt = (int)grid.x, nt = (int)grid.y, P = (int)grid.z;
    const long total = (long)mt * nt * P;
    const dim3 g((unsigned)(total < ncu ? total : ncu));
    static const bool nopf = getenv("PG_PAIR_PERSIST_NOPF") != nullptr;      // the persistent walk WITHOUT the early prologue
    if (nopf) {
      if (bn == 1256) PG_KLAUNCH((conv_bf16_pairp_kernel<256, true, false>), g, dim3(512), 0, st, k, mt, nt, P);
      else if (bn == 1128) PG_KLAUNCH((conv_bf16_pairp_kernel<128, true, false>), g, dim3(512), 0, st, k, mt, nt, P);
      else if (bn == 256) PG_KLAUNCH((conv_bf16_pairp_kernel<256, false, false>), g, dim3(512), 0, st, k, mt, nt, P);
      else if (bn == 64) PG_KLAUNCH((conv_bf16_pairp_kernel<64>), g, dim3(512), 0, st, k, mt, nt, P);
      else PG_KLAUNCH((conv_bf16_pairp_kernel<128, false, false>), g, dim3(512), 0, st, k, mt, nt, P);
      return;
    }
    if (bn == 1256) PG_KLAUNCH((conv_bf16_pairp_kernel<256, true>), g, dim3(512), 0, st, k, mt, nt, P);
    else if (bn == 1128) PG_KLAUNCH((conv_bf16_pairp_kernel<128, true>), g, dim3(512), 0, st, k, mt, nt, P);
    else if (bn == 256) PG_KLAUNCH((conv_bf16_pairp_kernel<256>), g, dim3(512), 0, st, k, mt, nt, P);
    else if (bn == 64) PG_KLAUNCH((conv_bf16_pairp_kernel<64>), g, dim3(512), 0, st, k, mt, nt, P);
    else PG_KLAUNCH((conv_bf16_pairp_kernel<128>), g, dim3(512), 0, st, k, mt, nt, P);
    return;
  }
#endif  // PG_TIMING_EXPERIMENTS
  const ConvK& k = k_in;
  if (bn == 1256) PG_KLAUNCH((conv_bf16_pair_kernel<256, true>), grid, dim3(512), 0, st, k);       // x-phase merged: 256 x (2 x 128)
  else if (bn == 1128) PG_KLAUNCH((conv_bf16_pair_kernel<128, true>), grid, dim3(512), 0, st, k);  // 256 x (2 x 64)
  else if (bn == 256) PG_KLAUNCH((conv_bf16_pair_kernel<256>), grid, dim3(512), 0, st, k);
  else if (bn == 64) PG_KLAUNCH((conv_bf16_pair_kernel<64>), grid, dim3(512), 0, st, k);
  else PG_KLAUNCH((conv_bf16_pair_kernel<128>), grid, dim3(512), 0, st, k);
}

}  // namespace pg
