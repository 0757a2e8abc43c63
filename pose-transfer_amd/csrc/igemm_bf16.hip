// 256-row bf16 implicit-GEMM convolution for gfx950 — the large-layer kernel of the bf16 data path
// (BASELINE.json north_star: generator forward + backward at batch 32; configs[2] / configs[4]).
//
// Same contraction, operands and epilogues as conv_igemm_kernel<.., PREC = 3> (igemm_conv.hip): A = materialised bf16
// NHWC activations / gradients gathered per (tap, source) with a zero page for padding rows, B = K-contiguous bf16
// weights, fp32 accumulation, raw fp32 outputs (+ fused per-sample statistics) or the data-gradient scatter.
// What changes is the tile and the pipeline:
//   * workgroup tile 256 x BN (BN = 256 or 128; 512 x 64 for the N = 64 layers) x 64, 512 threads = 8 waves (2 per SIMD), ONE workgroup per CU; a wave
//     owns 128 x 64 (BN = 256: 4 x 2 tiles of v_mfma_f32_32x32x16_bf16, 128 accumulator registers) or 64 x 64: half
//     the LDS and global->LDS bytes per FLOP of the 128 x 128 kernel (24 ds_read_b128 per 32 MFMAs instead of 16 per 16);
//   * operands go global -> LDS with global_load_lds_dwordx4 (rows of 64 bf16 = 128 B, 16-byte chunks XOR-swizzled on the
//     SOURCE side, unpadded lane-linear LDS image), two stages of 64 KB (48 KB);
//   * ONE barrier per K tile, placed BEFORE the last k-step's MFMAs: a wave enters the barrier with 8 MFMAs queued and
//     the next tile's first operand fetch is issued right behind it, so neither the barrier nor the LDS latency of the
//     tile switch is exposed; operand registers are double-buffered per k-step (fetch k+1 | 8 MFMAs of k), every
//     s_waitcnt is counted (lgkmcnt(6) / vmcnt(0) only at the barrier, one tile after the DMA was issued);
//   * every LDS access of the K loop is inline asm and the barrier is the raw s_barrier: hipcc would otherwise drain
//     vmcnt(0) in front of each LDS read while a DMA is in flight (the DMA is a pending LDS write to its memory model).
#include <cstdlib>

#include "igemm_bf16_epi.h"

namespace pg {

// PG_DEBUG_CONV_TIMELINE: per-workgroup time stamps (s_memtime) of the LAST launch: 0 start, 1 rows/taps set up, 2 first tile landed,
// 3 K loop done, 4 epilogue done, 5 wall clock (100 MHz) at start, 6 wall clock at the end, 7 XCC id
constexpr int TL_WGS = 16384, TL_SLOTS = 16;      // 8..10: inside the epilogue (see the stamps)
static __device__ unsigned long long kTimeline[TL_WGS * TL_SLOTS];

// Build-time experiment (round 3): -DPG_BIG_3STAGE=1 gives the BN = 128 variant a third 48 KB stage (a DMA then has two K tiles
// to land).  Measured on dec.5 forward at batch 32: K loop 41.2 us per workgroup against 38.5 - 40.5 with two stages — the
// 128-wide variant is not waiting for its operands; off.
#ifndef PG_BIG_3STAGE
#define PG_BIG_3STAGE 0
#endif
// PG_BIG_SPLIT_DMA: a tile's DMA instructions go out in two halves — the A rows right behind the tile-switch barrier (as before),
// the B rows behind the next tile's first k-step — instead of all 8 per wave at the one moment when every wave of the CU is in
// the same state.  K loop in shader cycles (the clock moves with the load): dec.4 forward 223 k -> 209 k, dec.5 data gradient
// 115.6 k -> 112.4 k, dec.5 forward (128 wide) unchanged; three parts: no further gain.
#ifndef PG_BIG_SPLIT_DMA
#define PG_BIG_SPLIT_DMA 1
#endif
// BK (round 4): K elements per tile.  64 everywhere except the 512 x 128 tile of the N = 128 layers (the last decoder block's
// forward, encoder level 1's forward, encoder level 2's data gradient): 256 x 128 x 64 gives a wave 64 x 64 — 16 ds_read_b128 and
// 6 DMA instructions per 16 MFMAs, the worst ratios of the family (mfma_util 0.31) — and 512 x 128 x 64 does not fit two stages
// into 160 KB.  512 x 128 x 32: rows of 32 bf16 (64 B, four 16-byte chunks, swizzle chunk ^ (row >> 2 & 3)), a wave owns 128 x 64
// as in the 256-wide kernel (12 ds_read_b128 and 5 DMA instructions per 16 MFMAs), THREE stages of 40 KB (a DMA has two tiles of
// MFMA time to land; one barrier per tile = per 16 MFMAs of a wave, as before).
template <int BN, int BK>
__global__ __launch_bounds__(512, 2) void conv_bf16_big_kernel(const ConvK p) {
  constexpr int BM = (BN == 64 || BK == 32) ? 512 : 256;    // BN = 64 (N = 64 layers at full resolution): 512 x 64, a wave owns 64 x 64
  constexpr int WGN = BN / 64, WGM = 8 / WGN;            // 256: 2 x 4 waves, 128: 4 x 2 waves, 64: 8 x 1 waves
  constexpr int TM = BM / WGM / 32, TN = 2;              // MFMA tiles per wave: 4 x 2 or 2 x 2
  constexpr int ROWB = BK * 2, KCH = BK / 8, KS = BK / 16;   // bytes per operand row, 16-byte chunks per row, k-steps per tile
  constexpr int RPP = 512 / KCH;                         // rows one DMA pass of the workgroup covers (512 lanes x 16 B)
  constexpr int A_PASS = BM / RPP, B_PASS = BN / RPP;    // global_load_lds per thread and tile
  constexpr int A_ST = BM * ROWB, B_ST = BN * ROWB;      // bytes per stage
  constexpr int STAGE = A_ST + B_ST;
  constexpr int NST = (BK == 32 || (BN == 128 && PG_BIG_3STAGE)) ? 3 : 2;
  constexpr int ROWS_OFF = NST * STAGE, TAPS_OFF = ROWS_OFF + BM * (int)sizeof(RowB);
  constexpr int STAT_OFF = (TAPS_OFF + MAXTAP * 4 + 7) & ~7, STAT_N = 8;         // per-workgroup statistics: STAT_N samples x (sum, sum of squares)
  static_assert(B_PASS >= 1 && 8 * (32 * (32 * TN + 4)) * 4 <= NST * STAGE, "tile / epilogue buffers");
  __shared__ __attribute__((aligned(1024))) char smem[STAT_OFF + STAT_N * 2 * 8];     // ONE LDS object (see header)
  RowB* rows = reinterpret_cast<RowB*>(smem + ROWS_OFF);
  int* taps_l = reinterpret_cast<int*>(smem + TAPS_OFF);
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool tl_on = PG_DBG(p, 16);
  const int tl_id = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  auto stamp = [&](int slot) {
    if (tl_on && tid == 0 && tl_id < TL_WGS) {
      kTimeline[tl_id * TL_SLOTS + slot] = __builtin_amdgcn_s_memtime();
      if (slot == 0) {
        kTimeline[tl_id * TL_SLOTS + 5] = wall_clock64();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        kTimeline[tl_id * TL_SLOTS + 7] = xcc & 0xf;
      }
      if (slot == 4) kTimeline[tl_id * TL_SLOTS + 6] = wall_clock64();
    }
  };
  stamp(0);
  // Workgroups are dealt round-robin to the 8 XCDs (one 4 MB L2 each) in dispatch order.  Remapped, the j-th workgroup an
  // XCD receives walks (sub-pixel phase fastest, then N tile, then M tile): the four phases of a transposed convolution /
  // conv data-gradient read the same input pixels through different taps, the N tiles of an M tile the same activation
  // rows — both are then served from that XCD's L2 instead of the fabric (MI355X_MICROARCH.md: per-XCD L2s).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_swizzle & 1) {          // host: gridDim.x % 8 == 0, ksplit == 1, no tap batch
    const int mt = (int)gridDim.x, nt = (int)gridDim.y, P = (int)gridDim.z;
    const int L = bx + mt * (by + nt * bz);
    const int xcd = L & 7, j = L >> 3;
    bz = j % P;
    const int r = j / P;
    by = r % nt;
    bx = (r / nt) * 8 + xcd;
  }
  const int zphase = bz / p.ksplit;
  const int split = bz - zphase * p.ksplit;
  const int phase = p.gtaps ? 0 : zphase;
  const long a_off_g = p.gtaps ? p.a_off[zphase] : 0, w_off_g = p.gtaps ? p.w_off[zphase] : 0;
  float* const out_g = p.out + (p.gtaps ? p.o_off[zphase] : 0);
  const int m0 = bx * BM, nb0 = by * BN;
  const int ntap = p.ntap[phase];

  if (tid >= 64 && tid < 64 + STAT_N * 2) reinterpret_cast<double*>(smem + STAT_OFF)[tid - 64] = 0.0;
  if (tid < MAXTAP)
    taps_l[tid] = (p.dy[phase][tid] & 0xff) | ((p.dx[phase][tid] & 0xff) << 8) | ((int)p.wtap[phase][tid] << 16);
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
      }
    }
    rows[tid] = ri;
  }
  __syncthreads();

  stamp(1);
  const int cpt = p.Ctot / BK;                      // K tiles per tap
  const int ktot = ntap * cpt;
  const int kper = (ktot + p.ksplit - 1) / p.ksplit;
  const int kt0 = split * kper;
  const int kt1 = PG_DBG(p, 8) ? min(ktot, kt0 + 1) : min(ktot, kt0 + kper);      // bit 3: PG_DEBUG_ONE_KTILE (fixed-cost experiment)
  if (kt0 >= kt1) return;

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

  // ---- DMA loader state: per-thread source pointers of this thread's A_PASS + B_PASS rows (chunk = slot ^ row swizzle)
  const char* const zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));
  const char* const wp = uniform_ptr(reinterpret_cast<const char*>(p.W) + w_off_g);
  const int chunk = (BK == 64) ? ((tid & 7) ^ ((tid >> 4) & 7)) : ((tid & 3) ^ ((tid >> 4) & 3));      // source chunk of this lane's LDS slot
  const char* pa[A_PASS];
  const char* pb[B_PASS];
  int ld_kt = kt0, ld_tap = kt0 / cpt, ld_ci = kt0 - (kt0 / cpt) * cpt;
  auto rebuild = [&]() {
    const int tp = lds_rd32_now(lds0 + TAPS_OFF + ld_tap * 4);
    const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
    const int cc = ld_ci * BK;
    const char* sp = reinterpret_cast<const char*>(p.src[0].ptr);
    int sC = p.src[0].C, cs = 0;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.nsrc && cc >= p.cstart[q]) { sp = reinterpret_cast<const char*>(p.src[q].ptr); sC = p.src[q].C; cs = p.cstart[q]; }
    sp = uniform_ptr(sp + a_off_g);
    const int cl = cc - cs + chunk * 8;
    // the rows' (sample, base coordinate) are re-read from LDS here (rare path) instead of living in 12 registers
    int rn[A_PASS], ryx[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const unsigned ra = lds0 + ROWS_OFF + (unsigned)((tid / KCH + RPP * i) * (int)sizeof(RowB));
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:8" : "=&v"(rn[i]), "=&v"(ryx[i]) : "v"(ra));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int iy = (int)(short)(ryx[i] & 0xffff) + dyv, ix = (ryx[i] >> 16) + dxv;
      const bool ok = (rn[i] >= 0) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      const long off = ((long)((rn[i] * p.Hi + iy) * p.Wi + ix) * sC + cl) * 2;
      pa[i] = ok ? sp + off : zero_pg + (tid & (KCH - 1)) * 16;
      if (PG_DBG(p, 2)) pa[i] = sp + (long)cl * 2;        // PG_DEBUG_OPERAND_A: every row reads pixel 0 (delivery experiment)
    }
    const int base = (tp >> 16) * p.wCout;
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      const int n = nb0 + tid / KCH + RPP * i;
      const long off = ((long)(base + p.n_off + n) * p.wCin + cc + chunk * 8) * 2;
      pb[i] = (n < p.n_cnt) ? wp + off : zero_pg + (tid & (KCH - 1)) * 16;
      if (PG_DBG(p, 4)) pb[i] = wp + ((long)(base + p.n_off) * p.wCin + cc + chunk * 8) * 2;      // PG_DEBUG_OPERAND_B
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): no scalar (kernel-argument) load stays in flight past here
  };
  auto advance = [&]() {
    if (ld_kt + 1 < kt1) {
      ++ld_kt;
      if (++ld_ci == cpt) { ld_ci = 0; ++ld_tap; rebuild(); }
      else {
        bool src_edge = false;
#pragma unroll
        for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && ld_ci * BK == p.cstart[q]) src_edge = true;
        if (src_edge) rebuild();
        else {
#pragma unroll
          for (int i = 0; i < A_PASS; ++i) pa[i] += ROWB;
#pragma unroll
          for (int i = 0; i < B_PASS; ++i) pb[i] += ROWB;
        }
      }
    }
  };
  auto issue = [&](int stage) {
    float* const As = reinterpret_cast<float*>(smem + stage * STAGE);
    float* const Bs = reinterpret_cast<float*>(smem + stage * STAGE + A_ST);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(pa[i]), As + (i * 8 + wave) * 256, 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PASS; ++i)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(pb[i]), Bs + (i * 8 + wave) * 256, 16, 0, 0);
  };

  // the same in two halves (PG_BIG_SPLIT_DMA): A rows / B rows of the tile
  auto issue_a = [&](int stage) {
    float* const As = reinterpret_cast<float*>(smem + stage * STAGE);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(pa[i]), As + (i * 8 + wave) * 256, 16, 0, 0);
  };
  auto issue_b = [&](int stage) {
    float* const Bs = reinterpret_cast<float*>(smem + stage * STAGE + A_ST);
#pragma unroll
    for (int i = 0; i < B_PASS; ++i)
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(pb[i]), Bs + (i * 8 + wave) * 256, 16, 0, 0);
  };

  // ---- operand fetch: k-step ks (16 k's) of a stage -> one register set (TM + TN ds_read_b128)
  const int swr = (BK == 64) ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
  unsigned fa[KS], fb[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    fa[ks] = lds0 + (unsigned)((wm0 + l31) * ROWB) + (unsigned)(((2 * ks + lhi) ^ swr) * 16);
    fb[ks] = lds0 + A_ST + (unsigned)((wn0 + l31) * ROWB) + (unsigned)(((2 * ks + lhi) ^ swr) * 16);
  }
  int dbg_sink = 0; (void)dbg_sink;
  auto fetch = [&](int stage, int ks, f32x4 (&va)[TM], f32x4 (&vb)[TN]) {
    const unsigned aa = fa[ks] + (unsigned)(stage * STAGE), bb = fb[ks] + (unsigned)(stage * STAGE);
#ifdef PG_TIMING_EXPERIMENTS
    // round 4 (results are wrong, times are not): PG_DEBUG_NO_FETCH = no operand reads at all, PG_DEBUG_HALF_A_FETCH = two of the
    // four A fragments (the LDS read volume a 128 x 128 wave tile would have: 4 instead of 6 reads per 8 MFMAs)
    if (PG_DBG(p, 8192)) return;
    if (PG_DBG(p, 4096)) {
      lds_rd128<0>(va[0], aa);
      lds_rd128<32 * ROWB>(va[1], aa);
      lds_rd128<0>(vb[0], bb);
      lds_rd128<32 * ROWB>(vb[1], bb);
      asm volatile("s_nop 0\n\ts_nop 0" ::: "memory");
      asm volatile("ds_read_b32 %0, %1" : "=v"(dbg_sink) : "v"(aa));      // keep the lgkmcnt accounting (6 reads per fetch)
      asm volatile("ds_read_b32 %0, %1" : "=v"(dbg_sink) : "v"(bb));
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
#endif
    lds_rd128<0>(va[0], aa);
    lds_rd128<32 * ROWB>(va[1], aa);
    if constexpr (TM == 4) { lds_rd128<64 * ROWB>(va[2], aa); lds_rd128<96 * ROWB>(va[3], aa); }
    lds_rd128<0>(vb[0], bb);
    lds_rd128<32 * ROWB>(vb[1], bb);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfmas = [&](const f32x4 (&va)[TM], const f32x4 (&vb)[TN]) {
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

  // ---- prologue: tile kt0 -> stage 0
  rebuild();
  issue(0);
  advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  stamp(2);
  f32x4 va0[TM], vb0[TN], va1[TM], vb1[TN];
  fetch(0, 0, va0, vb0);
  int stage = 0;
  // tile kt0 + 1 is issued before the loop; inside, the DMA of tile kt + 2 is issued RIGHT AFTER the tile-switch barrier of
  // tile kt (its stage was released by that barrier) and before the last 8 MFMAs of tile kt: a full tile of MFMA time
  // (2048 SIMD cycles with two waves per SIMD) lies between a DMA's issue and the vmcnt(0) that waits for it
  // (PMC, round 2: with the issue after those MFMAs the waves were parked at the wait for 37 % of their cycles).
  if (kt0 + 1 < kt1) { issue(1); advance(); }
  if constexpr (NST == 3) {
    if (kt0 + 2 < kt1) { issue(2); advance(); }
  }
  bool pend = false;                 // PG_BIG_SPLIT_DMA: the B half of the tile issued at the last switch is still to be issued
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    __builtin_amdgcn_sched_barrier(0);
    // k-steps 0 .. KS-2: fetch the next k-step's operands into the other register set, multiply the current one
    fetch(stage, 1, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    if constexpr (PG_BIG_SPLIT_DMA && NST == 2) {
      if (pend) { issue_b(stage ^ 1); advance(); pend = false; }      // the B half behind the first k-step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (KS == 4) {
      fetch(stage, 2, va0, vb0);
      PGB_LDS_WAIT(NRD);
      mfmas(va1, vb1);

      fetch(stage, 3, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
    }
    // tile switch: this wave's last operand fetch of the stage has landed (lgkmcnt(0)), its share of the next tile has
    // landed (two stages: vmcnt(0); three: the DMA instructions of tile kt + 2, issued later, may stay in flight); after the
    // barrier both hold for every wave
    if constexpr (NST == 3) {
      if (kt + 2 < kt1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A_PASS + B_PASS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if (!PG_DBG(p, 256)) __builtin_amdgcn_s_barrier();      // bit 8: PG_DEBUG_NO_KBARRIER (timing experiment, wrong results)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NST < kt1 && !PG_DBG(p, 512)) {        // tile kt + NST into the stage this tile just released (bit 9: PG_DEBUG_NO_KDMA)
      if constexpr (PG_BIG_SPLIT_DMA && NST == 2) {
        if (!(PG_DBG(p, 1024) && (kt & 3)) && !(PG_DBG(p, 16384) && (kt & 1))) issue_a(stage);      // bit 14: PG_DEBUG_A_EVERY_2ND (what tap-pair sharing of the A tile would issue)      // bit 10: PG_DEBUG_A_EVERY_4TH (timing experiment: A rows on one K tile in four)
        pend = true;
      }
      else { issue(stage); advance(); }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int nstage = (NST == 3) ? (stage == 2 ? 0 : stage + 1) : (stage ^ 1);
    if (more) fetch(nstage, 0, va0, vb0);
    mfmas(va1, vb1);
    stage = nstage;
  }

  // ------------------------------------------------------------------ epilogue (operand stages are free after a barrier)
  big_epilogue<TM, TN, STAT_OFF, STAT_N>(p, acc, smem, rows, tid, m0, nb0, wm0, wn0, bx, by, bz, split, out_g, false, tl_on, stamp);
}

}  // namespace pg
// debugging aid: copies the time stamps of the last PG_DEBUG_CONV_TIMELINE launch (n_wgs x 8 uint64) to host memory
extern "C" int pg_debug_conv_timeline(unsigned long long* host_out, int32_t n_wgs) {
  if (host_out == nullptr || n_wgs <= 0 || n_wgs > pg::TL_WGS) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pg::kTimeline), sizeof(unsigned long long) * pg::TL_SLOTS * (size_t)n_wgs, 0,
                             hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
namespace pg {
// launch helper used by conv_impl (igemm_conv.hip)
void launch_conv_bf16_big(const ConvK& k, int bn, dim3 grid, hipStream_t st) {
  if (bn == 256) PG_KLAUNCH((conv_bf16_big_kernel<256, 64>), grid, dim3(512), 0, st, k);
  else if (bn == 64) PG_KLAUNCH((conv_bf16_big_kernel<64, 64>), grid, dim3(512), 0, st, k);
  else if (bn == 129) PG_KLAUNCH((conv_bf16_big_kernel<128, 32>), grid, dim3(512), 0, st, k);      // 512 x 128 x 32
  else PG_KLAUNCH((conv_bf16_big_kernel<128, 64>), grid, dim3(512), 0, st, k);
}

}  // namespace pg
