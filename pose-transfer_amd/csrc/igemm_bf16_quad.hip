// 512 x 128 x 32 bf16 implicit-GEMM convolution with TAP-QUAD SHARING of the A operand (round 6).
//
// The N = 128 / short-K layers (encoder level 1 forward: Cin 64 -> Cout 128; its data gradient: two x-merged phases of 64 columns
// over K = 4 taps x 128) move 6 - 8 x the operand bytes per FLOP of the large launches through the global -> LDS path (DESIGN.md
// section 3.5): the 256 x 128 tap-pair tile re-fetches every input pixel once per tap PAIR and the 16 KB weight tile once per
// 256 rows.  This kernel cuts both:
//   * A: the 16 taps of a k4 s2 p1 convolution are 4 QUADS (dy0 + {0, si}) x (dx0 + {0, si}); the four taps of a quad read the
//     same input pixels shifted by one output-grid step along x and / or y: tap (dy0 + jy si, dx0 + jx si) at position (qy, qx) = tap
//     (dy0, dx0) at (qy + jy, qx + jx).  The four quads are the four parity classes of the input patch, so the union of their A tiles
//     is the (2 Th + 2) x (2 Tw + 2) halo patch of the tile, every pixel of it fetched ONCE (review item 1 of round 5: "spatial-tile
//     A staging", here streamed one parity class at a time).  As in the tap-pair kernel an A tile is a run of the linearised
//     (image row, slot) space with Gx + XS slots per image row: LDS row rho holds slot (t0 + rho) for the quad's first tap, and tap
//     (jy, jx) of tile row r reads LDS row rho(r) + jx + jy (Gx + XS).  The tile (512 consecutive output-grid positions) must lie inside
//     one sample (host: Gy Gx % 512 == 0), so "the next image row" is always the same sample's.
//   * B: 512 rows per workgroup halve the weight-tile fetches per output row of the 256-row kernels.
//   * the x-phase merged transposed form (MG, as in igemm_bf16_pair.hip): XS = 2, the right column half (phase (py, 1)) reads one slot
//     further and takes its weights from the second tap table; the 2 x 2 taps of a phase are ONE quad, so an A tile per channel chunk
//     serves the whole K loop of the chunk.
// K tiles are 32 channels (rows of 64 B, four 16-byte chunks, swizzle chunk ^ (row >> 2 & 3)) as in conv_bf16_big_kernel<128, 32>:
// 2 A stages of 42 KB + a ring of 4 B tiles of 8 KB.  One (quad, chunk) STEP = 4 tiles (one per tap, B stage = tap index); one barrier
// per tile; the DMA of a tile's B is issued 4 tiles ahead and the next step's A tile in three pieces spread over the current step, so
// every wait is a COUNTED vmcnt (a DMA has 2 - 4 tiles of MFMA time to land).  8 waves, wave tile 128 x 64, the shared epilogues.
#include "igemm_bf16_epi.h"

namespace pg {

#ifdef PG_TIMING_EXPERIMENTS
// per-workgroup time stamps of the LAST launch (tools/conv_timeline.py; slots as in igemm_bf16.hip)
constexpr int QTL_WGS = 16384, QTL_SLOTS = 16;
static __device__ unsigned long long kTimelineQ[QTL_WGS * QTL_SLOTS];
#endif

struct QRow {      // LDS row of the A ring -> the input pixel it holds, before the quad's tap offset
  short iy, ix;    // qy * si, slot * si   (qy may be Gy, slot may be >= Gx: the range check of the DMA source decides)
};

template <bool MG>
__global__ __launch_bounds__(512, 2) void conv_bf16_quad_kernel(const ConvK p) {
  constexpr int XS = MG ? 2 : 1;
  constexpr int BM = 512, BN = 128, TM = 4, TN = 2, WGN = 2;
  constexpr int ROWB = 64;
  constexpr int AROWS = 672;                              // 512 + image-row starts + one image row (Gx + XS <= 130) + the taps' slots
  constexpr int A_NI = AROWS / 16;                        // wave DMA instructions per A tile (16 rows of 64 B each): 42
  constexpr int A_PASS = (A_NI + 7) / 8;                  // 6 (the last: waves 0, 1 only)
  constexpr int A_ST = AROWS * ROWB, B_ST = BN * ROWB, NBS = 4;
  constexpr int B_OFF = 2 * A_ST;                         // [A ring: 2 stages][B ring: 4 stages]
  constexpr int OPS = B_OFF + NBS * B_ST;
  constexpr int ROWS_OFF = OPS, AROW_OFF = ROWS_OFF + BM * (int)sizeof(RowB), TAPS_OFF = AROW_OFF + AROWS * (int)sizeof(QRow);
  constexpr int STAT_OFF = (TAPS_OFF + MAXTAP * 8 + 7) & ~7, STAT_N = 8;
  static_assert(8 * (32 * (32 * TN + 4)) * 4 <= OPS && A_PASS == 6 && A_NI > 40, "epilogue tiles / DMA pieces");
  static_assert(STAT_OFF + STAT_N * 16 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) char smem[STAT_OFF + STAT_N * 2 * 8];
  RowB* rows = reinterpret_cast<RowB*>(smem + ROWS_OFF);
  int* taps_l = reinterpret_cast<int*>(smem + TAPS_OFF);
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
#ifdef PG_TIMING_EXPERIMENTS
  const bool tl_on = PG_DBG(p, 16);
  const int tl_id = (int)(blockIdx.x + gridDim.x * blockIdx.z);
  auto stamp = [&](int slot) {
    if (tl_on && tid == 0 && tl_id < QTL_WGS) {
      kTimelineQ[tl_id * QTL_SLOTS + slot] = __builtin_amdgcn_s_memtime();
      if (slot == 0) {
        kTimelineQ[tl_id * QTL_SLOTS + 5] = wall_clock64();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        kTimelineQ[tl_id * QTL_SLOTS + 7] = xcc & 0xf;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        kTimelineQ[tl_id * QTL_SLOTS + 10] = hwid;
      }
      if (slot == 4) kTimelineQ[tl_id * QTL_SLOTS + 6] = wall_clock64();
    }
  };
#else
  const bool tl_on = false;
  auto stamp = [](int) {};
#endif
  stamp(0);
  int bx = blockIdx.x, by = 0, bz = blockIdx.z;
  if (p.xcd_swizzle & 1) {          // (phase, M tile) order per XCD, as conv_bf16_big_kernel
    const int mt = (int)gridDim.x, P = (int)gridDim.z;
    const int L = bx + mt * bz;
    const int xcd = L & 7, j = L >> 3;
    bz = j % P;
    bx = (j / P) * 8 + xcd;
  }
  const int phase = bz;
  float* const out_g = p.out;
  const int m0 = bx * BM, nb0 = 0;
  const int ntap = p.ntap[phase];
  const int gx = p.Gx, gg = p.Gy * p.Gx;
  const int n_s = m0 / gg;                          // the tile's sample (host: gg % BM == 0)
  const int rem0 = m0 - n_s * gg;
  const int qy0 = rem0 / gx, qx0 = rem0 - qy0 * gx;
  const int IR = gx + XS;                           // LDS rows per image row

  if (tid >= 64 && tid < 64 + STAT_N * 2) reinterpret_cast<double*>(smem + STAT_OFF)[tid - 64] = 0.0;
  if (tid < MAXTAP)
    taps_l[tid] = (p.dy[phase][tid] & 0xff) | ((p.dx[phase][tid] & 0xff) << 8) | ((int)p.wtap[phase][tid] << 16);
  if (MG && tid >= 64 && tid < 64 + MAXTAP) {           // the right half's taps (phase (py, 1)): host keeps them in slot 2 + py
    const int q = tid - 64;
    taps_l[MAXTAP + q] = (p.dy[2 + phase][q] & 0xff) | ((p.dx[2 + phase][q] & 0xff) << 8) | ((int)p.wtap[2 + phase][q] << 16);
  }
  {
    RowB ri;
    const int m = m0 + tid;
    ri.n = -1; ri.opix = 0; ri.iy = 0; ri.ix = 0; ri.oy = 0; ri.ox = 0;
    if (m < p.M) {
      const int rem = rem0 + tid;
      const int qy = rem / gx;
      const int qx = rem - qy * gx;
      const int oy = qy * p.so + p.phy[phase];
      const int ox = qx * p.so + p.phx[phase];
      if (oy < p.Ho && ox < p.Wo) {
        ri.n = n_s; ri.iy = (short)(qy * p.si); ri.ix = (short)(qx * p.si); ri.oy = (short)oy; ri.ox = (short)ox;
        ri.opix = (n_s * p.Ho + oy) * p.Wo + ox;
        if (MG) ri.opix >>= 1;                            // pixel PAIR (ox = 2 qx, Wo even): the epilogues' rows are 2 Cout wide
      }
    }
    rows[tid] = ri;
  }
  for (int rho = tid; rho < AROWS; rho += 512) {          // LDS row -> (local image row j, slot k): rho + qx0 = j (Gx + XS) + k
    const int t = rho + qx0;
    const int j = t / IR, k = t - j * IR;
    QRow a;
    a.iy = (short)((qy0 + j) * p.si); a.ix = (short)(k * p.si);
    reinterpret_cast<QRow*>(smem + AROW_OFF)[rho] = a;
  }
  __syncthreads();
  stamp(1);

  const int cpt = p.Ctot / 32;                      // channel chunks per quad
  const int nsteps = PG_DBG(p, 8) ? 1 : (ntap >> 2) * cpt;      // (quad, chunk) steps; four tiles each  (bit 3: PG_DEBUG_ONE_KTILE, fixed-cost experiment)

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
  const int chunk = (tid & 3) ^ ((tid >> 4) & 3);   // source chunk of this lane's LDS slot (swizzle by LDS row; a DMA block is 16 rows)
  unsigned pa[A_PASS];                              // per-row source: 32-bit byte offset against the (uniform) source tensor
  unsigned pa_ok = 0;
  unsigned pb = 0;
  const char* a_src = zero_pg;
  long tapoff[4] = {0, 0, 0, 0};                    // uniform: byte offset of the B cursor's quad's taps in W (this wave's column half)
  int a_q = 0, a_ci = 0, a_left = nsteps;           // A cursor: next (quad, chunk) to load
  int b_q = 0, b_ci = 0, b_left = nsteps;           // B cursor
  auto rebuild_a = [&]() __attribute__((always_inline)) {
    const int tp = lds_rd32_now(lds0 + TAPS_OFF + (4 * a_q) * 4);
    const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
    const int cc = a_ci * 32;
    const char* sp = reinterpret_cast<const char*>(p.src[0].ptr);
    int sC = p.src[0].C, cs = 0;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.nsrc && cc >= p.cstart[q]) { sp = reinterpret_cast<const char*>(p.src[q].ptr); sC = p.src[q].C; cs = p.cstart[q]; }
    a_src = uniform_ptr(sp);
    const int cl = cc - cs + chunk * 8;
    pa_ok = 0;
    int ryx[A_PASS];
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int rho = min((i * 8 + wave) * 16 + (lane >> 2), AROWS - 1);
      const unsigned ra = lds0 + AROW_OFF + (unsigned)(rho * (int)sizeof(QRow));
      asm volatile("ds_read_b32 %0, %1" : "=&v"(ryx[i]) : "v"(ra));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int iy = (int)(short)(ryx[i] & 0xffff) + dyv, ix = (ryx[i] >> 16) + dxv;
      const bool ok = (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      pa[i] = ok ? ((unsigned)((n_s * p.Hi + iy) * p.Wi + ix) * (unsigned)sC + (unsigned)cl) * 2u : 0u;
      pa_ok |= (ok ? 1u : 0u) << i;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_a = [&]() __attribute__((always_inline)) {            // after a step's A tile was issued completely
    if (--a_left <= 0) return;
    if (++a_ci == cpt) { a_ci = 0; ++a_q; rebuild_a(); return; }
    bool src_edge = false;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && a_ci * 32 == p.cstart[q]) src_edge = true;
    if (src_edge) { rebuild_a(); return; }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) pa[i] += ROWB;       // (invalid rows: the offset is not used)
  };
  const bool right = MG && wave >= 4;                    // B rows 64 .. 127 = phase (py, 1)'s weights: waves 4 - 7 load them
  auto rebuild_b = [&]() __attribute__((always_inline)) {
    const unsigned tb = lds0 + TAPS_OFF + (unsigned)(((right ? MAXTAP : 0) + 4 * b_q) * 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tp = lds_rd32_now(tb + t * 4);
      tapoff[t] = (long)__builtin_amdgcn_readfirstlane(tp >> 16) * p.wCout * p.wCin * 2;
    }
    const int n = (tid >> 2) - (right ? 64 : 0);        // (host: n_cnt == 128, MG: Cout == 64 — every row valid)
    pb = ((unsigned)(p.n_off + n) * (unsigned)p.wCin + (unsigned)(b_ci * 32 + chunk * 8)) * 2u;
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_b = [&]() __attribute__((always_inline)) {            // after the four B tiles of a step were issued
    if (--b_left <= 0) return;
    if (++b_ci == cpt) { b_ci = 0; ++b_q; rebuild_b(); return; }
    pb += ROWB;
  };
  // A tile pieces: instruction i of this wave covers LDS rows (i * 8 + wave) * 16 ... + 15
  auto issue_a1 = [&](int stage, int i) __attribute__((always_inline)) {
    float* const As = reinterpret_cast<float*>(smem + stage * A_ST);
    const char* src = ((pa_ok >> i) & 1u) ? a_src + pa[i] : zero_pg + (tid & 3) * 16;
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), As + (i * 8 + wave) * 256, 16, 0, 0);
  };
  auto issue_a_p0 = [&](int stage) __attribute__((always_inline)) {
    if (5 * 8 + wave < A_NI) issue_a1(stage, 5);      // wave-uniform: only waves 0, 1 (the counted waits assume >= 1 instruction here)
    issue_a1(stage, 0);
  };
  auto issue_a_p1 = [&](int stage) __attribute__((always_inline)) { issue_a1(stage, 1); issue_a1(stage, 2); };
  auto issue_a_p2 = [&](int stage) __attribute__((always_inline)) { issue_a1(stage, 3); issue_a1(stage, 4); };
  auto issue_b = [&](int t) __attribute__((always_inline)) {       // tap t of the B cursor's step -> B stage t
    float* const Bs = reinterpret_cast<float*>(smem + B_OFF + t * B_ST);
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wp + tapoff[t] + pb), Bs + wave * 256, 16, 0, 0);
  };

  // ---- operand fetch.  A: LDS row of tile row r for tap (jy, jx): rho(r) + jx + jy IR, rho(r) = r + XS ((qx0 + r) / Gx) (MG, right
  // half: one slot further).  The 16-byte slot of k-step ks and lane half lhi is (2 ks + lhi) ^ swizzle(LDS row): one address per
  // (tap, M tile) for ks = 0, address(1) = address(0) ^ 32 (the xor sits in the asm statement: see igemm_bf16_pair.hip).
  unsigned abase[4][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm0 + 32 * i + l31;
    const int rho = r + XS * ((qx0 + r) / gx) + ((MG && wn0 >= BN / 2) ? 1 : 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = rho + (t & 1) + (t >> 1) * IR, s = (row >> 2) & 3;
      abase[t][i] = lds0 + (unsigned)(row * ROWB) + (unsigned)((lhi ^ s) << 4);
    }
  }
  const int swr = (l31 >> 2) & 3;
  const unsigned fb0 = lds0 + B_OFF + (unsigned)((wn0 + l31) * ROWB) + (unsigned)((lhi ^ swr) * 16);
  // AS: A ring stage, TT: tap of the quad = B ring stage, KS: k-step — all compile-time
  auto fetch = [&](auto asg, auto tt, auto ksc, f32x4 (&va)[TM], f32x4 (&vb)[TN]) __attribute__((always_inline)) {
    constexpr int AS = decltype(asg)::value, TT = decltype(tt)::value, KS = decltype(ksc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (KS == 0) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(va[i]) : "v"(abase[TT][i]), "n"(AS * A_ST));
      } else {
        unsigned t;
        asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                     : "=v"(va[i]), "=&v"(t) : "v"(abase[TT][i]), "n"(KS * 32), "n"(AS * A_ST));
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (KS == 0) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vb[j]) : "v"(fb0), "n"(TT * B_ST + j * 32 * ROWB));
      } else {
        unsigned t;
        asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                     : "=v"(vb[j]), "=&v"(t) : "v"(fb0), "n"(KS * 32), "n"(TT * B_ST + j * 32 * ROWB));
      }
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

  // ---- prologue: step 0's A tile -> A stage 0, its four B tiles -> the B ring; then the first piece of step 1's A tile
  rebuild_a();
#pragma unroll
  for (int i = 0; i < A_PASS; ++i)
    if (i * 8 + wave < A_NI) issue_a1(0, i);
  advance_a();
  rebuild_b();
#pragma unroll
  for (int t = 0; t < 4; ++t) issue_b(t);
  advance_b();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  stamp(2);
  if (nsteps > 1) issue_a_p0(1);
  f32x4 va0[TM], vb0[TN], va1[TM], vb1[TN];
  fetch(I0{}, I0{}, I0{}, va0, vb0);

  // One step = four tiles.  DMA batches behind the tile-switch barriers of step s (A piece first, then B), `more` = a step s + 1 exists:
  //   tile 0: A(s + 1) piece 1 (2 instructions), B(s + 1, tap 0)      tile 1: A(s + 1) piece 2 (2), [A cursor -> s + 2], B(s + 1, 1)
  //   tile 2: B(s + 1, 2)                                              tile 3: A(s + 2) piece 0 (1 - 2, if it exists), B(s + 1, 3), [B cursor]
  // A tile-switch wait needs the NEXT tile's B (issued three batches ago, last of its batch) and, at tile 3, all of A(s + 1) (its last
  // piece leads the batch of tile 1).  gfx950's vmcnt is in-order, so the wait allows exactly the instructions issued after the needed
  // one: 3 / 5 / 6 / 2 (counting piece 0 as its minimum of one instruction).  The last step issues nothing and waits for vmcnt(0).
#define PGQ_SWITCH(NW)                                                                  \
  do {                                                                                  \
    if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NW) : "memory");      \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                    \
    __builtin_amdgcn_s_barrier();                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
  auto step = [&](auto asg, int s) __attribute__((always_inline)) {
    constexpr int AS = decltype(asg)::value;
    typedef std::integral_constant<int, AS> IA;
    typedef std::integral_constant<int, AS ^ 1> IN;
    const bool more = s + 1 < nsteps;
    // ================= tile 0
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I0{}, I1{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    PGQ_SWITCH(3);
    if (more) { issue_a_p1(AS ^ 1); issue_b(0); }
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I1{}, I0{}, va0, vb0);
    mfmas(va1, vb1);
    // ================= tile 1
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I1{}, I1{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    PGQ_SWITCH(5);
    if (more) { issue_a_p2(AS ^ 1); advance_a(); issue_b(1); }
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I2{}, I0{}, va0, vb0);
    mfmas(va1, vb1);
    // ================= tile 2
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I2{}, I1{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    PGQ_SWITCH(6);
    if (more) issue_b(2);
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I3{}, I0{}, va0, vb0);
    mfmas(va1, vb1);
    // ================= tile 3
    __builtin_amdgcn_sched_barrier(0);
    fetch(IA{}, I3{}, I1{}, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    PGQ_SWITCH(2);
    if (more) {
      if (s + 2 < nsteps) issue_a_p0(AS);
      issue_b(3);
      advance_b();
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(IN{}, I0{}, I0{}, va0, vb0);
    mfmas(va1, vb1);
  };
#undef PGQ_SWITCH
  for (int s = 0; s < nsteps; s += 2) {
    step(I0{}, s);
    if (s + 1 < nsteps) step(I1{}, s + 1);
  }

  // ------------------------------------------------------------------ epilogue (shared; rows outside the problem are NOT zero here)
  big_epilogue<TM, TN, STAT_OFF, STAT_N>(p, acc, smem, rows, tid, m0, nb0, wm0, wn0, bx, by, bz, 0, out_g, true, tl_on, stamp);
}

// ---------------------------------------------------------------------------------------------------------------------
// FOUR-WAVE form, TWO workgroups per CU (round 6, second step).  The timeline of the 8-wave kernel above on encoder level 1 at batch
// 32 (profiles/round6_quad_timeline_enc1_8wave.txt): the K loop runs at the power ceiling of the matrix pipe (forward 28.3 us per
// tile = 1210 TFLOP/s in the loop, data gradient 13.0 us = 1320), but a third (forward: row table 1.0 + first tile 6.9 + epilogue
// 6.2 of 42.8 us) to a half (data gradient: 1.0 + 3.6 + 10.3 of 28.1 us) of a workgroup's life is prologue and epilogue, during which
// the CU's matrix pipe idles: with 512 threads x 256 registers and 130 KB of LDS nothing else is resident.  Here a workgroup is 4
// waves (256 threads x 256 registers) with a 256 x 128 tile (the SAME wave tile, 128 x 64) and <= 80 KB of LDS, so TWO are resident
// per CU and one's prologue / epilogue overlaps the other's K loop — what the persistent kernel of round 5 could not do inside one
// workgroup (one in-order vmcnt per wave; here the two workgroups' counters are independent).  Price: the A tile's halo (one image
// row) is amortised over 256 rows instead of 512 and the weight tile is fetched per 256 rows again (DMA bytes per output row: 1.6 x
// the 8-wave form, still 0.85 x the tap-pair kernel) — affordable, because the K loop is not DMA-bound.
//   LDS: 2 A stages of 400 rows x 64 B + a ring of THREE B tiles (a B tile is issued 3 tiles ahead: two tiles of lead) + row table.
//   The A ring's (LDS row -> pixel) table is gone: the loader rebuilds its 7 row offsets arithmetically (rare path).
//   DMA batches behind the tile-switch barriers of step s (`more` = a step s + 1 exists), A piece first, then the B tile (2
//   instructions per wave):  tile 0: A(s+1) piece 1 (2), B(s, tap 3), [B cursor -> s+1]     tile 1: A(s+1) piece 2 (2), [A cursor], B(s+1, 0)
//                            tile 2: B(s+1, 1)                                              tile 3: A(s+2) piece 0 (2 - 3), B(s+1, 2)
//   counted waits (instructions allowed in flight at the barriers of tiles 0 .. 3): 4 / 4 / 4 / 2; the last step waits for vmcnt(0).
template <bool MG>
__global__ __launch_bounds__(256, 2) void conv_bf16_quad2_kernel(const ConvK p) {
  constexpr int XS = MG ? 2 : 1;
  constexpr int BM = 256, BN = 128, TM = 4, TN = 2, WGN = 2, NWV = 4;
  constexpr int ROWB = 64;
  constexpr int AROWS = 400;                              // 256 + image-row starts + one image row (Gx + XS <= 130) + the taps' slots
  constexpr int A_NI = AROWS / 16;                        // 25 wave DMA instructions per A tile
  constexpr int A_PASS = (A_NI + NWV - 1) / NWV;          // 7 (the last: wave 0 only)
  constexpr int A_ST = AROWS * ROWB, B_ST = BN * ROWB, NBS = 3;
  constexpr int B_OFF = 2 * A_ST;
  constexpr int OPS = B_OFF + NBS * B_ST;
  constexpr int ROWS_OFF = OPS, TAPS_OFF = ROWS_OFF + BM * (int)sizeof(RowB);
  constexpr int STAT_OFF = (TAPS_OFF + MAXTAP * 8 + 7) & ~7, STAT_N = 8;
  static_assert(NWV * (32 * (32 * TN + 4)) * 4 <= OPS && A_PASS == 7 && A_NI == 25, "epilogue tiles / DMA pieces");
  static_assert(STAT_OFF + STAT_N * 16 <= 80 * 1024, "LDS: two workgroups per CU");
  __shared__ __attribute__((aligned(1024))) char smem[STAT_OFF + STAT_N * 2 * 8];
  RowB* rows = reinterpret_cast<RowB*>(smem + ROWS_OFF);
  int* taps_l = reinterpret_cast<int*>(smem + TAPS_OFF);
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PG_TIMING_EXPERIMENTS
  const bool tl_on = PG_DBG(p, 16);
  const int tl_id = (int)(blockIdx.x + gridDim.x * blockIdx.z);
  auto stamp = [&](int slot) {
    if (tl_on && tid == 0 && tl_id < QTL_WGS) {
      kTimelineQ[tl_id * QTL_SLOTS + slot] = __builtin_amdgcn_s_memtime();
      if (slot == 0) {
        kTimelineQ[tl_id * QTL_SLOTS + 5] = wall_clock64();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        kTimelineQ[tl_id * QTL_SLOTS + 7] = xcc & 0xf;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        kTimelineQ[tl_id * QTL_SLOTS + 10] = hwid;
      }
      if (slot == 4) kTimelineQ[tl_id * QTL_SLOTS + 6] = wall_clock64();
    }
  };
#else
  const bool tl_on = false;
  auto stamp = [](int) {};
#endif
  stamp(0);
  int bx = blockIdx.x, by = 0, bz = blockIdx.z;
  if (p.xcd_swizzle & 1) {
    const int mt = (int)gridDim.x, P = (int)gridDim.z;
    const int L = bx + mt * bz;
    const int xcd = L & 7, j = L >> 3;
    bz = j % P;
    bx = (j / P) * 8 + xcd;
  }
  const int phase = bz;
  float* const out_g = p.out;
  const int m0 = bx * BM, nb0 = 0;
  const int ntap = p.ntap[phase];
  const int gx = p.Gx, gg = p.Gy * p.Gx;
  const int n_s = m0 / gg;                          // the tile's sample (host: gg % BM == 0)
  const int rem0 = m0 - n_s * gg;
  const int qy0 = rem0 / gx, qx0 = rem0 - qy0 * gx;
  const int IR = gx + XS;

  if (tid >= 64 && tid < 64 + STAT_N * 2) reinterpret_cast<double*>(smem + STAT_OFF)[tid - 64] = 0.0;
  if (tid < MAXTAP)
    taps_l[tid] = (p.dy[phase][tid] & 0xff) | ((p.dx[phase][tid] & 0xff) << 8) | ((int)p.wtap[phase][tid] << 16);
  if (MG && tid >= 64 && tid < 64 + MAXTAP) {
    const int q = tid - 64;
    taps_l[MAXTAP + q] = (p.dy[2 + phase][q] & 0xff) | ((p.dx[2 + phase][q] & 0xff) << 8) | ((int)p.wtap[2 + phase][q] << 16);
  }
  __syncthreads();                                  // (the tap tables: the loaders below read them; the row table follows the first DMAs)

  const int cpt = p.Ctot / 32;
  const int nsteps = PG_DBG(p, 8) ? 1 : (ntap >> 2) * cpt;

  const int wm0 = (wave / WGN) * (TM * 32);
  const int wn0 = (wave % WGN) * (TN * 32);
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- DMA loader state
  const char* const zero_pg = uniform_ptr(reinterpret_cast<const char*>(kZeros));
  const char* const wp = uniform_ptr(reinterpret_cast<const char*>(p.W));
  const int chunk = (tid & 3) ^ ((tid >> 4) & 3);
  unsigned pa[A_PASS];
  unsigned pa_ok = 0;
  unsigned pb = 0;
  const char* a_src = zero_pg;
  long tapoff[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // uniform: byte offsets of the B cursor's taps in W for B rows 0 - 63 / 64 - 127
  int a_q = 0, a_ci = 0, a_left = nsteps;
  int b_q = 0, b_ci = 0, b_left = nsteps;
  auto rebuild_a = [&]() __attribute__((always_inline)) {
    const int tp = lds_rd32_now(lds0 + TAPS_OFF + (4 * a_q) * 4);
    const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
    const int cc = a_ci * 32;
    const char* sp = reinterpret_cast<const char*>(p.src[0].ptr);
    int sC = p.src[0].C, cs = 0;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.nsrc && cc >= p.cstart[q]) { sp = reinterpret_cast<const char*>(p.src[q].ptr); sC = p.src[q].C; cs = p.cstart[q]; }
    a_src = uniform_ptr(sp);
    const int cl = cc - cs + chunk * 8;
    pa_ok = 0;
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int rho = min((i * NWV + wave) * 16 + (lane >> 2), AROWS - 1);
      const int t = rho + qx0;                        // LDS row -> (local image row j, slot k): rho + qx0 = j (Gx + XS) + k
      const int j = t / IR, k = t - j * IR;
      const int iy = (qy0 + j) * p.si + dyv, ix = k * p.si + dxv;
      const bool ok = (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      pa[i] = ok ? ((unsigned)((n_s * p.Hi + iy) * p.Wi + ix) * (unsigned)sC + (unsigned)cl) * 2u : 0u;
      pa_ok |= (ok ? 1u : 0u) << i;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_a = [&]() __attribute__((always_inline)) {
    if (--a_left <= 0) return;
    if (++a_ci == cpt) { a_ci = 0; ++a_q; rebuild_a(); return; }
    bool src_edge = false;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && a_ci * 32 == p.cstart[q]) src_edge = true;
    if (src_edge) { rebuild_a(); return; }
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) pa[i] += ROWB;
  };
  auto rebuild_b = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tp = lds_rd32_now(lds0 + TAPS_OFF + (unsigned)((4 * b_q + t) * 4));
      tapoff[0][t] = (long)__builtin_amdgcn_readfirstlane(tp >> 16) * p.wCout * p.wCin * 2;
      if constexpr (MG) {                            // B rows 64 - 127: phase (py, 1)'s taps, the same 64 output channels
        const int tq = lds_rd32_now(lds0 + TAPS_OFF + (unsigned)((MAXTAP + 4 * b_q + t) * 4));
        tapoff[1][t] = (long)__builtin_amdgcn_readfirstlane(tq >> 16) * p.wCout * p.wCin * 2;
      } else {
        tapoff[1][t] = tapoff[0][t] + (long)64 * p.wCin * 2;      // the same tap, output channels 64 - 127
      }
    }
    pb = ((unsigned)(p.n_off + (tid >> 2)) * (unsigned)p.wCin + (unsigned)(b_ci * 32 + chunk * 8)) * 2u;      // (every row valid: host)
    __builtin_amdgcn_s_waitcnt(0xC07F);
  };
  auto advance_b = [&]() __attribute__((always_inline)) {
    if (--b_left <= 0) return;
    if (++b_ci == cpt) { b_ci = 0; ++b_q; rebuild_b(); return; }
    pb += ROWB;
  };
  auto issue_a1 = [&](int stage, int i) __attribute__((always_inline)) {
    float* const As = reinterpret_cast<float*>(smem + stage * A_ST);
    const char* src = ((pa_ok >> i) & 1u) ? a_src + pa[i] : zero_pg + (tid & 3) * 16;
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), As + (i * NWV + wave) * 256, 16, 0, 0);
  };
  auto issue_a_p0 = [&](int stage) __attribute__((always_inline)) {
    if (6 * NWV + wave < A_NI) issue_a1(stage, 6);    // wave 0 only (the counted waits assume >= 2 instructions in this piece)
    issue_a1(stage, 0);
    issue_a1(stage, 1);
  };
  auto issue_a_p1 = [&](int stage) __attribute__((always_inline)) { issue_a1(stage, 2); issue_a1(stage, 3); };
  auto issue_a_p2 = [&](int stage) __attribute__((always_inline)) { issue_a1(stage, 4); issue_a1(stage, 5); };
  auto issue_b = [&](int t, int bstage) __attribute__((always_inline)) {       // tap t of the B cursor's step -> B ring stage `bstage`
    float* const Bs = reinterpret_cast<float*>(smem + B_OFF) + bstage * (B_ST / 4);
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wp + tapoff[0][t] + pb), Bs + wave * 256, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(wp + tapoff[1][t] + pb), Bs + (NWV + wave) * 256, 16, 0, 0);
  };

  // ---- prologue, first half: step 0's operands are requested BEFORE the row table is built
  rebuild_a();
#pragma unroll
  for (int i = 0; i < A_PASS; ++i)
    if (i * NWV + wave < A_NI) issue_a1(0, i);
  advance_a();
  rebuild_b();
  issue_b(0, 0); issue_b(1, 1); issue_b(2, 2);
  {
    RowB ri;
    const int m = m0 + tid;
    ri.n = -1; ri.opix = 0; ri.iy = 0; ri.ix = 0; ri.oy = 0; ri.ox = 0;
    if (m < p.M) {
      const int rem = rem0 + tid;
      const int qy = rem / gx;
      const int qx = rem - qy * gx;
      const int oy = qy * p.so + p.phy[phase];
      const int ox = qx * p.so + p.phx[phase];
      if (oy < p.Ho && ox < p.Wo) {
        ri.n = n_s; ri.iy = (short)(qy * p.si); ri.ix = (short)(qx * p.si); ri.oy = (short)oy; ri.ox = (short)ox;
        ri.opix = (n_s * p.Ho + oy) * p.Wo + ox;
        if (MG) ri.opix >>= 1;
      }
    }
    rows[tid] = ri;
  }
  stamp(1);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  unsigned abase[4][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm0 + 32 * i + l31;
    const int rho = r + XS * ((qx0 + r) / gx) + ((MG && wn0 >= BN / 2) ? 1 : 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = rho + (t & 1) + (t >> 1) * IR, s = (row >> 2) & 3;
      abase[t][i] = lds0 + (unsigned)(row * ROWB) + (unsigned)((lhi ^ s) << 4);
    }
  }
  const int swr = (l31 >> 2) & 3;
  const unsigned fb0 = lds0 + B_OFF + (unsigned)((wn0 + l31) * ROWB) + (unsigned)((lhi ^ swr) * 16);
  unsigned fbc = fb0;                               // B fragment address in the CURRENT ring stage
  int bs_rd = 0;                                    // ring stage of the current tile (uniform)
  // AS: A ring stage, TT: tap of the quad, KS: k-step — compile-time; the B ring stage is the run-time address `fbc`
  auto fetch = [&](auto asg, auto tt, auto ksc, unsigned fbv, f32x4 (&va)[TM], f32x4 (&vb)[TN]) __attribute__((always_inline)) {
    constexpr int AS = decltype(asg)::value, TT = decltype(tt)::value, KS = decltype(ksc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (KS == 0) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(va[i]) : "v"(abase[TT][i]), "n"(AS * A_ST));
      } else {
        unsigned t;
        asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                     : "=v"(va[i]), "=&v"(t) : "v"(abase[TT][i]), "n"(KS * 32), "n"(AS * A_ST));
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (KS == 0) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vb[j]) : "v"(fbv), "n"(j * 32 * ROWB));
      } else {
        unsigned t;
        asm volatile("v_xor_b32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4"
                     : "=v"(vb[j]), "=&v"(t) : "v"(fbv), "n"(KS * 32), "n"(j * 32 * ROWB));
      }
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

  // ---- prologue, second half
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  stamp(2);
  if (nsteps > 1) issue_a_p0(1);
  f32x4 va0[TM], vb0[TN], va1[TM], vb1[TN];
  fetch(I0{}, I0{}, I0{}, fbc, va0, vb0);

#define PGQ_SWITCH(NW)                                                                  \
  do {                                                                                  \
    if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NW) : "memory");      \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                    \
    __builtin_amdgcn_s_barrier();                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
  // the ring stage the tile that just ended used is free: the batch's B tile goes there; the next tile reads the next stage
#define PGQ_NEXT_STAGE()                                                                \
  const int bs_free = bs_rd;                                                            \
  bs_rd = bs_rd == NBS - 1 ? 0 : bs_rd + 1;                                             \
  fbc = fb0 + (unsigned)(bs_rd * B_ST)
  auto step = [&](auto asg, int s) __attribute__((always_inline)) {
    constexpr int AS = decltype(asg)::value;
    typedef std::integral_constant<int, AS> IA;
    typedef std::integral_constant<int, AS ^ 1> IN;
    const bool more = s + 1 < nsteps;
    {   // ================= tile 0
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I0{}, I1{}, fbc, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      PGQ_SWITCH(4);
      PGQ_NEXT_STAGE();
      if (more) issue_a_p1(AS ^ 1);
      issue_b(3, bs_free);
      advance_b();
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I1{}, I0{}, fbc, va0, vb0);
      mfmas(va1, vb1);
    }
    {   // ================= tile 1
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I1{}, I1{}, fbc, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      PGQ_SWITCH(4);
      PGQ_NEXT_STAGE();
      if (more) { issue_a_p2(AS ^ 1); advance_a(); issue_b(0, bs_free); }
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I2{}, I0{}, fbc, va0, vb0);
      mfmas(va1, vb1);
    }
    {   // ================= tile 2
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I2{}, I1{}, fbc, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      PGQ_SWITCH(4);
      PGQ_NEXT_STAGE();
      if (more) issue_b(1, bs_free);
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I3{}, I0{}, fbc, va0, vb0);
      mfmas(va1, vb1);
    }
    {   // ================= tile 3
      __builtin_amdgcn_sched_barrier(0);
      fetch(IA{}, I3{}, I1{}, fbc, va1, vb1);
      PGB_LDS_WAIT(NRD);
      mfmas(va0, vb0);
      PGQ_SWITCH(2);
      PGQ_NEXT_STAGE();
      if (more) {
        if (s + 2 < nsteps) issue_a_p0(AS);
        issue_b(2, bs_free);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) fetch(IN{}, I0{}, I0{}, fbc, va0, vb0);
      mfmas(va1, vb1);
    }
  };
#undef PGQ_SWITCH
#undef PGQ_NEXT_STAGE
  for (int s = 0; s < nsteps; s += 2) {
    step(I0{}, s);
    if (s + 1 < nsteps) step(I1{}, s + 1);
  }

  big_epilogue<TM, TN, STAT_OFF, STAT_N>(p, acc, smem, rows, tid, m0, nb0, wm0, wn0, bx, by, bz, 0, out_g, true, tl_on, stamp);
}

// merged: the x-phase merged transposed form (grid.z = the two py); waves = 8: 512-row tiles, one workgroup per CU; 4: 256-row tiles, two
void launch_conv_bf16_quad(const ConvK& k, bool merged, int waves, dim3 grid, hipStream_t st) {
  if (waves == 4) {
    if (merged) PG_KLAUNCH((conv_bf16_quad2_kernel<true>), grid, dim3(256), 0, st, k);
    else PG_KLAUNCH((conv_bf16_quad2_kernel<false>), grid, dim3(256), 0, st, k);
    return;
  }
  if (merged) PG_KLAUNCH((conv_bf16_quad_kernel<true>), grid, dim3(512), 0, st, k);
  else PG_KLAUNCH((conv_bf16_quad_kernel<false>), grid, dim3(512), 0, st, k);
}

}  // namespace pg

#ifdef PG_TIMING_EXPERIMENTS
// timing builds only: the time stamps of the last PG_DEBUG_CONV_TIMELINE launch of the quad kernel (n_wgs x 16 uint64)
extern "C" int pg_debug_conv_timeline_quad(unsigned long long* host_out, int32_t n_wgs) {
  if (host_out == nullptr || n_wgs <= 0 || n_wgs > pg::QTL_WGS) return 1;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pg::kTimelineQ), sizeof(unsigned long long) * pg::QTL_SLOTS * (size_t)n_wgs, 0,
                             hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
