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

#include "igemm_common.h"

namespace pg {

template <int OFF>
__device__ __forceinline__ void lds_rd128(f32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ int lds_rd32_now(unsigned addr) {      // opaque LDS word read, waited for
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
#define PGB_LDS_WAIT(n)                                             \
  do {                                                              \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");      \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)

template <int BN>
__global__ __launch_bounds__(512, 2) void conv_bf16_big_kernel(const ConvK p) {
  constexpr int BM = (BN == 64) ? 512 : 256;                // BN = 64 (N = 64 layers at full resolution): 512 x 64, a wave owns 64 x 64
  constexpr int WGN = BN / 64, WGM = 8 / WGN;            // 256: 2 x 4 waves, 128: 4 x 2 waves, 64: 8 x 1 waves
  constexpr int TM = BM / WGM / 32, TN = 2;              // MFMA tiles per wave: 4 x 2 or 2 x 2
  constexpr int A_PASS = BM / 64, B_PASS = BN / 64;      // global_load_lds per thread and tile (64 rows per pass)
  constexpr int A_ST = BM * 128, B_ST = BN * 128;        // bytes per stage
  constexpr int STAGE = A_ST + B_ST;
  constexpr int ROWS_OFF = 2 * STAGE, TAPS_OFF = ROWS_OFF + BM * (int)sizeof(RowInfo);
  __shared__ __attribute__((aligned(1024))) char smem[TAPS_OFF + MAXTAP * 4];     // ONE LDS object (see header)
  RowInfo* rows = reinterpret_cast<RowInfo*>(smem + ROWS_OFF);
  int* taps_l = reinterpret_cast<int*>(smem + TAPS_OFF);
  const unsigned lds0 = (unsigned)(size_t)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // Workgroups are dealt round-robin to the 8 XCDs (one 4 MB L2 each) in dispatch order.  Remapped, the j-th workgroup an
  // XCD receives walks (sub-pixel phase fastest, then N tile, then M tile): the four phases of a transposed convolution /
  // conv data-gradient read the same input pixels through different taps, the N tiles of an M tile the same activation
  // rows — both are then served from that XCD's L2 instead of the fabric (MI355X_MICROARCH.md: per-XCD L2s).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_swizzle) {          // host: gridDim.x % 8 == 0, ksplit == 1, no tap batch
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
        ri.n = n; ri.iy = (short)(qy * p.si); ri.ix = (short)(qx * p.si); ri.oy = (short)oy; ri.ox = (short)ox;
      }
    }
    rows[tid] = ri;
  }
  __syncthreads();

  const int cpt = p.Ctot / 64;                      // K tiles per tap
  const int ktot = ntap * cpt;
  const int kper = (ktot + p.ksplit - 1) / p.ksplit;
  const int kt0 = split * kper;
  const int kt1 = min(ktot, kt0 + kper);
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
  const int chunk = (tid & 7) ^ ((tid >> 4) & 7);
  const char* pa[A_PASS];
  const char* pb[B_PASS];
  int ld_kt = kt0, ld_tap = kt0 / cpt, ld_ci = kt0 - (kt0 / cpt) * cpt;
  auto rebuild = [&]() {
    const int tp = lds_rd32_now(lds0 + TAPS_OFF + ld_tap * 4);
    const int dyv = (int)(signed char)(tp & 0xff), dxv = (int)(signed char)((tp >> 8) & 0xff);
    const int cc = ld_ci * 64;
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
      const unsigned ra = lds0 + ROWS_OFF + (unsigned)(((tid >> 3) + 64 * i) * (int)sizeof(RowInfo));
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4" : "=&v"(rn[i]), "=&v"(ryx[i]) : "v"(ra));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
      const int iy = (int)(short)(ryx[i] & 0xffff) + dyv, ix = (ryx[i] >> 16) + dxv;
      const bool ok = (rn[i] >= 0) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      const long off = ((long)((rn[i] * p.Hi + iy) * p.Wi + ix) * sC + cl) * 2;
      pa[i] = ok ? sp + off : zero_pg + (tid & 7) * 16;
    }
    const int base = (tp >> 16) * p.wCout;
#pragma unroll
    for (int i = 0; i < B_PASS; ++i) {
      const int n = nb0 + (tid >> 3) + 64 * i;
      const long off = ((long)(base + p.n_off + n) * p.wCin + cc + chunk * 8) * 2;
      pb[i] = (n < p.n_cnt) ? wp + off : zero_pg + (tid & 7) * 16;
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
        for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.nsrc && ld_ci * 64 == p.cstart[q]) src_edge = true;
        if (src_edge) rebuild();
        else {
#pragma unroll
          for (int i = 0; i < A_PASS; ++i) pa[i] += 128;
#pragma unroll
          for (int i = 0; i < B_PASS; ++i) pb[i] += 128;
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

  // ---- operand fetch: k-step ks (16 k's) of a stage -> one register set (TM + TN ds_read_b128)
  const int swr = (l31 >> 1) & 7;
  unsigned fa[4], fb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fa[ks] = lds0 + (unsigned)((wm0 + l31) * 128) + (unsigned)(((2 * ks + lhi) ^ swr) * 16);
    fb[ks] = lds0 + A_ST + (unsigned)((wn0 + l31) * 128) + (unsigned)(((2 * ks + lhi) ^ swr) * 16);
  }
  auto fetch = [&](int stage, int ks, f32x4 (&va)[TM], f32x4 (&vb)[TN]) {
    const unsigned aa = fa[ks] + (unsigned)(stage * STAGE), bb = fb[ks] + (unsigned)(stage * STAGE);
    lds_rd128<0>(va[0], aa);
    lds_rd128<4096>(va[1], aa);
    if constexpr (TM == 4) { lds_rd128<8192>(va[2], aa); lds_rd128<12288>(va[3], aa); }
    lds_rd128<0>(vb[0], bb);
    lds_rd128<4096>(vb[1], bb);
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
  f32x4 va0[TM], vb0[TN], va1[TM], vb1[TN];
  fetch(0, 0, va0, vb0);
  int stage = 0;
  // tile kt0 + 1 is issued before the loop; inside, the DMA of tile kt + 2 is issued RIGHT AFTER the tile-switch barrier of
  // tile kt (its stage was released by that barrier) and before the last 8 MFMAs of tile kt: a full tile of MFMA time
  // (2048 SIMD cycles with two waves per SIMD) lies between a DMA's issue and the vmcnt(0) that waits for it
  // (PMC, round 2: with the issue after those MFMAs the waves were parked at the wait for 37 % of their cycles).
  if (kt0 + 1 < kt1) { issue(1); advance(); }
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    __builtin_amdgcn_sched_barrier(0);
    fetch(stage, 1, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    fetch(stage, 2, va0, vb0);
    PGB_LDS_WAIT(NRD);
    mfmas(va1, vb1);
    fetch(stage, 3, va1, vb1);
    PGB_LDS_WAIT(NRD);
    mfmas(va0, vb0);
    // tile switch: this wave's last operand fetch of the stage has landed (lgkmcnt(0)), its share of the next tile has
    // landed (vmcnt(0)); after the barrier both hold for every wave
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < kt1) { issue(stage); advance(); }          // tile kt + 2 into the stage this tile just released
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch(stage ^ 1, 0, va0, vb0);
    mfmas(va1, vb1);
    stage ^= 1;
  }

  // ------------------------------------------------------------------ epilogue (operand stages are free after a barrier)
  __syncthreads();
  float* const T = reinterpret_cast<float*>(smem) + wave * (32 * (32 * TN + 4));
  const int ngc = nb0 + wn0 + (lane % (8 * TN)) * 4;            // first of this lane's 4 columns
  if (p.part != nullptr) {                                      // split-K through the workspace: plain partial tiles
    float* pp = p.part + (long)split * p.part_stride;
    float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < TM / 2; ++h)
      vec_store_64x64<TN>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, pp, p.n_cnt,
                          p.Ho, p.Wo, ngc, make_float4(0.f, 0.f, 0.f, 0.f), false, 0, s0, s1, nullptr);
    return;
  }
  if (p.epilogue == 0) {
    const bool do_stats = p.stats != nullptr;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && ngc < p.n_cnt) bv = *reinterpret_cast<const float4*>(p.bias + ngc);
    // fused per-sample statistics of the following norm layer: a 64-row half of the wave tile spans at most two
    // consecutive samples on the layers this kernel serves (>= 64 pixels per sample); the rare rest goes to stat_spill
    double* red = reinterpret_cast<double*>(smem + 8 * (32 * (32 * TN + 4)) * 4);       // behind the 8 wave tiles
    int* redn = reinterpret_cast<int*>(red + 8 * 2 * 2 * 2);
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
      int stat_n0 = 0;
      if (do_stats) {
        const int nn = rows[wm0 + 64 * h + lane].n;
        int nf = nn >= 0 ? nn : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nf = min(nf, __shfl_xor(nf, o));
        stat_n0 = nf;
      }
      float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};
      if (p.out_bf16)
        vec_store_64x64<TN, true>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, out_g, p.n_cnt,
                                  p.Ho, p.Wo, ngc, bv, do_stats, stat_n0, st_s, st_q, p.stats);
      else
        vec_store_64x64<TN, false>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, out_g, p.n_cnt,
                                   p.Ho, p.Wo, ngc, bv, do_stats, stat_n0, st_s, st_q, p.stats);
      if (do_stats) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double ds = wave_sum_d((double)st_s[k]), dq = wave_sum_d((double)st_q[k]);
          if (lane == 0) { red[((wave * 2 + h) * 2 + k) * 2] = ds; red[((wave * 2 + h) * 2 + k) * 2 + 1] = dq; }
        }
        if (lane == 0) redn[wave * 2 + h] = stat_n0;
      }
    }
    if (do_stats) {
      // block-level merge: (wave, half, k) entries of equal sample are summed by the first of them, then ONE pair of
      // double atomics per sample and workgroup, spread over PG_STAT_SLOTS addresses per sample
      __syncthreads();
      constexpr int NE = 8 * (TM / 2) * 2;
      if (tid < NE) {
        const int w = tid / ((TM / 2) * 2), hk = tid - w * ((TM / 2) * 2), h = hk >> 1, k = hk & 1;
        auto sample = [&](int e) {
          const int ww = e / ((TM / 2) * 2), hh = (e - ww * ((TM / 2) * 2)) >> 1, kk = e & 1;
          const int b = redn[ww * 2 + hh];
          return b == 0x7fffffff ? -1 : b + kk;
        };
        auto slot_of = [&](int e) {
          const int ww = e / ((TM / 2) * 2), hh = (e - ww * ((TM / 2) * 2)) >> 1, kk = e & 1;
          return ((ww * 2 + hh) * 2 + kk) * 2;
        };
        (void)h; (void)k;
        const int n = sample(tid);
        double ds = red[slot_of(tid)], dq = red[slot_of(tid) + 1];
        bool first = true;
        for (int o = 0; o < tid; ++o) if (sample(o) == n) first = false;
        if (first && n >= 0) {
          for (int o = tid + 1; o < NE; ++o) if (sample(o) == n) { ds += red[slot_of(o)]; dq += red[slot_of(o) + 1]; }
          if (ds != 0.0 || dq != 0.0) {
            const int slot = (bx + by * 5 + bz * 3) % PG_STAT_SLOTS;
            atomicAdd(&p.stats[((long)n * PG_STAT_SLOTS + slot) * 2], ds);
            atomicAdd(&p.stats[((long)n * PG_STAT_SLOTS + slot) * 2 + 1], dq);
          }
        }
      }
    }
    return;
  }
  // ---- data-gradient scatter (host guarantees vec_dst: every destination C % 32 == 0, aligned)
  {
    const bool cval = ngc < p.n_cnt;
    const int ngs = cval ? ngc : 0;
    float* gradp = p.dst[0].grad;
    const float *fwd0 = p.dst[0].fwd, *aff0 = p.dst[0].aff, *mask0 = p.dst[0].mask;
    int C = p.dst[0].C, dact = p.dst[0].act, dacc = p.dst[0].accumulate, cst = 0, dfl = p.dst[0].flags;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.ndst && ngs >= p.dstart[q]) {
        gradp = p.dst[q].grad; fwd0 = p.dst[q].fwd; aff0 = p.dst[q].aff; mask0 = p.dst[q].mask;
        C = p.dst[q].C; dact = p.dst[q].act; dacc = p.dst[q].accumulate; cst = p.dstart[q]; dfl = p.dst[q].flags;
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
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
      if (p.dst_io == 0)
        vec_scatter_64x64<TN, 0>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, ld, cval, p.Ho, p.Wo);
      else if (p.dst_io == 1)
        vec_scatter_64x64<TN, 1>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, ld, cval, p.Ho, p.Wo);
      else
        vec_scatter_64x64<TN, 2>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, ld, cval, p.Ho, p.Wo);
    }
  }
}

// launch helper used by conv_impl (igemm_conv.hip)
void launch_conv_bf16_big(const ConvK& k, int bn, dim3 grid, hipStream_t st) {
  if (bn == 256) PG_KLAUNCH((conv_bf16_big_kernel<256>), grid, dim3(512), 0, st, k);
  else if (bn == 64) PG_KLAUNCH((conv_bf16_big_kernel<64>), grid, dim3(512), 0, st, k);
  else PG_KLAUNCH((conv_bf16_big_kernel<128>), grid, dim3(512), 0, st, k);
}

}  // namespace pg
