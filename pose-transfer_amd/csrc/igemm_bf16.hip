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

// Timing experiments of round 3 (per-workgroup phase stamps; K loops without barrier / DMA / every 4th A tile — "results are
// wrong, times are not"): compiled in only with -DPG_TIMING_EXPERIMENTS (tools/conv_timeline.py builds its own library).
#ifdef PG_TIMING_EXPERIMENTS
#define PG_DBG(p, bit) (((p).xcd_swizzle & (bit)) != 0)
#else
#define PG_DBG(p, bit) false
#endif

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


// Row table entry of this kernel: RowInfo plus the output pixel index (n*Ho + oy)*Wo + ox, so that an epilogue row pass needs one
// ds_read_b64 (n, opix) and one multiply-add for its address.  The epilogues below are VALU-bound with one workgroup per CU (every
// wave64 VALU instruction is 4 cycles, two waves per SIMD): ~38 instructions per row pass were 1.4 us per 32-row half.
struct RowB {
  int n;           // sample index, -1 = row outside the problem
  int opix;        // output pixel index
  short iy, ix;    // input base coordinate (q*si)
  short oy, ox;    // output coordinate (shared epilogues of igemm_common.h)
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// wave-wide fp32 sum without LDS traffic: four DPP steps inside the rows of 16 lanes, then the four row sums by v_readlane
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);       // row_half_mirror
  v += dpp_f<0x140>(v);       // row_mirror
  const int iv = __builtin_bit_cast(int, v);
  return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
         (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}

// Forward-output store of a wave's 64 x 64 sub-tile through its private LDS tile (layout as vec_store_64x64), bf16 or fp32 rows.
// n_lo = sample of the block's first row, one_sample = all 64 rows belong to it (wave-uniform, from the tile geometry).
// Without a bias the rows outside the problem are exact zeros (their A rows were the zero page): they add nothing to the sums.
template <int TN_, bool OB>
__device__ __forceinline__ void big_store_64x64(const f32x16 (&acc)[2][TN_], float* T, const RowB* rows, int wm0, int lane, void* obase,
                                                int n_cnt, int ngc, bool has_bias, float4 bv, bool do_stats, int n_lo, bool one_sample,
                                                float (&st_s)[2], float (&st_q)[2], double* stats) {
  const bool cval = ngc < n_cnt;             // n_cnt < the tile width only on the 32-column output-convolution launch
  constexpr int PITCH = 32 * TN_ + 4, LPR = 8 * TN_, RPP = 64 / LPR, NP = 32 / RPP;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int rsel = lane / LPR, c4 = (lane % LPR) * 4;
  char* const ob = reinterpret_cast<char*>(obase) + (size_t)ngc * (OB ? 2 : 4);
  const unsigned rowb = (unsigned)n_cnt * (OB ? 2u : 4u);
  float tot_s = 0.f, tot_q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhi) * PITCH + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int2 ro[NP];
    float4 v[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int row = it * RPP + rsel;
      ro[it] = *reinterpret_cast<const int2*>(&rows[wm0 + i * 32 + row]);            // (n, opix)
      v[it] = *reinterpret_cast<const float4*>(&T[row * PITCH + c4]);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const bool ok = (ro[it].x >= 0) & cval;
      if (has_bias) { v[it].x += bv.x; v[it].y += bv.y; v[it].z += bv.z; v[it].w += bv.w; }
      if (ok) {
        char* const dst = ob + (size_t)(unsigned)ro[it].y * rowb;
        if constexpr (OB) *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16(v[it].x, v[it].y), pack_bf16(v[it].z, v[it].w));
        else *reinterpret_cast<float4*>(dst) = v[it];
      }
      if (do_stats) {
        float s4 = (v[it].x + v[it].y) + (v[it].z + v[it].w);
        float q4 = fmaf(v[it].x, v[it].x, fmaf(v[it].y, v[it].y, fmaf(v[it].z, v[it].z, v[it].w * v[it].w)));
        if (has_bias) { s4 = ok ? s4 : 0.f; q4 = ok ? q4 : 0.f; }
        if (one_sample) { tot_s += s4; tot_q += q4; }
        else {
          const int dn = ro[it].x - n_lo;
          st_s[0] += (ok && dn == 0) ? s4 : 0.f; st_q[0] += (ok && dn == 0) ? q4 : 0.f;
          st_s[1] += (ok && dn == 1) ? s4 : 0.f; st_q[1] += (ok && dn == 1) ? q4 : 0.f;
          if (ok && dn > 1) {
            stat_spill(stats, ro[it].x, v[it].x); stat_spill(stats, ro[it].x, v[it].y);
            stat_spill(stats, ro[it].x, v[it].z); stat_spill(stats, ro[it].x, v[it].w);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (one_sample) { st_s[0] += tot_s; st_q[0] += tot_q; }
}

// Data-gradient scatter of a wave's whole (TM*32) x 64 tile, gradient AND forward tensors in bf16 STORAGE, every sample >= 32
// pixels (host: dst_io == 1).  gfx950 counts loads and stores in ONE vmcnt, so waiting for a load that was issued behind a store
// also waits for that store's acknowledgement (~2 us under load): the round-2 scheme (per batch of four row passes: loads, wait,
// compute, stores) paid that once per batch — 30 us of a 97 us workgroup on the dec.5 data gradient (tools/conv_timeline.py).
// Here the global loads of 32-row half h+1 (forward values, previous gradients, the two candidate samples' affine / mask) are
// issued BEFORE the stores of half h (whose results wait in 16 registers), and nothing else reads global memory: one counted
// wait per half, never behind a store.
// SIMPLE (wave-uniform, decided by the caller): no lane's destination has a dropout mask or accumulates — the case of the large
// decoder data gradients; drops the previous-gradient and mask loads and about half of the VALU work per element.
// BS (round 4): the destination carries `bsums` — while a half's results are in registers, add (sum r, sum r * f) of its valid
// elements to the sums of the norm backward that will read this gradient next (f = the raw forward value the scatter loads
// anyway), per sample: a lane-local pair for the half's first sample, flushed (DPP wave sum -> LDS double atomic on the
// workgroup's table `stab`, indexed by sample - nbase) when the sample changes, and a second pair for rows of the next sample.
template <int TM_, int TN_, bool SIMPLE, bool BS>
__device__ __forceinline__ void big_scatter_tile(const f32x16 (&acc)[TM_][TN_], float* T, const RowB* rows, int wm0, int lane,
                                                 const LaneDst& d, bool cval, int m_first, int gg, int M, int N,
                                                 double* stab = nullptr, int nbase = 0, int stat_n = 0, int gslot = 0) {
  constexpr int PITCH = 32 * TN_ + 4, LPR = 8 * TN_, RPP = 64 / LPR, NP = 32 / RPP;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int rsel = lane / LPR, c4 = (lane % LPR) * 4;
  struct Half {
    uint2 fb[NP], ob[NP];
    float2 ab[2];
    float4 mk[2];
    int nlo;
    unsigned ok;
  };
  const unsigned short* const fwd16 = reinterpret_cast<const unsigned short*>(d.fwdp);
  unsigned short* const grad16 = reinterpret_cast<unsigned short*>(d.gradp);
  auto issue = [&](int hh, Half& L) {
    const int mf = min(m_first + 32 * hh, M - 1);
    L.nlo = __builtin_amdgcn_readfirstlane(mf / gg);
    const int nhi = min(L.nlo + 1, N - 1);
    L.ok = 0;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int2 ro = *reinterpret_cast<const int2*>(&rows[wm0 + hh * 32 + it * RPP + rsel]);
      const bool ok = (ro.x >= 0) & cval;
      const unsigned idx = ok ? (unsigned)ro.y * (unsigned)d.C + (unsigned)d.c : (unsigned)d.c;
      L.ok |= (ok ? 1u : 0u) << it;
      L.fb[it] = *reinterpret_cast<const uint2*>(fwd16 + (d.has_fwd ? idx : (unsigned)d.c));
      if constexpr (!SIMPLE) L.ob[it] = *reinterpret_cast<const uint2*>(grad16 + (d.accum ? idx : (unsigned)d.c));
    }
    L.ab[0] = *reinterpret_cast<const float2*>(d.affp + d.affmul * L.nlo);
    L.ab[1] = *reinterpret_cast<const float2*>(d.affp + d.affmul * nhi);
    if constexpr (!SIMPLE) {
      L.mk[0] = *reinterpret_cast<const float4*>(d.maskp + (d.has_mask ? L.nlo * d.C + d.c : (d.c & 511)));
      L.mk[1] = *reinterpret_cast<const float4*>(d.maskp + (d.has_mask ? nhi * d.C + d.c : (d.c & 511)));
    }
  };
  int run_n = -1;                    // BS: sample of the running pair
  float run_s = 0.f, run_q = 0.f;
  auto flush = [&](int n, float s_, float q_) {
    const double ds = (double)wave_sum_dpp(s_), dq = (double)wave_sum_dpp(q_);
    if (lane == 0 && n >= 0 && (ds != 0.0 || dq != 0.0)) {
      const int sl = n - nbase;
      if (sl >= 0 && sl < stat_n) { atomicAdd(&stab[sl * 2], ds); atomicAdd(&stab[sl * 2 + 1], dq); }
      else {
        atomicAdd(&d.bsums[((long)n * PG_STAT_SLOTS + gslot) * 2], ds);
        atomicAdd(&d.bsums[((long)n * PG_STAT_SLOTS + gslot) * 2 + 1], dq);
      }
    }
  };
  // one 32-row half: `cur` holds its loads; before its stores go out the loads of half `nh` are issued into `nx` (nh < 0: none)
  auto step = [&](int hh, Half& cur, Half& nx, int nh) {
    float hi_s = 0.f, hi_q = 0.f;
    if constexpr (BS) {
      if (cur.nlo != run_n) { flush(run_n, run_s, run_q); run_n = cur.nlo; run_s = 0.f; run_q = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhi) * PITCH + j * 32 + l31] = acc[hh][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint2 res[NP];
    unsigned oidx[NP];
#pragma unroll
    for (int b = 0; b < NP / 4; ++b) {
      float4 v[4];
      int2 ro[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (b * 4 + u) * RPP + rsel;
        v[u] = *reinterpret_cast<const float4*>(&T[row * PITCH + c4]);
        ro[u] = *reinterpret_cast<const int2*>(&rows[wm0 + hh * 32 + row]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = b * 4 + u;
        const bool hi = ro[u].x > cur.nlo;
        const float a = hi ? cur.ab[1].x : cur.ab[0].x, bb = hi ? cur.ab[1].y : cur.ab[0].y;
        const float g4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        const float f4[4] = {bf16_lo_f32(cur.fb[it].x), bf16_hi_f32(cur.fb[it].x), bf16_lo_f32(cur.fb[it].y), bf16_hi_f32(cur.fb[it].y)};
        float r4[4];
        if constexpr (SIMPLE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) r4[e] = g4[e] * act_grad_s(fmaf(f4[e], a, bb), d.dslope);
        } else {
          const float m4[4] = {hi ? cur.mk[1].x : cur.mk[0].x, hi ? cur.mk[1].y : cur.mk[0].y, hi ? cur.mk[1].z : cur.mk[0].z,
                               hi ? cur.mk[1].w : cur.mk[0].w};
          const float o4[4] = {bf16_lo_f32(cur.ob[it].x), bf16_hi_f32(cur.ob[it].x), bf16_lo_f32(cur.ob[it].y), bf16_hi_f32(cur.ob[it].y)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(f4[e], a, bb) * m4[e];
            r4[e] = fmaf(g4[e] * m4[e], act_grad_s(z, d.dslope), d.accum ? o4[e] : 0.f);
          }
        }
        res[it] = make_uint2(pack_bf16(r4[0], r4[1]), pack_bf16(r4[2], r4[3]));
        oidx[it] = (unsigned)ro[u].y * (unsigned)d.C + (unsigned)d.c;
        if constexpr (BS) {
          const float okf = ((cur.ok >> it) & 1u) ? 1.f : 0.f;
          const float s4 = okf * ((r4[0] + r4[1]) + (r4[2] + r4[3]));
          const float q4 = okf * fmaf(r4[0], f4[0], fmaf(r4[1], f4[1], fmaf(r4[2], f4[2], r4[3] * f4[3])));
          run_s += hi ? 0.f : s4; run_q += hi ? 0.f : q4;
          hi_s += hi ? s4 : 0.f; hi_q += hi ? q4 : 0.f;
        }
      }
    }
    if constexpr (BS) {
      if (__builtin_amdgcn_ballot_w64(hi_s != 0.f || hi_q != 0.f) != 0) flush(cur.nlo + 1, hi_s, hi_q);      // rows of the next sample (rare)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned okh = cur.ok;
    if (nh >= 0 && nh < TM_) issue(nh, nx);       // later halves' loads go out BEFORE this half's stores
#pragma unroll
    for (int it = 0; it < NP; ++it)
      if ((okh >> it) & 1u) *reinterpret_cast<uint2*>(grad16 + oidx[it]) = res[it];
  };
  Half ha, hb, hc;
  issue(0, ha);
  if constexpr (SIMPLE) {
    // few registers per half: two halves of loads in flight (the whole HBM latency is behind a half of compute)
    if (TM_ > 1) issue(1, hb);
    step(0, ha, hc, 2);
    if (TM_ > 1) step(1, hb, ha, 3);
    if (TM_ > 2) step(2, hc, hb, -1);
    if (TM_ > 3) step(3, ha, hb, -1);
  } else {
    step(0, ha, hb, 1);
    if (TM_ > 1) step(1, hb, ha, 2);
    if (TM_ > 2) step(2, ha, hb, 3);
    if (TM_ > 3) step(3, hb, ha, -1);
  }
  if constexpr (BS) flush(run_n, run_s, run_q);
}

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
  auto fetch = [&](int stage, int ks, f32x4 (&va)[TM], f32x4 (&vb)[TN]) {
    const unsigned aa = fa[ks] + (unsigned)(stage * STAGE), bb = fb[ks] + (unsigned)(stage * STAGE);
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
        if (!(PG_DBG(p, 1024) && (kt & 3))) issue_a(stage);      // bit 10: PG_DEBUG_A_EVERY_4TH (timing experiment: A rows on one K tile in four)
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
  __syncthreads();
  stamp(3);
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
    // Merge inside the workgroup through LDS double atomics on a table indexed by (sample - first sample of the tile), then
    // ONE pair of global double atomics per sample and workgroup, spread over PG_STAT_SLOTS addresses per sample.  (Round 2
    // merged with a serial scan over the (wave, half) entries: 3.8 - 6.3 us per workgroup, tools/conv_timeline.py.)
    double* const stab = reinterpret_cast<double*>(smem + STAT_OFF);
    const int nbase = m0 / (p.Gy * p.Gx);
    const int gslot = (bx + by * 5 + bz * 3) % PG_STAT_SLOTS;
    const int gg = p.Gy * p.Gx;
    const int wrow0 = __builtin_amdgcn_readfirstlane(m0 + wm0);
    const bool has_bias = p.bias != nullptr;
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
      // sample of the block's first row / of its last row inside the problem: wave-uniform, from the tile geometry
      const int mf = wrow0 + 64 * h;
      const bool any = mf < p.M;
      const int n_lo = min(mf, p.M - 1) / gg;
      const bool one_sample = min(mf + 63, p.M - 1) / gg == n_lo;
      float st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f};
      if (h == 0) stamp(14);
      if (p.out_bf16)
        big_store_64x64<TN, true>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, out_g, p.n_cnt,
                                  ngc, has_bias, bv, do_stats, n_lo, one_sample, st_s, st_q, p.stats);
      else
        big_store_64x64<TN, false>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, out_g, p.n_cnt,
                                   ngc, has_bias, bv, do_stats, n_lo, one_sample, st_s, st_q, p.stats);
      if (do_stats && any) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k == 1 && one_sample) break;
          const double ds = (double)wave_sum_dpp(st_s[k]), dq = (double)wave_sum_dpp(st_q[k]);
          if (lane == 0 && (ds != 0.0 || dq != 0.0)) {
            const int n = n_lo + k, sl = n - nbase;
            if (sl >= 0 && sl < STAT_N) { atomicAdd(&stab[sl * 2], ds); atomicAdd(&stab[sl * 2 + 1], dq); }
            else {
              atomicAdd(&p.stats[((long)n * PG_STAT_SLOTS + gslot) * 2], ds);
              atomicAdd(&p.stats[((long)n * PG_STAT_SLOTS + gslot) * 2 + 1], dq);
            }
          }
        }
      }
      if (h == 0) stamp(15);
    }
    stamp(8);                   // stores issued, wave sums done
    if (do_stats) {
      __syncthreads();
      stamp(9);                 // every wave's stores drained (the barrier waits for vmcnt(0))
      if (tid < STAT_N * 2) {
        const double v = stab[tid];
        if (v != 0.0) atomicAdd(&p.stats[((long)(nbase + (tid >> 1)) * PG_STAT_SLOTS + gslot) * 2 + (tid & 1)], v);
      }
    }
    if (tl_on) { __syncthreads(); stamp(4); }
    return;
  }
  // ---- data-gradient scatter (host guarantees vec_dst: every destination C % 32 == 0, aligned)
  {
    bool cval = ngc < p.n_cnt;
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
    double* bs0 = p.dst[0].bsums;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.ndst && ngs >= p.dstart[q]) bs0 = p.dst[q].bsums;
    LaneDst ld;
    ld.bsums = bs0;
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
    if (PG_DBG(p, 32)) { ld.has_fwd = false; ld.fwdp = gradp; }      // PG_DEBUG_EPI_NOFWD / _NOSTORE / _NOACC: epilogue experiments
    if (PG_DBG(p, 64)) cval = false;
    if (PG_DBG(p, 128)) ld.accum = false;
    stamp(8);
    if (p.dst_io == 1) {          // host: every destination and forward tensor in bf16 STORAGE, >= 32 pixels per sample
      const bool plain = __builtin_amdgcn_ballot_w64(ld.has_mask || ld.accum) == 0;       // wave-uniform
      const int mfw = __builtin_amdgcn_readfirstlane(m0 + wm0);
      // host (conv_impl): a workgroup's column tile lies inside ONE destination whenever a destination carries bsums, so
      // the choice is workgroup-uniform and the table `stab` holds one destination's sums
      const bool bs_on = __builtin_amdgcn_readfirstlane((int)(ld.bsums != nullptr && ld.has_fwd)) != 0;
      double* const stab = reinterpret_cast<double*>(smem + STAT_OFF);
      const int nbase = m0 / (p.Gy * p.Gx), gslot = (bx + by * 5 + bz * 3) % PG_STAT_SLOTS;
      if (bs_on) {
        if (plain) big_scatter_tile<TM, TN, true, true>(acc, T, rows, wm0, lane, ld, cval, mfw, p.Gy * p.Gx, p.M, p.N, stab, nbase, STAT_N, gslot);
        else big_scatter_tile<TM, TN, false, true>(acc, T, rows, wm0, lane, ld, cval, mfw, p.Gy * p.Gx, p.M, p.N, stab, nbase, STAT_N, gslot);
        __syncthreads();
        if (tid < STAT_N * 2) {
          const double v = stab[tid];
          if (v != 0.0) atomicAdd(&ld.bsums[((long)(nbase + (tid >> 1)) * PG_STAT_SLOTS + gslot) * 2 + (tid & 1)], v);
        }
      } else if (plain) big_scatter_tile<TM, TN, true, false>(acc, T, rows, wm0, lane, ld, cval, mfw, p.Gy * p.Gx, p.M, p.N);
      else big_scatter_tile<TM, TN, false, false>(acc, T, rows, wm0, lane, ld, cval, mfw, p.Gy * p.Gx, p.M, p.N);
    } else {
#pragma unroll
      for (int h = 0; h < TM / 2; ++h) {
        if (p.dst_io == 0)
          vec_scatter_64x64<TN, 0, RowB, false>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, ld, cval, p.Ho, p.Wo);
        else
          vec_scatter_64x64<TN, 2, RowB, false>(*reinterpret_cast<const f32x16(*)[2][TN]>(&acc[2 * h][0]), T, rows, wm0 + 64 * h, lane, ld, cval, p.Ho, p.Wo);
      }
    }
  }
  if (tl_on) { __syncthreads(); stamp(4); }
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
