// Shared pieces of the 256-row bf16 convolution kernels (igemm_bf16.hip: one A tile per tap; igemm_bf16_pair.hip, round 4: one A
// tile per PAIR of taps): LDS access helpers, the row table entry, the forward-store / data-gradient-scatter epilogues.
#pragma once
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
                                                float (&st_s)[2], float (&st_q)[2], double* stats, const float* bias = nullptr) {
  if constexpr (OB) {
    // bf16 rows (round 4): EIGHT columns per lane — 16-byte stores, half the store and row-table instructions of the 4-column
    // form below (whose bf16 stores were 8 bytes); `bias` = the launch's bias vector or null (the caller's bv is per 4 columns)
    constexpr int PITCH = 32 * TN_ + 4, LPR = 4 * TN_, RPP = 64 / LPR, NP = 32 / RPP;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int rsel = lane / LPR, c8 = (lane % LPR) * 8;
    const int ncb = ngc - (lane % (8 * TN_)) * 4;            // the wave's first column
    const int ng8 = ncb + c8;
    const bool cval = ng8 < n_cnt;
    char* const ob = reinterpret_cast<char*>(obase) + (size_t)ng8 * 2;
    const unsigned rowb = (unsigned)n_cnt * 2u;
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (bias != nullptr && cval) { b0 = *reinterpret_cast<const float4*>(bias + ng8); b1 = *reinterpret_cast<const float4*>(bias + ng8 + 4); }
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
      float4 va[NP], vb[NP];
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int row = it * RPP + rsel;
        ro[it] = *reinterpret_cast<const int2*>(&rows[wm0 + i * 32 + row]);            // (n, opix)
        va[it] = *reinterpret_cast<const float4*>(&T[row * PITCH + c8]);
        vb[it] = *reinterpret_cast<const float4*>(&T[row * PITCH + c8 + 4]);
      }
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const bool ok = (ro[it].x >= 0) & cval;
        if (has_bias) {
          va[it].x += b0.x; va[it].y += b0.y; va[it].z += b0.z; va[it].w += b0.w;
          vb[it].x += b1.x; vb[it].y += b1.y; vb[it].z += b1.z; vb[it].w += b1.w;
        }
        if (ok)
          *reinterpret_cast<uint4*>(ob + (size_t)(unsigned)ro[it].y * rowb) =
              make_uint4(pack_bf16(va[it].x, va[it].y), pack_bf16(va[it].z, va[it].w), pack_bf16(vb[it].x, vb[it].y), pack_bf16(vb[it].z, vb[it].w));
        if (do_stats) {
          float s4 = ((va[it].x + va[it].y) + (va[it].z + va[it].w)) + ((vb[it].x + vb[it].y) + (vb[it].z + vb[it].w));
          float q4 = fmaf(va[it].x, va[it].x, fmaf(va[it].y, va[it].y, fmaf(va[it].z, va[it].z, va[it].w * va[it].w)));
          q4 = fmaf(vb[it].x, vb[it].x, fmaf(vb[it].y, vb[it].y, fmaf(vb[it].z, vb[it].z, fmaf(vb[it].w, vb[it].w, q4))));
          if (has_bias) { s4 = ok ? s4 : 0.f; q4 = ok ? q4 : 0.f; }
          if (one_sample) { tot_s += s4; tot_q += q4; }
          else {
            const int dn = ro[it].x - n_lo;
            st_s[0] += (ok && dn == 0) ? s4 : 0.f; st_q[0] += (ok && dn == 0) ? q4 : 0.f;
            st_s[1] += (ok && dn == 1) ? s4 : 0.f; st_q[1] += (ok && dn == 1) ? q4 : 0.f;
            if (ok && dn > 1) {
              const float e8[8] = {va[it].x, va[it].y, va[it].z, va[it].w, vb[it].x, vb[it].y, vb[it].z, vb[it].w};
#pragma unroll
              for (int q = 0; q < 8; ++q) stat_spill(stats, ro[it].x, e8[q]);
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (one_sample) { st_s[0] += tot_s; st_q[0] += tot_q; }
    return;
  }
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

// (round 4) The SIMPLE scatter with EIGHT columns per lane: 16-byte loads of the forward values and 16-byte stores of the
// gradient — half the global and row-table instructions of the 4-column form.  Caller: the wave's 64 columns lie in ONE destination
// (wave-uniform descriptor), no mask, no accumulation, every column valid; `c0` = the wave's first column inside the destination.
template <int TM_, int TN_, bool BS, bool ACC>
__device__ __forceinline__ void big_scatter_tile8(const f32x16 (&acc)[TM_][TN_], float* T, const RowB* rows, int wm0, int lane,
                                                  const LaneDst& d, int c0, int m_first, int gg, int M, int N,
                                                  double* stab = nullptr, int nbase = 0, int stat_n = 0, int gslot = 0) {
  constexpr int PITCH = 32 * TN_ + 4, LPR = 4 * TN_, RPP = 64 / LPR, NP = 32 / RPP;
  static_assert(NP == 4, "big_scatter_tile8: 64-column wave tiles");
  const int l31 = lane & 31, lhi = lane >> 5;
  const int rsel = lane / LPR, c8 = (lane % LPR) * 8;
  const unsigned dc = (unsigned)(c0 + c8);
  struct Half {
    uint4 fb[NP];
    uint4 ob[ACC ? NP : 1];      // ACC: the gradient already stored there (the encoders' skip gradients are added to)
    float2 ab[2];
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
      const bool ok = ro.x >= 0;
      const unsigned idx = ok ? (unsigned)ro.y * (unsigned)d.C + dc : dc;
      L.ok |= (ok ? 1u : 0u) << it;
      L.fb[it] = *reinterpret_cast<const uint4*>(fwd16 + (d.has_fwd ? idx : dc));
      if constexpr (ACC) L.ob[it] = *reinterpret_cast<const uint4*>(grad16 + idx);
    }
    L.ab[0] = *reinterpret_cast<const float2*>(d.affp + d.affmul * L.nlo);
    L.ab[1] = *reinterpret_cast<const float2*>(d.affp + d.affmul * nhi);
  };
  int run_n = -1;
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
    uint4 res[NP];
    unsigned oidx[NP];
    float4 va[NP], vb[NP];
    int2 ro[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int row = u * RPP + rsel;
      va[u] = *reinterpret_cast<const float4*>(&T[row * PITCH + c8]);
      vb[u] = *reinterpret_cast<const float4*>(&T[row * PITCH + c8 + 4]);
      ro[u] = *reinterpret_cast<const int2*>(&rows[wm0 + hh * 32 + row]);
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const bool hi = ro[u].x > cur.nlo;
      const float a = hi ? cur.ab[1].x : cur.ab[0].x, bb = hi ? cur.ab[1].y : cur.ab[0].y;
      const float g8[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
      const uint4 fq = cur.fb[u];
      const float f8[8] = {bf16_lo_f32(fq.x), bf16_hi_f32(fq.x), bf16_lo_f32(fq.y), bf16_hi_f32(fq.y),
                           bf16_lo_f32(fq.z), bf16_hi_f32(fq.z), bf16_lo_f32(fq.w), bf16_hi_f32(fq.w)};
      float r8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r8[e] = g8[e] * act_grad_s(fmaf(f8[e], a, bb), d.dslope);
      if constexpr (ACC) {
        const uint4 oq = cur.ob[u];
        const float o8[8] = {bf16_lo_f32(oq.x), bf16_hi_f32(oq.x), bf16_lo_f32(oq.y), bf16_hi_f32(oq.y),
                             bf16_lo_f32(oq.z), bf16_hi_f32(oq.z), bf16_lo_f32(oq.w), bf16_hi_f32(oq.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) r8[e] += o8[e];
      }
      res[u] = make_uint4(pack_bf16(r8[0], r8[1]), pack_bf16(r8[2], r8[3]), pack_bf16(r8[4], r8[5]), pack_bf16(r8[6], r8[7]));
      oidx[u] = (unsigned)ro[u].y * (unsigned)d.C + dc;
      if constexpr (BS) {
        const float okf = ((cur.ok >> u) & 1u) ? 1.f : 0.f;
        float s8 = 0.f, q8 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s8 += r8[e]; q8 = fmaf(r8[e], f8[e], q8); }
        s8 *= okf; q8 *= okf;
        run_s += hi ? 0.f : s8; run_q += hi ? 0.f : q8;
        hi_s += hi ? s8 : 0.f; hi_q += hi ? q8 : 0.f;
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
      if ((okh >> it) & 1u) *reinterpret_cast<uint4*>(grad16 + oidx[it]) = res[it];
  };
  if constexpr (!ACC) {
    Half ha, hb, hc;              // few registers per half: two halves of loads in flight
    issue(0, ha);
    if (TM_ > 1) issue(1, hb);
    step(0, ha, hc, 2);
    if (TM_ > 1) step(1, hb, ha, 3);
    if (TM_ > 2) step(2, hc, hb, -1);
    if (TM_ > 3) step(3, ha, hb, -1);
  } else {
    Half ha, hb;
    issue(0, ha);
    step(0, ha, hb, 1);
    if (TM_ > 1) step(1, hb, ha, 2);
    if (TM_ > 2) step(2, ha, hb, 3);
    if (TM_ > 3) step(3, hb, ha, -1);
  }
  if constexpr (BS) flush(run_n, run_s, run_q);
}

// The whole epilogue of a 256-row kernel: partial tiles (split-K workspace), forward store (+ fused statistics), data-gradient
// scatter (+ fused norm-backward sums).  sel_ok: rows outside the problem may hold non-zero accumulators (tap-pair kernel: their A
// rows are shared with a neighbour) — the statistics then select on the row flag as they do with a bias.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// Tw (round 5, persistent tap-pair kernel): this wave's transposition tile when it is not the default slot at the start of the
// operand rings; after_sync: called by every thread right behind the barrier that ends the K loop (the rings are free: the
// persistent kernel issues the next tile's prologue there).
template <int TM, int TN, int STAT_OFF, int STAT_N, typename Stamp, typename Hook = NoHook>
__device__ __forceinline__ void big_epilogue(const ConvK& p, f32x16 (&acc)[TM][TN], char* smem, const RowB* rows, int tid, int m0, int nb0,
                                             int wm0, int wn0, int bx, int by, int bz, int split, float* out_g, bool sel_ok, bool tl_on,
                                             Stamp&& stamp, float* Tw = nullptr, Hook&& after_sync = Hook()) {
  const int lane = tid & 63, wave = tid >> 6;
  __syncthreads();
  stamp(3);
  after_sync();
  float* const T = Tw != nullptr ? Tw : reinterpret_cast<float*>(smem) + wave * (32 * (32 * TN + 4));
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
    const bool has_bias = p.bias != nullptr || sel_ok;
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
                                  ngc, has_bias, bv, do_stats, n_lo, one_sample, st_s, st_q, p.stats, p.bias);
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
      // (round 4) the wave's 64 columns in ONE destination, every column valid: the 8-columns-per-lane form of the plain scatter
      const int cw0 = ld.c - (lane % (8 * TN)) * 4;           // the wave's first column inside its destination, if uniform
      bool wide = false;
      if constexpr (TN == 2) {
        const int cw0u = __builtin_amdgcn_readfirstlane(cw0);
        const unsigned long long gp = reinterpret_cast<unsigned long long>(ld.gradp), fp = reinterpret_cast<unsigned long long>(ld.fwdp);
        const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gp), ghi = __builtin_amdgcn_readfirstlane((unsigned)(gp >> 32));
        const unsigned flo = __builtin_amdgcn_readfirstlane((unsigned)fp), fhi = __builtin_amdgcn_readfirstlane((unsigned)(fp >> 32));
        wide = __builtin_amdgcn_ballot_w64(ld.has_mask) == 0 && __builtin_amdgcn_ballot_w64(cw0 != cw0u || (unsigned)gp != glo || (unsigned)(gp >> 32) != ghi || (unsigned)fp != flo ||
                                                    (unsigned)(fp >> 32) != fhi || !cval) == 0;
      }
      if (bs_on) {
        if constexpr (TN == 2) {
          if (wide && plain) big_scatter_tile8<TM, TN, true, false>(acc, T, rows, wm0, lane, ld, cw0, mfw, p.Gy * p.Gx, p.M, p.N, stab, nbase, STAT_N, gslot);
          else if (wide) big_scatter_tile8<TM, TN, true, true>(acc, T, rows, wm0, lane, ld, cw0, mfw, p.Gy * p.Gx, p.M, p.N, stab, nbase, STAT_N, gslot);
        }
        if (wide) {}
        else if (plain) big_scatter_tile<TM, TN, true, true>(acc, T, rows, wm0, lane, ld, cval, mfw, p.Gy * p.Gx, p.M, p.N, stab, nbase, STAT_N, gslot);
        else big_scatter_tile<TM, TN, false, true>(acc, T, rows, wm0, lane, ld, cval, mfw, p.Gy * p.Gx, p.M, p.N, stab, nbase, STAT_N, gslot);
        __syncthreads();
        if (tid < STAT_N * 2) {
          const double v = stab[tid];
          if (v != 0.0) atomicAdd(&ld.bsums[((long)(nbase + (tid >> 1)) * PG_STAT_SLOTS + gslot) * 2 + (tid & 1)], v);
        }
      } else if (wide) {
        if constexpr (TN == 2) {
          if (plain) big_scatter_tile8<TM, TN, false, false>(acc, T, rows, wm0, lane, ld, cw0, mfw, p.Gy * p.Gx, p.M, p.N);
          else big_scatter_tile8<TM, TN, false, true>(acc, T, rows, wm0, lane, ld, cw0, mfw, p.Gy * p.Gx, p.M, p.N);
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
