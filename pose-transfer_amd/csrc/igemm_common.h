// Shared pieces of the implicit-GEMM convolution kernels (igemm_conv.hip: fp32 / operand-precision modes and the 128-wide
// bf16 data-path kernel; igemm_bf16.hip: the 256-wide bf16 data-path kernel): the kernel argument, per-row tile info, the
// constant tables and the row-major 16-byte epilogues.
#pragma once
#include "common.h"

namespace pg {

constexpr int BK = 32;
constexpr int MAXTAP = 16;
enum { A_VEC = 0, A_SCALAR = 1 };
enum { B_NT = 0, B_NN = 1, B_SCALAR = 2 };

struct ConvK {
  pg_src_t src[PG_MAX_SRC];
  int nsrc, Ctot;
  int cstart[PG_MAX_SRC + 1];
  int N, Hi, Wi, act;
  int Gy, Gx, so, si, Ho, Wo, M;
  int nphase;
  int phy[4], phx[4], ntap[4];
  signed char dy[4][MAXTAP], dx[4][MAXTAP];
  unsigned char wtap[4][MAXTAP];
  const float* W;
  int wCout, wCin, w_transposed, n_off, n_cnt;
  int ksplit;
  int epilogue, out_act;
  float* out;
  const float* bias;
  long oN, oC, oH, oW;
  pg_dst_t dst[PG_MAX_SRC];
  int ndst;
  int dstart[PG_MAX_SRC + 1];
  double* stats;            // epilogue 0, ksplit 1: per-sample (sum, sum of squares) of the stored values, or null
  int dst_uniform;          // every destination's C is a multiple of 32 (wave-uniform descriptor in the scatter)
  // batched-tap GEMM (bf16 weight gradient): blockIdx.z / ksplit selects one of `gtaps` independent products that
  // differ only in operand / output base offsets (a_off, w_off in BYTES; o_off in floats)
  int gtaps;
  long a_off[MAXTAP], w_off[MAXTAP], o_off[MAXTAP];
  int xcd_swizzle;          // bf16 data path: the N tiles of one M tile run back-to-back on ONE XCD (shared L2)
  int vec_dst;              // epilogue 1: every destination has C % 4 == 0, 16-byte aligned pointers, < 2^32 elements
  int vec_out;              // epilogue 0 / partial tiles: dense [pixel][n_cnt] rows, n_cnt % 4 == 0, 16-byte aligned base
  int out_bf16;             // epilogue 0: `out` is a bf16 NHWC tensor (bf16 STORAGE on the bf16 data path; statistics still from fp32)
  int dst_io;               // epilogue 1: 0 = every destination fp32, 1 = every gradient and forward tensor bf16, 2 = mixed (flags)
  float* part;              // split-K with a workspace: split s stores its plain partial tile at part + s*part_stride,
  long part_stride;         // laid out [pixel = (n*Ho+oy)*Wo+ox][n_cnt]; splitk_fixup_kernel reduces and applies the epilogue
};

struct RowInfo {   // per M-row of the block tile, built once in LDS (12 bytes)
  int n;           // sample index, -1 = row outside the problem
  short iy, ix;    // input base coordinate (q*si)
  short oy, ox;    // output coordinate
};

// All global loads of the K loop are UNCONDITIONAL straight-line code (a load inside a branch makes the compiler drain
// vmcnt at the join, exposing the full memory latency every K tile).  Rows without a dropout mask read this table.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wc99-designator"
static __device__ __attribute__((aligned(16))) const float kOnes[2048] = {[0 ... 2047] = 1.0f};
static __device__ __attribute__((aligned(16))) const float kZeros[4096] = {};      // source of zero rows for the LDS-DMA loaders
static __device__ __attribute__((aligned(16))) const float kIdentAff[2] = {1.0f, 0.0f};
#pragma clang diagnostic pop

// f32 MFMA runs on the SIMD's fp32 lanes: every VALU instruction of a co-resident wave steals matrix throughput
// (measured: an idle partner leaves the MFMA+fetch loop at 139 TFLOP/s, an active loader partner at 100).  So the
// loaders below are written for minimum VALU count: 32-bit byte offsets against wave-uniform bases (one v_add per
// row per tile instead of 64-bit pointer arithmetic), the activation picked once per tile, masks only when present,
// and zero padding through an affine of (0,0) instead of per-element selects.
// ---- low-precision operand modes (PREC template parameter): 1 = bf16 MFMA operands (fp32 storage / accumulate),
// 2 = "bf16x3": every fp32 operand is split as hi + lo (two bf16) and a*b ~ ah*bh + ah*bl + al*bh on the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate) — ~2^-16 relative product error, fp32-class results.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {      // RNE, lo -> bits 0..15
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf16_lo_f32(unsigned packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi_f32(unsigned packed) { return __uint_as_float(packed & 0xffff0000u); }

// ---- bf16 STORAGE helpers (round 3): 4 consecutive elements of a tensor that is either fp32 (16 bytes) or bf16 (8 bytes);
// `idx` is the ELEMENT index of the first one.  The branch is wave-uniform (a property of the launch).
__device__ __forceinline__ float4 ld4_any(const void* base, unsigned idx, bool bf) {
  if (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    return make_float4(bf16_lo_f32(u.x), bf16_hi_f32(u.x), bf16_lo_f32(u.y), bf16_hi_f32(u.y));
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
__device__ __forceinline__ void st4_any(void* base, unsigned long idx, bool bf, float4 v) {
  if (bf) {
    uint2 u;
    u.x = pack_bf16(v.x, v.y); u.y = pack_bf16(v.z, v.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx) = u;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = v;
  }
}

template <int OFF>
__device__ __forceinline__ void lds_read128(f32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_read32(float& v, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

// rare path of the fused statistics (a wave tile spanning more than two samples: only the 4x4 / 8x8 layers)
static __device__ __noinline__ void stat_spill(double* stats, int n, float g) {
  atomicAdd(&stats[(long)n * PG_STAT_SLOTS * 2], (double)g);
  atomicAdd(&stats[(long)n * PG_STAT_SLOTS * 2 + 1], (double)g * (double)g);
}

// Row-major 16-byte stores of a wave's 64x64 accumulator sub-tile (2x2 MFMA tiles): the MFMA layout gives a lane ONE
// column and 16 scattered rows per tile (4-byte stores, 64 store instructions per wave); through a wave-private LDS
// tile (32 rows x 64 columns, pitch 68) a lane gets 4 consecutive columns of a row: 16 float4 stores per wave, a full
// 256-byte row segment per 16 lanes.  Used for dense [pixel][n_cnt] destinations (forward output incl. bias and the
// fused statistics, split-K partial tiles).
template <int TN_, bool OB = false, typename RowT = RowInfo>
__device__ __forceinline__ void vec_store_64x64(const f32x16 (&acc)[2][TN_], float* T, const RowT* rows, int wm0, int lane,
                                                float* obase, int n_cnt, int Ho, int Wo, int ngc, float4 bv,
                                                bool do_stats, int stat_n0, float (&st_s)[2], float (&st_q)[2],
                                                double* stats) {
  constexpr int PITCH = 32 * TN_ + 4, LPR = 8 * TN_, RPP = 64 / LPR;     // lanes per row, rows per pass
  const int l31 = lane & 31, lhi = lane >> 5;
  const int rsel = lane / LPR, c4 = (lane % LPR) * 4;
  const bool cval = ngc < n_cnt;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhi) * PITCH + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // every row pass's LDS reads (row table + transposed values) are issued as ONE batch, then the stores go out back to
    // back: a pass that reads its row, branches on it, reads its values and stores is three dependent LDS round trips, and
    // the 8 passes of a half cost ~2.5 us per wave with one workgroup per CU (round 3, tools/conv_timeline.py)
    constexpr int NP = 32 / RPP;
    int rn[NP], roy[NP], rox[NP];
    float4 v[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int row = it * RPP + rsel;
      const RowT ri = rows[wm0 + i * 32 + row];
      rn[it] = ri.n; roy[it] = ri.oy; rox[it] = ri.ox;
      v[it] = *reinterpret_cast<const float4*>(&T[row * PITCH + c4]);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const bool ok = rn[it] >= 0 && cval;
      v[it].x += bv.x; v[it].y += bv.y; v[it].z += bv.z; v[it].w += bv.w;
      if (ok) st4_any(obase, (unsigned long)((long)((rn[it] * Ho + roy[it]) * Wo + rox[it]) * n_cnt + ngc), OB, v[it]);
      if (do_stats) {
        const float s4 = (v[it].x + v[it].y) + (v[it].z + v[it].w);
        const float q4 = fmaf(v[it].x, v[it].x, fmaf(v[it].y, v[it].y, fmaf(v[it].z, v[it].z, v[it].w * v[it].w)));
        const int dn = rn[it] - stat_n0;
        st_s[0] += (ok && dn == 0) ? s4 : 0.f; st_q[0] += (ok && dn == 0) ? q4 : 0.f;
        st_s[1] += (ok && dn == 1) ? s4 : 0.f; st_q[1] += (ok && dn == 1) ? q4 : 0.f;
        if (ok && dn > 1) {
          stat_spill(stats, rn[it], v[it].x); stat_spill(stats, rn[it], v[it].y);
          stat_spill(stats, rn[it], v[it].z); stat_spill(stats, rn[it], v[it].w);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// The data-gradient scatter with the same row-major re-layout: a lane owns 4 consecutive columns of a row, so the forward
// value / mask / previous gradient / result of 4 elements move as ONE 16-byte access each (the MFMA-layout version
// issues 3 loads + 1 store of 4 bytes per element).  The destination descriptor is per LANE (a row of 64 columns may
// cover two destinations) and is passed in already selected; absent forward / mask / accumulate inputs read a
// per-lane-constant dummy address so that every load stays unconditional.
struct LaneDst {
  float* gradp; const float* fwdp; const float* affp; const float* maskp;
  int C, c, affmul; float dslope; bool has_fwd, has_mask, accum;
  bool grad_bf16 = false, fwd_bf16 = false;      // bf16 STORAGE of the gradient / forward tensor (pg_dst_t.flags)
  double* bsums = nullptr;                       // (round 4) fused sums of the following norm backward (pg_dst_t.bsums), or null
};
// (round 4) per-lane partial sums (sum r, sum r * f) of the values a scatter stores, for the two samples n0 / n0 + 1 of a wave's
// 64 rows (pg_dst_t.bsums: the norm backward that reads this gradient next); rows of later samples add straight to memory
struct BsAcc { int n0; float s[2], q[2]; };
// IO: 0 = fp32 tensors, 1 = gradient AND forward tensor in bf16 STORAGE (compile-time: the batched loads stay straight-line
// code), 2 = per-destination run-time flags (mixed launches; the loads sit under wave-uniform branches)
template <int TN_, int IO = 0, typename RowT = RowInfo, bool PIPE = true, bool BS = false>
__device__ __forceinline__ void vec_scatter_64x64(const f32x16 (&acc)[2][TN_], float* T, const RowT* rows, int wm0, int lane,
                                                  const LaneDst& d, bool cval, int Ho, int Wo, BsAcc* bs = nullptr) {
  static_assert(!BS || PIPE, "fused sums: pipelined order only");
  const bool gbf = IO == 1 ? true : (IO == 0 ? false : d.grad_bf16);
  const bool fbf = IO == 1 ? true : (IO == 0 ? false : d.fwd_bf16);
  constexpr int PITCH = 32 * TN_ + 4, LPR = 8 * TN_, RPP = 64 / LPR;     // lanes per row, rows per pass
  const int l31 = lane & 31, lhi = lane >> 5;
  const int rsel = lane / LPR, c4 = (lane % LPR) * 4;
  if constexpr (!PIPE) {       // the round-2 order (256-row kernel's fp32 / mixed destinations: no registers to spare there)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
  #pragma unroll
      for (int j = 0; j < TN_; ++j)
  #pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhi) * PITCH + j * 32 + l31] = acc[i][j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
  #pragma unroll
      for (int h = 0; h < 32 / RPP / 4; ++h) {             // batches of four row passes: 16 loads in flight per lane
        float4 f[4], m[4], old[4], v[4];
        float2 ab[4];
        unsigned idx[4];
        bool ok[4];
  #pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (h * 4 + u) * RPP + rsel;
          const RowT ri = rows[wm0 + i * 32 + row];
          ok[u] = (ri.n >= 0) & cval;
          const int nn = ok[u] ? ri.n : 0;
          idx[u] = ok[u] ? (unsigned)((nn * Ho + ri.oy) * Wo + ri.ox) * (unsigned)d.C + (unsigned)d.c : (unsigned)d.c;
          v[u] = *reinterpret_cast<const float4*>(&T[row * PITCH + c4]);
          f[u] = ld4_any(d.fwdp, d.has_fwd ? idx[u] : (unsigned)d.c, IO == 2 ? (d.has_fwd ? fbf : gbf) : fbf);
          ab[u] = *reinterpret_cast<const float2*>(d.affp + d.affmul * nn);
          m[u] = *reinterpret_cast<const float4*>(d.maskp + (d.has_mask ? nn * d.C + d.c : (d.c & 511)));
          old[u] = ld4_any(d.gradp, d.accum ? idx[u] : (unsigned)d.c, gbf);
        }
  #pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float g4[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, f4[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
          const float m4[4] = {m[u].x, m[u].y, m[u].z, m[u].w}, o4[4] = {old[u].x, old[u].y, old[u].z, old[u].w};
          float r4[4];
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaf(f4[e], ab[u].x, ab[u].y) * m4[e];
            r4[e] = fmaf(g4[e] * m4[e], act_grad_s(z, d.dslope), d.accum ? o4[e] : 0.f);
          }
          if (ok[u]) st4_any(d.gradp, idx[u], gbf, make_float4(r4[0], r4[1], r4[2], r4[3]));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // Batches of four row passes (16 loads in flight per lane).  The loads of batch b+1 are issued BEFORE the stores of batch b
  // (whose results wait in 16 registers): gfx950 counts loads and stores in ONE vmcnt, so a load waited for behind a store
  // also waits for that store's acknowledgement — the round-2 order (loads, wait, compute, stores per batch) paid a store
  // round trip per batch (tools/conv_timeline.py, round 3).
  constexpr int NB = 32 / RPP / 4;
  struct Batch {
    float4 f[4], m[4], old[4];
    float2 ab[4];
    unsigned idx[4];
    unsigned ok;
    int n[BS ? 4 : 1];
  };
  auto issue = [&](int i, int h, Batch& L) {
    L.ok = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = (h * 4 + u) * RPP + rsel;
      const RowT ri = rows[wm0 + i * 32 + row];
      const bool ok = (ri.n >= 0) & cval;
      const int nn = ok ? ri.n : 0;
      L.ok |= (ok ? 1u : 0u) << u;
      if constexpr (BS) L.n[u] = nn;
      L.idx[u] = ok ? (unsigned)((nn * Ho + ri.oy) * Wo + ri.ox) * (unsigned)d.C + (unsigned)d.c : (unsigned)d.c;
      L.f[u] = ld4_any(d.fwdp, d.has_fwd ? L.idx[u] : (unsigned)d.c, IO == 2 ? (d.has_fwd ? fbf : gbf) : fbf);
      L.ab[u] = *reinterpret_cast<const float2*>(d.affp + d.affmul * nn);
      L.m[u] = *reinterpret_cast<const float4*>(d.maskp + (d.has_mask ? nn * d.C + d.c : (d.c & 511)));
      L.old[u] = ld4_any(d.gradp, d.accum ? L.idx[u] : (unsigned)d.c, gbf);
    }
  };
  Batch cur;
  issue(0, 0, cur);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhi) * PITCH + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      float4 res[4];
      unsigned sidx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (h * 4 + u) * RPP + rsel;
        const float4 v = *reinterpret_cast<const float4*>(&T[row * PITCH + c4]);
        const float g4[4] = {v.x, v.y, v.z, v.w}, f4[4] = {cur.f[u].x, cur.f[u].y, cur.f[u].z, cur.f[u].w};
        const float m4[4] = {cur.m[u].x, cur.m[u].y, cur.m[u].z, cur.m[u].w};
        const float o4[4] = {cur.old[u].x, cur.old[u].y, cur.old[u].z, cur.old[u].w};
        float r4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = fmaf(f4[e], cur.ab[u].x, cur.ab[u].y) * m4[e];
          r4[e] = fmaf(g4[e] * m4[e], act_grad_s(z, d.dslope), d.accum ? o4[e] : 0.f);
        }
        res[u] = make_float4(r4[0], r4[1], r4[2], r4[3]);
        sidx[u] = cur.idx[u];
        if constexpr (BS) {
          const bool okb = ((cur.ok >> u) & 1u) != 0;
          const float s4 = (r4[0] + r4[1]) + (r4[2] + r4[3]);
          const float q4 = fmaf(r4[0], f4[0], fmaf(r4[1], f4[1], fmaf(r4[2], f4[2], r4[3] * f4[3])));
          const int dn = cur.n[u] - bs->n0;
          bs->s[0] += (okb && dn == 0) ? s4 : 0.f; bs->q[0] += (okb && dn == 0) ? q4 : 0.f;
          bs->s[1] += (okb && dn == 1) ? s4 : 0.f; bs->q[1] += (okb && dn == 1) ? q4 : 0.f;
          if (okb && dn > 1) {       // host: >= 64 positions per sample, so a wave's 64 rows hold at most two samples — never taken
            atomicAdd(&d.bsums[(long)cur.n[u] * PG_STAT_SLOTS * 2], (double)s4);
            atomicAdd(&d.bsums[(long)cur.n[u] * PG_STAT_SLOTS * 2 + 1], (double)q4);
          }
        }
      }
      const unsigned sok = cur.ok;
      if (h + 1 < NB) issue(i, h + 1, cur);
      else if (i == 0) issue(1, 0, cur);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if ((sok >> u) & 1u) st4_any(d.gradp, sidx[u], gbf, res[u]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace pg
