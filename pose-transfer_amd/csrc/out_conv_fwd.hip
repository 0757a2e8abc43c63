// The generator's output convolution, forward, on the bf16 data path — ONE streaming kernel (round 5).
//
// Reference models/networks.py:228 (decoder tail): ReLU -> Conv2d(cin -> 3, k3, p1, bias) -> Tanh over the concat
// [last decoder block's normalised output | skips of encoder level 0].  Until round 4 this was four launches at bf16 storage:
// pg_materialise_bf16_norm of the block's raw output (read 2 B + write 2 B per element), a 1 x 1 contraction to the 27 (tap, channel)
// columns on the 512 x 64 MFMA tile (reads all cin channels again: 1.07 GB at batch 32 — 0.30 ms, HBM-bound with half the tile
// empty), the fp32 tap tensor (268 MB written, read back) and pg_tap_gather (bias + tanh).  GEMM-N is 27: the operand traffic is
// the whole cost.  This kernel reads every input byte once and writes only what later passes need:
//   * a workgroup owns an 8 x 32 pixel tile of one sample; for the 10 x 34 halo pixels it forms Y[pixel][(tap, co)] =
//     W[(tap, co)][:] . x[pixel][:] with v_mfma_f32_32x32x16_bf16 — a wave reads 32 pixels as they lie in memory, the A fragments
//     go through a wave-private 8 KB LDS tile (no workgroup barrier in the loop), the 27 x cin weight lives in registers as B
//     fragments (converted from the fp32 master weight in the prologue: no padded bf16 copy of it);
//   * source 0 (the raw bf16 block output) is normalised (per-sample affine, folded pg_norm_finalize as in
//     pg_materialise_bf16_norm), ReLU'd and rounded to bf16 in registers; the rounded value is the MFMA operand AND, for the
//     pixels the tile owns, is stored as the activated operand the backward pass reads (pg_out_conv_bwd_direct: weight-gradient
//     operand and ReLU derivative) — the values pg_materialise_bf16_norm wrote (+0 where that pass leaves -0);
//   * the tap sums go through LDS ([halo pixel][27] fp32, 38 KB): out = tanh(bias + sum over the 3 x 3 neighbourhood), NCHW fp32.
// HBM bytes per pixel at cin = 256 (C0 = 128): 512 read + 256 (operand) + 12 written — 1.64 GB at batch 32 (0.39 - 0.40 ms =
// 4.1 - 4.2 TB/s) against 2.7 GB / 0.56 ms before; the halo (33 % more rows read) is served by L2: tiles are dealt to the XCDs in
// contiguous runs.
#include "common.h"

namespace pg {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 ocf_bf16x8;

struct OutFwdK {
  const unsigned short* x[3];   // x[0]: raw bf16 block output [N][H][W][C[0]]; x[1], x[2]: bf16 ACTIVATED operands (C = 0: absent)
  int C[3];
  const float* aff;             // published per-sample affine (a, b) of x[0], or null with nf.sums set
  NormFold nf;
  unsigned short* op0;          // out: bf16(relu(a x0 + b)) [N][H][W][C[0]]
  const float* Wp;              // fp32 [27][cin], row = tap * 3 + co
  const float* bias;            // [3] or null
  float* out;                   // [N][3][H][W] fp32
  int N, H, W, out_act;
  int tiles_x, tiles_y, per_xcd;
};

__device__ __forceinline__ unsigned ocf_pack2(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// C0 / C1 / C2: channels of the three sources (64 or 128; C1, C2 may be 0).  Version 2 — the first version fetched the A
// fragments straight from global memory in MFMA layout (lane = pixel: 256-byte lane stride, 16 bytes used of every line a lane
// touched; the other k-steps' reads of the same lines came after the vector L1 had lost them): 435 us at batch 32, no faster than
// the chain it replaced.  Here a wave reads its 32 pixels the way they lie in memory (C / 8 lanes per pixel: whole 128- / 256-byte
// pixel rows per 8 / 16 lanes), normalises in that layout, stores the operand in that layout (16 lanes = one 256-byte row) and
// hands the values to the MFMA layout through a wave-private LDS tile (8 KB, 16-byte slots XOR-swizzled by pixel so that both
// the row-major ds_write_b128 and the pixel-major ds_read_b128 are conflict-free).
template <int C>
struct OcfSrc {
  static constexpr int CPP = C / 8;            // 16-byte chunks per pixel
  static constexpr int PPP = 64 / CPP;         // pixels per wave pass
  static constexpr int NP = C / 16;            // passes per 32-pixel group = k-steps of the source
  static __device__ __forceinline__ int sw(int pixel) { return CPP == 16 ? (pixel & 15) : ((pixel >> 1) & 7); }
};

template <int C0, int C1, int C2>
__global__ __launch_bounds__(256, 2) void out_conv_fwd_kernel(const OutFwdK p) {
  constexpr int TH = 8, TW = 32, HWD = TW + 2, NPX = (TH + 2) * HWD, NG = (NPX + 31) / 32, CT = 27;
  constexpr int CIN = C0 + C1 + C2, NK = CIN / 16;
  static_assert((C0 == 64 || C0 == 128) && (C1 == 0 || C1 == 64 || C1 == 128) && (C2 == 0 || C2 == 64 || C2 == 128), "channels");
  __shared__ float Yt[NG * 32 * CT];
  __shared__ uint4 stage_all[4][512];          // per wave: 32 pixels x 256 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  uint4* const stage = stage_all[wave];
  // tiles in contiguous runs per XCD (workgroups are dealt round-robin to the 8 XCDs): neighbouring tiles share halo rows in ONE L2
  const int b = (int)blockIdx.x;
  const int t = (b & 7) * p.per_xcd + (b >> 3);
  const int tpi = p.tiles_x * p.tiles_y;
  if (t >= p.N * tpi) return;
  const int n = t / tpi, tr = t - n * tpi;
  const int tyi = tr / p.tiles_x, txi = tr - tyi * p.tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;

  float a = 1.f, bsh = 0.f;
  if (p.nf.sums != nullptr) norm_fold_affine(p.nf, n, tr == 0 && tid == 0, a, bsh);
  else if (p.aff != nullptr) { a = p.aff[2 * n]; bsh = p.aff[2 * n + 1]; }

  // ---- B fragments: column l31 = (tap, co) row of the weight, k = 16 ks + 8 lhi .. + 8
  unsigned bf[NK][4];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    float w[8];
    if (l31 < CT) {
      const float4* wp = reinterpret_cast<const float4*>(p.Wp + (long)l31 * CIN + ks * 16 + lhi * 8);
      const float4 w0 = wp[0], w1 = wp[1];
      w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) bf[ks][e] = ocf_pack2(w[2 * e], w[2 * e + 1]);
  }

  // pixel `pi` of group g: its element index in an [N][H][W] image, or -1 outside the image / beyond the halo; `own`: inside the tile
  auto pixel_of = [&](int g, int pi, bool& own) -> long {
    const int hp = g * 32 + pi;
    const int r = hp / HWD, c = hp - r * HWD;
    const int yy = y0 + r - 1, xx = x0 + c - 1;
    const bool valid = hp < NPX && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
    own = valid && r >= 1 && r <= TH && c >= 1 && c <= TW;
    return valid ? ((long)n * p.H + yy) * p.W + xx : -1;
  };
  for (int g = wave; g < NG; g += 4) {
    typedef OcfSrc<C0> S0;
    typedef OcfSrc<(C1 ? C1 : 64)> S1;
    typedef OcfSrc<(C2 ? C2 : 64)> S2;
    // ---- all loads of the group first: NK x 16 bytes per lane in flight, whole pixel rows per 8 / 16 lanes
    uint4 v0[S0::NP], v1[C1 ? S1::NP : 1], v2[C2 ? S2::NP : 1];
    long px0[S0::NP];
    unsigned own0 = 0;
#pragma unroll
    for (int i = 0; i < S0::NP; ++i) {
      bool own;
      px0[i] = pixel_of(g, i * S0::PPP + lane / S0::CPP, own);
      own0 |= (own ? 1u : 0u) << i;
      v0[i] = px0[i] >= 0 ? *reinterpret_cast<const uint4*>(p.x[0] + px0[i] * C0 + (lane % S0::CPP) * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
    if constexpr (C1 != 0) {
#pragma unroll
      for (int i = 0; i < S1::NP; ++i) {
        bool own;
        const long px = pixel_of(g, i * S1::PPP + lane / S1::CPP, own);
        v1[i] = px >= 0 ? *reinterpret_cast<const uint4*>(p.x[1] + px * C1 + (lane % S1::CPP) * 8) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    if constexpr (C2 != 0) {
#pragma unroll
      for (int i = 0; i < S2::NP; ++i) {
        bool own;
        const long px = pixel_of(g, i * S2::PPP + lane / S2::CPP, own);
        v2[i] = px >= 0 ? *reinterpret_cast<const uint4*>(p.x[2] + px * C2 + (lane % S2::CPP) * 8) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // ---- source 0: normalise, ReLU, round in the memory layout; the owner stores the operand; into the wave's LDS tile
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < S0::NP; ++i) {
      const unsigned w[4] = {v0[i].x, v0[i].y, v0[i].z, v0[i].w};
      unsigned o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = fmaxf(fmaf(__uint_as_float(w[e] << 16), a, bsh), 0.f);
        const float hi = fmaxf(fmaf(__uint_as_float(w[e] & 0xffff0000u), a, bsh), 0.f);
        o[e] = px0[i] >= 0 ? ocf_pack2(lo, hi) : 0u;       // (pixels outside the image are zeros AFTER the affine: zero padding of the operand)
      }
      const uint4 r = make_uint4(o[0], o[1], o[2], o[3]);
      if ((own0 >> i) & 1u) *reinterpret_cast<uint4*>(p.op0 + px0[i] * C0 + (lane % S0::CPP) * 8) = r;
      const int pi = i * S0::PPP + lane / S0::CPP;
      stage[pi * S0::CPP + ((lane % S0::CPP) ^ S0::sw(pi))] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ks = 0; ks < S0::NP; ++ks) {
      const uint4 f = stage[l31 * S0::CPP + ((2 * ks + lhi) ^ S0::sw(l31))];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ocf_bf16x8, f), __builtin_bit_cast(ocf_bf16x8, bf[ks]), acc, 0, 0, 0);
    }
    // ---- sources 1, 2: activated operands already (LDS accesses of one wave execute in order: the tile is reused)
    if constexpr (C1 != 0) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < S1::NP; ++i) {
        const int pi = i * S1::PPP + lane / S1::CPP;
        stage[pi * S1::CPP + ((lane % S1::CPP) ^ S1::sw(pi))] = v1[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int ks = 0; ks < S1::NP; ++ks) {
        const uint4 f = stage[l31 * S1::CPP + ((2 * ks + lhi) ^ S1::sw(l31))];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ocf_bf16x8, f), __builtin_bit_cast(ocf_bf16x8, bf[S0::NP + ks]), acc, 0, 0, 0);
      }
    }
    if constexpr (C2 != 0) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < S2::NP; ++i) {
        const int pi = i * S2::PPP + lane / S2::CPP;
        stage[pi * S2::CPP + ((lane % S2::CPP) ^ S2::sw(pi))] = v2[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int ks = 0; ks < S2::NP; ++ks) {
        const uint4 f = stage[l31 * S2::CPP + ((2 * ks + lhi) ^ S2::sw(l31))];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ocf_bf16x8, f), __builtin_bit_cast(ocf_bf16x8, bf[S0::NP + (C1 ? S1::NP : 0) + ks]), acc,
                                                      0, 0, 0);
      }
    }
    // accumulator register q of lane (l31, lhi) = pixel row (q & 3) + 8 (q >> 2) + 4 lhi of the group, column l31
    if (l31 < CT) {
#pragma unroll
      for (int q = 0; q < 16; ++q) Yt[(g * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi) * CT + l31] = acc[q];
    }
  }
  __syncthreads();

  const float b0 = p.bias ? p.bias[0] : 0.f, b1 = p.bias ? p.bias[1] : 0.f, b2 = p.bias ? p.bias[2] : 0.f;
  const long plane = (long)p.H * p.W;
  static_assert(TH * TW == 256, "one output pixel per thread");
  {
    const int ty = tid >> 5, tx = tid & 31;
    const int y = y0 + ty, x = x0 + tx;
    if (y < p.H && x < p.W) {
      float s0 = b0, s1 = b1, s2 = b2;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const float* yq = Yt + ((ty + r) * HWD + tx + s) * CT + (r * 3 + s) * 3;
          s0 += yq[0]; s1 += yq[1]; s2 += yq[2];
        }
      if (p.out_act == PG_OUT_TANH) { s0 = tanhf(s0); s1 = tanhf(s1); s2 = tanhf(s2); }
      float* ob = p.out + (long)n * 3 * plane + (long)y * p.W + x;
      ob[0] = s0; ob[plane] = s1; ob[2 * plane] = s2;
    }
  }
}

}  // namespace pg

using namespace pg;

extern "C" int pg_out_conv_fwd_fused(const void* x0_bf16, int32_t C0, const float* aff, const double* sums, const float* gamma,
                                     const float* beta, int64_t L, float eps, float* mr, float* aff_out, void* op0_bf16,
                                     const void* x1_bf16, int32_t C1, const void* x2_bf16, int32_t C2, const float* W27,
                                     const float* bias, int32_t N, int32_t H, int32_t W, int32_t out_act, float* out, void* stream) {
  PG_REQUIRE(x0_bf16 && op0_bf16 && W27 && out && N > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && C2 >= 0,
             "pg_out_conv_fwd_fused: bad arguments");
  const int combo = (C0 == 128 && C1 == 64 && C2 == 64) ? 1 : (C0 == 128 && C1 == 64 && C2 == 0) ? 2 : (C0 == 64 && C1 == 64 && C2 == 0) ? 3 : 0;
  PG_REQUIRE(combo != 0 && (C1 == 0 || x1_bf16) && (C2 == 0 || x2_bf16),
             "pg_out_conv_fwd_fused: source channels (%d, %d, %d) are not one of (128, 64, 64), (128, 64, 0), (64, 64, 0)", C0, C1, C2);
  PG_REQUIRE(aff == nullptr || sums == nullptr, "pg_out_conv_fwd_fused: pass either the published affine or the statistics of the norm layer");
  PG_REQUIRE(sums == nullptr || (gamma && beta && mr && aff_out && L > 0), "pg_out_conv_fwd_fused: incomplete norm-fold arguments");
  PG_REQUIRE((((size_t)x0_bf16 | (size_t)op0_bf16 | (size_t)x1_bf16 | (size_t)x2_bf16 | (size_t)W27) & 15) == 0,
             "pg_out_conv_fwd_fused: 16-byte aligned pointers required");
  OutFwdK k;
  memset(&k, 0, sizeof(k));
  k.x[0] = reinterpret_cast<const unsigned short*>(x0_bf16); k.C[0] = C0;
  k.x[1] = reinterpret_cast<const unsigned short*>(C1 ? x1_bf16 : x0_bf16); k.C[1] = C1;
  k.x[2] = reinterpret_cast<const unsigned short*>(C2 ? x2_bf16 : x0_bf16); k.C[2] = C2;
  k.aff = aff;
  k.nf.sums = sums; k.nf.gamma = gamma; k.nf.beta = beta; k.nf.L = (long)L; k.nf.eps = eps; k.nf.mr = mr; k.nf.aff = aff_out;
  k.op0 = reinterpret_cast<unsigned short*>(op0_bf16);
  k.Wp = W27; k.bias = bias; k.out = out; k.N = N; k.H = H; k.W = W; k.out_act = out_act;
  k.tiles_x = (W + 31) / 32; k.tiles_y = (H + 7) / 8;
  const long tiles = (long)N * k.tiles_x * k.tiles_y;
  PG_REQUIRE(tiles < (1L << 28), "pg_out_conv_fwd_fused: too many tiles");
  k.per_xcd = (int)((tiles + 7) / 8);
  const dim3 grid((unsigned)(k.per_xcd * 8));
  if (combo == 1) PG_KLAUNCH((out_conv_fwd_kernel<128, 64, 64>), grid, dim3(256), 0, (hipStream_t)stream, k);
  else if (combo == 2) PG_KLAUNCH((out_conv_fwd_kernel<128, 64, 0>), grid, dim3(256), 0, (hipStream_t)stream, k);
  else PG_KLAUNCH((out_conv_fwd_kernel<64, 64, 0>), grid, dim3(256), 0, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_out_conv_fwd_fused");
  return 0;
}
