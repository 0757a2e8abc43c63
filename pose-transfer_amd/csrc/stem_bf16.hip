// First-layer ("stem") convolutions of the bf16 data path (gfx950): few NCHW fp32 input channels -> 64 NHWC channels.
//   encoder level 0 : nn.Conv2d(cin, 64, k3, s1, p1, bias)      reference models/networks.py:186   (cin = 3+P / P)
//   discriminator   : nn.Conv2d(3+2P+3, 64, k4, s2, p0, bias)   reference models/networks.py:341
// and their weight gradients (what autograd computes for loss.backward(), models/pose_gan.py:112,170).
//
// The fp32 versions (edge.hip / small_cin_wgrad.hip) feed v_mfma_f32_32x32x2f32 one K element per LDS read and run at
// 0.12 of the HBM rate they could stream at (round-2 hbm_kernels block).  On the bf16 data path the operands are rounded
// to bf16 anyway, so here the input patch of a pixel tile is staged ONCE as bf16, CHANNEL-LAST ([row][col][channels]):
//   forward : A (pixels x K) lane = pixel, its 8 consecutive k are 8 channels of one tap -> one ds_read_b128 per MFMA
//             operand straight out of the patch (no im2col tile); B = weights repacked [chunk][co][8 k] (pg_stem_pack_bf16),
//             v_mfma_f32_32x32x16_bf16.  K = taps x padded channels, split into channel groups of CG (24 / 40: pixel
//             stride 48 / 80 B, an odd number of 16-byte units -> conflict-free b128 reads) so that any Cin fits the LDS.
//   wgrad   : dW[(tap,ci)][co] = sum_pixels dY[pixel][co] * x[pixel*S + tap][ci]: K = pixels, which is the SLOW index of
//             both LDS images (dY tile [pixel][64], patch [pixel][channels]) -> both MFMA operands come from the
//             transposing read ds_read_b64_tr_b16 (lane 4r+c of a 16-lane group supplies the address of 4 channels of
//             pixel r and receives channel `lane` of pixels 0..3).  All taps share one pass over dY; accumulators stay in
//             registers across the tiles of a persistent workgroup; partial results go through a workspace.
// Supported: k3 s1 and k4 s2, 64 output channels, Cin <= 80 (conv) / Cin <= 36 (k3 wgrad) / Cin <= 72 (k4 wgrad) — P = 18
// and P = 32 key-points (BASELINE.json configs[1..4]).
#include "igemm_common.h"

namespace pg {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ unsigned short to_bf16(float v) { return (unsigned short)(pack_bf16(v, 0.f) & 0xffffu); }

struct StemK {
  pg_src_t src[PG_MAX_SRC];
  int nsrc, Ctot;
  int cstart[PG_MAX_SRC + 1];
  int N, Hi, Wi, Ho, Wo, pad;
  const unsigned short* Wp;   // conv: packed bf16 weights (pg_stem_pack_bf16)
  const float* bias;
  float* out;                 // conv: fp32 NHWC [N][Ho][Wo][64], or NULL (bf16 STORAGE: only bf16 outputs)
  unsigned short* out_bf16;   // conv, optional: bf16(act(out)) NHWC, act slope `slope` (the next layer's operand)
  float slope;
  unsigned short* obf2;       // conv, optional (round 3): further bf16 outputs bf16(act_k(out)) with slopes slope2 / slope3 —
  unsigned short* obf3;       //   slope 1 = the raw tensor in bf16 STORAGE, 0.2 / 0 = LeakyReLU / ReLU operands
  float slope2, slope3;
  const float* dY;            // wgrad: NHWC [N][Ho][Wo][64]
  int dy_bf16;                // wgrad: dY is a bf16 tensor (bf16 STORAGE)
  int ones_slot;              // wgrad (round 4): patch channel slot Ctot holds the constant 1 — its column of one tap IS the bias gradient
  float* part;                // wgrad: per-workgroup partial results [blocks][64][npad]
  int npad;
  int tiles_x, tiles_y, ntiles, ngroups;
};

// Stage channels [c_first, c_first + CH) of the virtual concat as bf16 into the channel-last patch at channel slot 0..:
// one item = (channel, patch row, patch column); lanes run along the image row (coalesced 4-byte loads); batches of U
// branch-free loads are in flight before the first LDS write.
template <int S, int PH, int PW, int ROWP, int PWH, int PS, int NT>
__device__ __forceinline__ void stem_stage_patch(const StemK& p, char* patch, int n, int iy0, int ix0, int c_first, int CH,
                                                 int tid) {
  constexpr int U = 8;
  const int c_last = min(c_first + CH, p.Ctot);
  for (int j = 0; j < p.nsrc; ++j) {
    const int lo = max(p.cstart[j], c_first), hi = min(p.cstart[j + 1], c_last);
    if (lo >= hi) continue;
    const int sC = (int)p.src[j].sC, sH = (int)p.src[j].sH, sW = (int)p.src[j].sW;
    const char* ptr = uniform_ptr(reinterpret_cast<const char*>(p.src[j].ptr + (long)n * p.src[j].sN + (long)(lo - p.cstart[j]) * sC));
    const int slot0 = lo - c_first;
    const int E = (hi - lo) * PH * PW;
    for (int e0 = tid; e0 < E; e0 += NT * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + NT * u;
        const int col = e % PW;
        const int rr = (e / PW) % PH;
        const int c = e / (PW * PH);
        const int iy = iy0 + rr, ix = ix0 + col;
        const bool ok = (e < E) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
        const int off = ok ? c * sC + iy * sH + ix * sW : 0;
        v[u] = ldg32(ptr, (long)off * 4);
        if (!ok) v[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + NT * u;
        const int col = e % PW;
        const int rr = (e / PW) % PH;
        const int c = e / (PW * PH);
        const int ci = (S == 1) ? col : (col & 1) * PWH + (col >> 1);
        if (e < E) *reinterpret_cast<unsigned short*>(patch + (rr * ROWP + ci) * PS + (slot0 + c) * 2) = to_bf16(v[u]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward.  Workgroup = 4 waves, output tile TH x 16 pixels x 64 channels; wave w owns rows 2*TMW*w .. (TMW M-tiles of two
// image rows each) and both 32-channel halves.
template <int K, int S, int CG, int TH>
__global__ __launch_bounds__(256) void stem_conv_bf16_kernel(const StemK p) {
  constexpr int TW = 16;
  constexpr int PH = (TH - 1) * S + K, PW = (TW - 1) * S + K, PWH = (PW + 1) / 2;
  constexpr int ROWP = (S == 1) ? PW : 2 * PWH;             // patch pixels per row (S = 2: even columns, then odd ones)
  constexpr int PS = CG * 2;                                // bytes per patch pixel
  constexpr int PATCH_B = (PH * ROWP * PS + 15) / 16 * 16;
  constexpr int CPT = CG / 8;                               // 8-channel chunks per tap
  constexpr int TPS = (K == 3) ? 9 : 4;                     // taps per weight stage (k3: all; k4: one filter row)
  constexpr int NST = (K == 3) ? 1 : 4;
  constexpr int NCH = TPS * CPT, NKS = (NCH + 1) / 2;       // chunks / MFMA k-steps per stage
  constexpr int W_B = 2 * NKS * 1024;                       // bytes of one weight stage: [chunk][co 64][8 k] bf16
  constexpr int TMW = TH / 8;                               // M-tiles (2 rows x 16 columns) per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const wl = smem + PATCH_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // (round 3: the filter stage is re-read per tile — 28 KB from L2 — because the epilogue's transposition tiles reuse its LDS)
  const bool resident = false;

  for (int i = tid; i < PATCH_B / 16; i += 256) reinterpret_cast<float4*>(patch)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage_w = [&](int g, int st) {
    const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.Wp) + (size_t)(g * NST + st) * W_B);
#pragma unroll
    for (int u = 0; u < (W_B / 16 + 255) / 256; ++u) {
      const int e = tid + 256 * u;
      if (e < W_B / 16) reinterpret_cast<float4*>(wl)[e] = src[e];
    }
  };
  if (resident) stage_w(0, 0);

  // lane constants: A = pixel (row 2*mt + (l31 >> 4), column l31 & 15) of the wave's first M-tile; B = (k half, channel)
  const int a_base = (((2 * TMW * wave + (l31 >> 4)) * S) * ROWP + (l31 & 15)) * PS;
  const int b_base = lhi * 1024 + l31 * 16;
  const float bias0 = p.bias ? p.bias[l31] : 0.f, bias1 = p.bias ? p.bias[32 + l31] : 0.f;

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    int b = t;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    f32x16 acc[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int g = 0; g < p.ngroups; ++g) {
      __syncthreads();                                      // every wave is done with the previous patch / filter stage
      stem_stage_patch<S, PH, PW, ROWP, PWH, PS, 256>(p, patch, n, iy0, ix0, g * CG, CG, tid);
      for (int st = 0; st < NST; ++st) {
        if (!resident) {
          if (st > 0) __syncthreads();
          stage_w(g, st);
        }
        __syncthreads();
        const int a_st = a_base + ((K == 3) ? 0 : st * ROWP * PS);      // k4: stage = filter row
#pragma unroll
        for (int kk = 0; kk < NKS; ++kk) {
          // compile-time patch offsets of chunk 2kk (lanes 0-31) and 2kk+1 (lanes 32-63)
          int off[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * kk + h;
            const int tp = (j < NCH) ? j / CPT : 0, c8 = (j < NCH) ? j % CPT : 0;
            const int r = (K == 3) ? tp / 3 : 0, s = (K == 3) ? tp % 3 : tp;
            const int ci = (S == 1) ? s : (s & 1) * PWH + (s >> 1);
            off[h] = (r * ROWP + ci) * PS + c8 * 16;
          }
          const int ao = a_st + (lhi ? off[1] : off[0]);
          const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(wl + b_base + kk * 2048);
          const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wl + b_base + kk * 2048 + 512);
#pragma unroll
          for (int i = 0; i < TMW; ++i) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(patch + ao + i * (2 * S * ROWP * PS));
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[i][1], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue (round 3): + bias, then every M-tile (32 pixels x 64 channels) goes through a wave-private LDS tile (pitch 68
    //      floats, in the filter stage's memory) and comes back as 4 consecutive channels of a pixel per lane: 16-byte fp32 /
    //      8-byte bf16 stores, a full 128-byte bf16 row per 16 lanes.  One transposition serves all outputs — the optional fp32
    //      tensor and up to three bf16(act_k(out)) tensors (raw bf16 STORAGE / the next layers' activated operands).  The
    //      channel-per-lane form cost a lane exchange + a 4-byte store per two values and output: the kernel was bound by it.
    __syncthreads();                                        // every wave is done with the filter stage
    {
      float* const T = reinterpret_cast<float*>(smem + PATCH_B) + wave * (32 * 68);
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float bj = j ? bias1 : bias0;
#pragma unroll
          for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 68 + j * 32 + l31] = acc[i][j][r] + bj;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 8 channels per lane (round 4): 16-byte bf16 stores — half the store instructions of the 4-channel form
        float4 v[4][2];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const float* tr = &T[(it * 8 + (lane >> 3)) * 68 + (lane & 7) * 8];
          v[it][0] = *reinterpret_cast<const float4*>(tr);
          v[it][1] = *reinterpret_cast<const float4*>(tr + 4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = it * 8 + (lane >> 3);
          const int oy = oy0 + 2 * (TMW * wave + i) + (m >> 4), ox = ox0 + (m & 15);
          if (oy >= p.Ho || ox >= p.Wo) continue;
          const long o = (((long)n * p.Ho + oy) * p.Wo + ox) * 64 + (lane & 7) * 8;
          if (p.out != nullptr) {
            *reinterpret_cast<float4*>(p.out + o) = v[it][0];
            *reinterpret_cast<float4*>(p.out + o + 4) = v[it][1];
          }
#pragma unroll
          for (int ob = 0; ob < 3; ++ob) {
            unsigned short* const optr = ob == 0 ? p.out_bf16 : (ob == 1 ? p.obf2 : p.obf3);
            const float oslope = ob == 0 ? p.slope : (ob == 1 ? p.slope2 : p.slope3);
            if (optr == nullptr) continue;
            const float4 a = v[it][0], b = v[it][1];
            *reinterpret_cast<uint4*>(optr + o) = make_uint4(pack_bf16(apply_act_s(a.x, oslope), apply_act_s(a.y, oslope)),
                                                             pack_bf16(apply_act_s(a.z, oslope), apply_act_s(a.w, oslope)),
                                                             pack_bf16(apply_act_s(b.x, oslope), apply_act_s(b.y, oslope)),
                                                             pack_bf16(apply_act_s(b.z, oslope), apply_act_s(b.w, oslope)));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, pipelined form (round 4) for ONE source, one channel group and k3 s1 — the generator's first layers.  The kernel above
// spends a tile's life in dependent latencies: four batches of input loads before the patch is written, the 28 KB filter
// re-read from L2 (its LDS doubles as the epilogue's transposition tiles), MFMAs, stores — 31 us per 16 x 16 tile and
// workgroup, 2.6 TB/s.  Here the filter stays resident, the transposition tiles have their own LDS (79 KB per workgroup, two
// per CU), a thread's patch elements (channel, row, column) are the same for every tile, so their offsets live in registers,
// and the NEXT tile's input is loaded into registers before the current tile's MFMAs and stores.
template <int CG, int TH>
__global__ __launch_bounds__(256, 2) void stem_conv_bf16_pf_kernel(const StemK p) {
  constexpr int K = 3, S = 1, TW = 16;
  constexpr int PH = TH + 2, PW = TW + 2, ROWP = PW;
  constexpr int PS = CG * 2;
  constexpr int PATCH_B = (PH * ROWP * PS + 15) / 16 * 16;
  constexpr int CPT = CG / 8, NCH = 9 * CPT, NKS = (NCH + 1) / 2;
  constexpr int W_B = 2 * NKS * 1024;
  constexpr int TMW = TH / 8;
  constexpr int NU = (CG * PH * PW + 255) / 256;            // patch elements per thread (upper bound: CG channels)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const wl = smem + PATCH_B;
  float* const T = reinterpret_cast<float*>(smem + PATCH_B + W_B) + (threadIdx.x >> 6) * (32 * 68);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  for (int i = tid; i < PATCH_B / 16; i += 256) reinterpret_cast<float4*>(patch)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const float4* src = reinterpret_cast<const float4*>(p.Wp);
    for (int e = tid; e < W_B / 16; e += 256) reinterpret_cast<float4*>(wl)[e] = src[e];
  }
  const int sC = (int)p.src[0].sC, sH = (int)p.src[0].sH, sW = (int)p.src[0].sW;
  const int E = p.Ctot * PH * PW;
  // per-thread element table: global offset relative to the tile origin, LDS byte offset, (row, column)
  int rel[NU];
  unsigned meta[NU];                                        // LDS byte offset | row << 16 | column << 24
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int e = tid + 256 * u;
    const int col = e % PW, rr = (e / PW) % PH, c = e / (PW * PH);
    rel[u] = c * sC + rr * sH + col * sW;
    meta[u] = (unsigned)((rr * ROWP + col) * PS + c * 2) | ((unsigned)rr << 16) | ((unsigned)col << 24);
  }
  static_assert(PH * ROWP * PS < 65536, "stem conv: LDS offsets in 16 bits");
  const int a_base = ((2 * TMW * wave + (l31 >> 4)) * ROWP + (l31 & 15)) * PS;
  const int b_base = lhi * 1024 + l31 * 16;
  const float bias0 = p.bias ? p.bias[l31] : 0.f, bias1 = p.bias ? p.bias[32 + l31] : 0.f;

  float v[NU];
  unsigned okm = 0;
  auto prefetch = [&](int t) {
    int b = t;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int iy0 = ty * TH - p.pad, ix0 = tx * TW - p.pad;
    const char* ptr = uniform_ptr(reinterpret_cast<const char*>(p.src[0].ptr + (long)n * p.src[0].sN));
    const int org = iy0 * sH + ix0 * sW;
    okm = 0;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int iy = iy0 + (int)((meta[u] >> 16) & 0xffu), ix = ix0 + (int)(meta[u] >> 24);
      const bool ok = (tid + 256 * u < E) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      v[u] = ldg32(ptr, (long)(ok ? org + rel[u] : 0) * 4);
      okm |= (ok ? 1u : 0u) << u;
    }
  };
  static_assert(NU <= 32, "stem conv: element mask");

  int t = blockIdx.x;
  if (t < p.ntiles) prefetch(t);
  for (; t < p.ntiles; t += gridDim.x) {
    int b = t;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    __syncthreads();                                        // every wave is done with the previous tile's patch
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (tid + 256 * u < E) *reinterpret_cast<unsigned short*>(patch + (meta[u] & 0xffffu)) = to_bf16(((okm >> u) & 1u) ? v[u] : 0.f);
    __syncthreads();
    if (t + (int)gridDim.x < p.ntiles) prefetch(t + gridDim.x);      // in flight behind the MFMAs and the stores below

    f32x16 acc[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKS; ++kk) {
      int off[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * kk + h;
        const int tp = (j < NCH) ? j / CPT : 0, c8 = (j < NCH) ? j % CPT : 0;
        off[h] = ((tp / 3) * ROWP + tp % 3) * PS + c8 * 16;
      }
      const int ao = a_base + (lhi ? off[1] : off[0]);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(wl + b_base + kk * 2048);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wl + b_base + kk * 2048 + 512);
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(patch + ao + i * (2 * ROWP * PS));
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[i][1], 0, 0, 0);
      }
    }
    // epilogue as in the kernel above, through this wave's own transposition tile
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float bj = j ? bias1 : bias0;
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 68 + j * 32 + l31] = acc[i][j][r] + bj;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // 8 channels per lane (round 4): 16-byte stores — half the store instructions of the 4-channel form
      float4 o4[4][2];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float* tr = &T[(it * 8 + (lane >> 3)) * 68 + (lane & 7) * 8];
        o4[it][0] = *reinterpret_cast<const float4*>(tr);
        o4[it][1] = *reinterpret_cast<const float4*>(tr + 4);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = it * 8 + (lane >> 3);
        const int oy = oy0 + 2 * (TMW * wave + i) + (m >> 4), ox = ox0 + (m & 15);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const long o = (((long)n * p.Ho + oy) * p.Wo + ox) * 64 + (lane & 7) * 8;
        if (p.out != nullptr) {
          *reinterpret_cast<float4*>(p.out + o) = o4[it][0];
          *reinterpret_cast<float4*>(p.out + o + 4) = o4[it][1];
        }
#pragma unroll
        for (int ob = 0; ob < 3; ++ob) {
          unsigned short* const optr = ob == 0 ? p.out_bf16 : (ob == 1 ? p.obf2 : p.obf3);
          const float oslope = ob == 0 ? p.slope : (ob == 1 ? p.slope2 : p.slope3);
          if (optr == nullptr) continue;
          const float4 a = o4[it][0], b = o4[it][1];
          *reinterpret_cast<uint4*>(optr + o) = make_uint4(pack_bf16(apply_act_s(a.x, oslope), apply_act_s(a.y, oslope)),
                                                           pack_bf16(apply_act_s(a.z, oslope), apply_act_s(a.w, oslope)),
                                                           pack_bf16(apply_act_s(b.x, oslope), apply_act_s(b.y, oslope)),
                                                           pack_bf16(apply_act_s(b.z, oslope), apply_act_s(b.w, oslope)));
        }
      }
    }
  }
}

// W packed fp32 [tap][co 64][Cin]  ->  bf16 [group][stage][chunk][co][8]; chunk j of a stage = tap j / CPT, channels
// group*CG + (j % CPT)*8 .. +7; zero where the channel / chunk does not exist.
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* W, int K, int Cin, int CG, int ngroups, unsigned short* Wp) {
  const int CPT = CG / 8, TPS = (K == 3) ? 9 : 4, NST = (K == 3) ? 1 : 4;
  const int NCH = TPS * CPT, NCHP = (NCH + 1) / 2 * 2;
  const long total = (long)ngroups * NST * NCHP * 64 * 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int e = (int)(i & 7);
  const int co = (int)((i >> 3) & 63);
  long q = i >> 9;
  const int j = (int)(q % NCHP); q /= NCHP;
  const int st = (int)(q % NST);
  const int g = (int)(q / NST);
  float v = 0.f;
  if (j < NCH) {
    const int tp = j / CPT, c = g * CG + (j % CPT) * 8 + e;
    const int tap = (K == 3) ? tp : st * 4 + tp;
    if (c < Cin) v = W[((long)tap * 64 + co) * Cin + c];
  }
  Wp[i] = to_bf16(v);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient.  NW waves; wave w: output-channel half mt = w & 1, N-tiles (w >> 1) + (NW/2) * i of the T*CP columns
// n = tap * CP + ci (CP = channels per tap padded to a multiple of 4: a transposing read moves 4-channel blocks).
template <int K, int S, int TH, int CP, int NW, int NTW>
__global__ __launch_bounds__(NW * 64) void stem_wgrad_bf16_kernel(const StemK p) {
  constexpr int TW = 16, T = K * K, NT = NW * 64;
  constexpr int PH = (TH - 1) * S + K, PW = (TW - 1) * S + K, PWH = (PW + 1) / 2;
  constexpr int ROWP = (S == 1) ? PW : 2 * PWH;
  constexpr int PS = CP * 2;
  constexpr int DY_B = TH * TW * 128;                       // [pixel][64] bf16, 16-byte slots XOR-swizzled by pixel bit 1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dyt = smem;
  char* const patch = smem + DY_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int gq = lane >> 4, rr = (lane >> 2) & 3, c4 = lane & 3;      // transposing-read roles inside the 16-lane group
  const int mt = wave & 1;

  // A (dY^T): this lane supplies pixel (8 * khalf + rr) (+ 4 for the second read), channels co0 .. co0 + 3
  const int co0 = mt * 32 + 16 * (gq & 1) + 4 * c4;
  const int a_base = (8 * (gq >> 1) + rr) * 128 + (((co0 >> 3) ^ (((rr >> 1) & 1) << 2)) << 4) + (co0 & 7) * 2;
  // B (patch^T): columns n0 .. n0 + 3 of each of the wave's N-tiles
  int b_base[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    int n0 = ((wave >> 1) + (NW / 2) * i) * 32 + 16 * (gq & 1) + 4 * c4;
    if (n0 >= T * CP) n0 = 0;                               // columns beyond the filter: computed, never stored
    const int tap = n0 / CP, ci0 = n0 - tap * CP;
    const int r = tap / K, s = tap % K;
    const int ci = (S == 1) ? s : (s & 1) * PWH + (s >> 1);
    b_base[i] = (r * ROWP + ci + 8 * (gq >> 1) + rr) * PS + ci0 * 2;
  }
  f32x16 acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // (round 4) the bias gradient rides along: a spare channel slot of the patch (Ctot < CP) holds 1.0 at EVERY patch pixel — the
  // staging below never writes slots >= Ctot — so column (tap, Ctot) of dW is sum over pixels of dY for any tap; the reduce
  // kernel adds ONE tap's column (one whose input pixel exists for every output pixel: the centre for k3 p1, tap 0 for p0) to db
  if (p.ones_slot)
    for (int e = tid; e < PH * ROWP; e += NT) *reinterpret_cast<unsigned short*>(patch + e * PS + p.Ctot * 2) = 0x3f80;
  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    int b = t;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    __syncthreads();
    // ---- gradient tile: fp32 NHWC -> bf16, one item = 8 channels of a pixel
    {
      constexpr int NI = (TH * TW * 8 + NT - 1) / NT;
      if (p.dy_bf16) {            // bf16 STORAGE: the tile is copied as it is (16 bytes = 8 channels per item)
        uint4 v[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
          const int e = tid + NT * u;
          const int px = e >> 3, sl = e & 7;
          const int oy = oy0 + px / TW, ox = ox0 + px % TW;
          const bool ok = (e < TH * TW * 8) & (oy < p.Ho) & (ox < p.Wo);
          v[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.dY) +
                                                 (ok ? (((long)n * p.Ho + oy) * p.Wo + ox) * 64 + sl * 8 : 0));
          if (!ok) v[u] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
          const int e = tid + NT * u;
          const int px = e >> 3, sl = e & 7;
          if (e < TH * TW * 8) *reinterpret_cast<uint4*>(dyt + px * 128 + ((sl ^ (((px >> 1) & 1) << 2)) << 4)) = v[u];
        }
      } else {
      float4 v[NI][2];
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        const int e = tid + NT * u;
        const int px = e >> 3, sl = e & 7;
        const int oy = oy0 + px / TW, ox = ox0 + px % TW;
        const bool ok = (e < TH * TW * 8) & (oy < p.Ho) & (ox < p.Wo);
        const float4* src = reinterpret_cast<const float4*>(p.dY + (ok ? (((long)n * p.Ho + oy) * p.Wo + ox) * 64 + sl * 8 : 0));
        v[u][0] = src[0]; v[u][1] = src[1];
        if (!ok) { v[u][0] = make_float4(0.f, 0.f, 0.f, 0.f); v[u][1] = v[u][0]; }
      }
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        const int e = tid + NT * u;
        const int px = e >> 3, sl = e & 7;
        uint4 w;
        w.x = pack_bf16(v[u][0].x, v[u][0].y); w.y = pack_bf16(v[u][0].z, v[u][0].w);
        w.z = pack_bf16(v[u][1].x, v[u][1].y); w.w = pack_bf16(v[u][1].z, v[u][1].w);
        if (e < TH * TW * 8) *reinterpret_cast<uint4*>(dyt + px * 128 + ((sl ^ (((px >> 1) & 1) << 2)) << 4)) = w;
      }
      }
    }
    // ---- input patch, channel-last bf16
    stem_stage_patch<S, PH, PW, ROWP, PWH, PS, NT>(p, patch, n, iy0, ix0, 0, CP, tid);
    __syncthreads();
    // ---- MFMA: one k-step = the 16 pixels of a tile row
#pragma unroll
    for (int y = 0; y < TH; ++y) {
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dyt + a_base + (y * 16) * 128));
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dyt + a_base + (y * 16 + 4) * 128));
      const s16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(patch + b_base[i] + (y * S * ROWP) * PS));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(patch + b_base[i] + (y * S * ROWP + 4) * PS));
        const s16x8 bb = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), acc[i], 0, 0, 0);
      }
    }
  }
  // ---- partial result [block][co][n]: plain coalesced stores; stem_wgrad_reduce_kernel adds the blocks up
  float* o = p.part + (long)blockIdx.x * 64 * p.npad + (long)(mt * 32 + 4 * lhi) * p.npad + l31;
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int ntile = (wave >> 1) + (NW / 2) * i;
    if (ntile * 32 < p.npad) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * p.npad + ntile * 32] = acc[i][r];
    }
  }
}

// dW[tap][co][ci] += sum over workgroups of part[b][co][n = tap*CP + ci]; blockIdx.y strides over the workgroups
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* part, int nblocks, int npad, int T, int CP, int Ctot,
                                                                float* dW, float* db, int tap_b) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 64 * npad) return;
  const int co = e / npad, n = e - co * npad;
  const int tap = n / CP, ci = n - tap * CP;
  const bool bias_col = db != nullptr && tap == tap_b && ci == Ctot;       // the constant-one channel's column of one tap
  if (tap >= T || (ci >= Ctot && !bias_col)) return;
  const long stride = (long)64 * npad;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int b = blockIdx.y;
  const int step = gridDim.y;
  for (; b + 3 * step < nblocks; b += 4 * step) {
    t0 += part[b * stride + e];
    t1 += part[(b + step) * stride + e];
    t2 += part[(b + 2 * step) * stride + e];
    t3 += part[(b + 3 * step) * stride + e];
  }
  for (; b < nblocks; b += step) t0 += part[b * stride + e];
  if (bias_col) atomicAdd(db + co, (t0 + t1) + (t2 + t3));
  else atomicAdd(dW + ((long)tap * 64 + co) * Ctot + ci, (t0 + t1) + (t2 + t3));
}

static int stem_fill(StemK& k, const pg_src_t* src, int nsrc, int N, int Hi, int Wi, int K, int stride, int pad) {
  memset(&k, 0, sizeof(k));
  int c = 0;
  for (int j = 0; j < nsrc; ++j) {
    if (src[j].aff || src[j].mask) return -1;
    k.src[j] = src[j]; k.cstart[j] = c; c += src[j].C;
  }
  for (int j = nsrc; j <= PG_MAX_SRC; ++j) k.cstart[j] = c;
  k.nsrc = nsrc; k.Ctot = c;
  k.N = N; k.Hi = Hi; k.Wi = Wi; k.pad = pad;
  k.Ho = (Hi + 2 * pad - K) / stride + 1; k.Wo = (Wi + 2 * pad - K) / stride + 1;
  return 0;
}

static int stem_cu_count() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int K, int S, int CG, int TH>
static int launch_stem_conv(StemK& k, hipStream_t st) {
  constexpr int PH = (TH - 1) * S + K, PW = 15 * S + K, PWH = (PW + 1) / 2, ROWP = (S == 1) ? PW : 2 * PWH;
  constexpr int PATCH_B = (PH * ROWP * CG * 2 + 15) / 16 * 16;
  constexpr int NCH = ((K == 3) ? 9 : 4) * (CG / 8), NKS = (NCH + 1) / 2;
  constexpr int LDS = PATCH_B + (2 * NKS * 1024 > 4 * 32 * 68 * 4 ? 2 * NKS * 1024 : 4 * 32 * 68 * 4);      // filter stage / the epilogue's four transposition tiles
  static_assert(LDS <= 160 * 1024, "stem conv: LDS");
  k.tiles_x = (k.Wo + 15) / 16; k.tiles_y = (k.Ho + TH - 1) / TH;
  k.ntiles = k.tiles_x * k.tiles_y * k.N;
  auto kern = stem_conv_bf16_kernel<K, S, CG, TH>;
  static bool set = false;
  if (!set) {
    PG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    set = true;
  }
  int per_cu = (160 * 1024) / LDS;
  if (per_cu > 4) per_cu = 4;
  int blocks = stem_cu_count() * per_cu;
  if (blocks > k.ntiles) blocks = k.ntiles;
  PG_KLAUNCH(kern, dim3((unsigned)blocks), dim3(256), LDS, st, k);
  return 0;
}

// pipelined forward kernel (one source, one channel group, k3 s1, CG = 24): 2 workgroups per CU
template <int CG, int TH>
static int launch_stem_conv_pf(StemK& k, hipStream_t st) {
  constexpr int PH = TH + 2, PW = 18;
  constexpr int PATCH_B = (PH * PW * CG * 2 + 15) / 16 * 16;
  constexpr int NCH = 9 * (CG / 8), NKS = (NCH + 1) / 2;
  constexpr int LDS = PATCH_B + 2 * NKS * 1024 + 4 * 32 * 68 * 4;
  static_assert(2 * LDS <= 160 * 1024, "stem conv (pipelined): two workgroups per CU");
  k.tiles_x = (k.Wo + 15) / 16; k.tiles_y = (k.Ho + TH - 1) / TH;
  k.ntiles = k.tiles_x * k.tiles_y * k.N;
  auto kern = stem_conv_bf16_pf_kernel<CG, TH>;
  static bool set = false;
  if (!set) {
    PG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    set = true;
  }
  int blocks = stem_cu_count() * 2;
  if (blocks > k.ntiles) blocks = k.ntiles;
  PG_KLAUNCH(kern, dim3((unsigned)blocks), dim3(256), LDS, st, k);
  return 0;
}

// return code of launch_stem_wgrad: 0 = launched, STEM_RC_BIAS_FUSED = launched AND the bias gradient rode along, anything else
// (PG_FAIL / PG_REQUIRE / PG_HIP codes 1..3, -2 = workspace too small) = error
constexpr int STEM_RC_BIAS_FUSED = 100;
template <int K, int S, int TH, int CP, int NW, int NTW>
static int launch_stem_wgrad(StemK& k, float* dW, float* ws, long ws_floats, hipStream_t st, float* db = nullptr) {
  constexpr int PH = (TH - 1) * S + K, PW = 15 * S + K, PWH = (PW + 1) / 2, ROWP = (S == 1) ? PW : 2 * PWH;
  constexpr int LDS = TH * 16 * 128 + (PH * ROWP * CP * 2 + 15) / 16 * 16;
  static_assert(LDS <= 160 * 1024, "stem wgrad: LDS");
  static_assert(NTW * (NW / 2) * 32 >= K * K * CP, "stem wgrad: N-tiles");
  k.tiles_x = (k.Wo + 15) / 16; k.tiles_y = (k.Ho + TH - 1) / TH;
  k.ntiles = k.tiles_x * k.tiles_y * k.N;
  k.npad = (K * K * CP + 31) / 32 * 32;
  auto kern = stem_wgrad_bf16_kernel<K, S, TH, CP, NW, NTW>;
  static bool set = false;
  if (!set) {
    PG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    set = true;
  }
  int per_cu = (160 * 1024) / LDS;
  const int reg_cap = (NTW * 16 + 64 <= 128) ? 4 / (NW / 4) : (NTW * 16 + 64 <= 256 ? 2 / (NW / 4) : 1);
  if (per_cu > reg_cap) per_cu = reg_cap;
  if (per_cu < 1) per_cu = 1;
  int blocks = stem_cu_count() * per_cu;
  if (blocks > k.ntiles) blocks = k.ntiles;
  const long need = (long)blocks * 64 * k.npad;
  if (ws == nullptr || ws_floats < need) {
    blocks = (int)(ws_floats / ((long)64 * k.npad));
    if (ws == nullptr || blocks < 1) return -2;
  }
  k.part = ws;
  // bias gradient through the constant-one channel: needs a spare slot and a tap whose input pixel exists for every output pixel
  const bool fuse_b = db != nullptr && k.Ctot < CP && ((K == 3 && k.pad == 1) || k.pad == 0);
  k.ones_slot = fuse_b ? 1 : 0;
  PG_KLAUNCH(kern, dim3((unsigned)blocks), dim3(NW * 64), LDS, st, k);
  const int ry = (blocks >= 32 && !deterministic()) ? 16 : 1;      // PG_DETERMINISTIC: one serial walk per element, no float atomics in arrival order
  PG_KLAUNCH(stem_wgrad_reduce_kernel, dim3((unsigned)((64 * k.npad + 255) / 256), (unsigned)ry), dim3(256), 0, st, k.part,
                     blocks, k.npad, K * K, CP, k.Ctot, dW, fuse_b ? db : (float*)nullptr, (K == 3 && k.pad == 1) ? 4 : 0);
  return fuse_b ? STEM_RC_BIAS_FUSED : 0;
}

}  // namespace pg

static int pg_stem_group_channels(int32_t Cin) { return (Cin <= 24 || (Cin > 40 && Cin <= 48)) ? 24 : 40; }

extern "C" int64_t pg_stem_pack_elems(int32_t K, int32_t Cin) {
  const int CG = pg_stem_group_channels(Cin), ng = (Cin + CG - 1) / CG;
  const int NCH = ((K == 3) ? 9 : 4) * (CG / 8), NCHP = (NCH + 1) / 2 * 2;
  return (int64_t)ng * ((K == 3) ? 1 : 4) * NCHP * 512;
}

extern "C" int pg_stem_pack_bf16(const float* W, int32_t K, int32_t Cin, uint16_t* Wp, void* stream) {
  PG_REQUIRE(W && Wp && (K == 3 || K == 4) && Cin > 0 && Cin <= 80, "pg_stem_pack_bf16: bad arguments (K=%d Cin=%d)", K, Cin);
  const int CG = pg_stem_group_channels(Cin), ng = (Cin + CG - 1) / CG;
  const long total = pg_stem_pack_elems(K, Cin);
  PG_KLAUNCH(pg::stem_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, K, Cin, CG,
                     ng, Wp);
  PG_LAUNCH_OK("pg_stem_pack_bf16");
  return 0;
}

extern "C" int pg_stem_conv_bf16_ex(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                                    int32_t pad, const uint16_t* Wp, const float* bias, float* out, uint16_t* out_bf16, int32_t act,
                                    void* stream);
extern "C" int pg_stem_conv_bf16_v3(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                                    int32_t pad, const uint16_t* Wp, const float* bias, float* out, uint16_t* out_bf16, int32_t act,
                                    uint16_t* out2_bf16, int32_t act2, uint16_t* out3_bf16, int32_t act3, void* stream);

extern "C" int pg_stem_conv_bf16(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                                 int32_t pad, const uint16_t* Wp, const float* bias, float* out, void* stream) {
  return pg_stem_conv_bf16_ex(src, nsrc, N, Hi, Wi, K, stride, pad, Wp, bias, out, nullptr, PG_ACT_NONE, stream);
}

extern "C" int pg_stem_conv_bf16_ex(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                                    int32_t pad, const uint16_t* Wp, const float* bias, float* out, uint16_t* out_bf16, int32_t act,
                                    void* stream) {
  PG_REQUIRE(out != nullptr, "pg_stem_conv_bf16: out null");
  return pg_stem_conv_bf16_v3(src, nsrc, N, Hi, Wi, K, stride, pad, Wp, bias, out, out_bf16, act, nullptr, PG_ACT_NONE, nullptr,
                              PG_ACT_NONE, stream);
}

// Round 3 (bf16 STORAGE): `out` (fp32) may be NULL; up to three bf16 outputs bf16(act_k(conv + bias)), act NONE = the raw tensor
extern "C" int pg_stem_conv_bf16_v3(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                                    int32_t pad, const uint16_t* Wp, const float* bias, float* out, uint16_t* out_bf16, int32_t act,
                                    uint16_t* out2_bf16, int32_t act2, uint16_t* out3_bf16, int32_t act3, void* stream) {
  PG_REQUIRE(src && nsrc >= 1 && nsrc <= PG_MAX_SRC && Wp && (out || out_bf16 || out2_bf16 || out3_bf16), "pg_stem_conv_bf16: bad arguments");
  PG_REQUIRE((K == 3 && stride == 1) || (K == 4 && stride == 2), "pg_stem_conv_bf16: only k3s1 / k4s2 (got k%d s%d)", K, stride);
  pg::StemK k;
  PG_REQUIRE(pg::stem_fill(k, src, nsrc, N, Hi, Wi, K, stride, pad) == 0, "pg_stem_conv_bf16: sources must not carry aff / mask");
  PG_REQUIRE(k.Ctot <= 80 && k.Ho > 0 && k.Wo > 0 && N > 0, "pg_stem_conv_bf16: Cin <= 80 and a non-empty output required");
  k.Wp = Wp; k.bias = bias; k.out = out;
  k.out_bf16 = out_bf16;
  k.slope = act == PG_ACT_RELU ? 0.f : (act == PG_ACT_LEAKY ? 0.2f : 1.f);
  k.obf2 = out2_bf16; k.slope2 = act2 == PG_ACT_RELU ? 0.f : (act2 == PG_ACT_LEAKY ? 0.2f : 1.f);
  k.obf3 = out3_bf16; k.slope3 = act3 == PG_ACT_RELU ? 0.f : (act3 == PG_ACT_LEAKY ? 0.2f : 1.f);
  const int CG = pg_stem_group_channels(k.Ctot);
  k.ngroups = (k.Ctot + CG - 1) / CG;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  static const bool no_pf = getenv("PG_NO_STEM_PF") != nullptr;        // ablation switch: the round-3 kernel everywhere
  if (K == 3 && CG == 24 && nsrc == 1 && k.ngroups == 1 && !no_pf &&
      (double)Hi * Wi * k.Ctot * 4.0 < 2147483648.0) rc = pg::launch_stem_conv_pf<24, 16>(k, st);
  else if (K == 3) rc = (CG == 24) ? pg::launch_stem_conv<3, 1, 24, 16>(k, st) : pg::launch_stem_conv<3, 1, 40, 16>(k, st);
  else rc = (CG == 24) ? pg::launch_stem_conv<4, 2, 24, 16>(k, st) : pg::launch_stem_conv<4, 2, 40, 8>(k, st);
  if (rc) return rc;
  PG_LAUNCH_OK("pg_stem_conv_bf16");
  return 0;
}

extern "C" int pg_stem_wgrad_bf16_v2(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K,
                                     int32_t stride, int32_t pad, const void* dY, int32_t dy_is_bf16, float* dW, float* dbias,
                                     float* workspace, int64_t workspace_floats, void* stream);
extern "C" int pg_stem_wgrad_bf16_ex(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K,
                                     int32_t stride, int32_t pad, const void* dY, int32_t dy_is_bf16, float* dW, float* workspace,
                                     int64_t workspace_floats, void* stream) {
  return pg_stem_wgrad_bf16_v2(src, nsrc, N, Hi, Wi, K, stride, pad, dY, dy_is_bf16, dW, nullptr, workspace, workspace_floats, stream);
}
extern "C" int pg_stem_wgrad_bf16(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K,
                                  int32_t stride, int32_t pad, const float* dY, float* dW, float* workspace,
                                  int64_t workspace_floats, void* stream) {
  return pg_stem_wgrad_bf16_ex(src, nsrc, N, Hi, Wi, K, stride, pad, dY, 0, dW, workspace, workspace_floats, stream);
}
// dy_is_bf16: the gradient of the first layer's output is stored as bf16 (bf16 STORAGE, round 3)
// dbias (round 4, optional): db[co] += sum over pixels of dY — the layer's bias gradient from the same pass (a constant-one input
// channel); done when pg_last_launch_info() has PG_INFO_STEM_BIAS set, otherwise the caller runs pg_bias_grad*
extern "C" int pg_stem_wgrad_bf16_v2(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K,
                                     int32_t stride, int32_t pad, const void* dY, int32_t dy_is_bf16, float* dW, float* dbias,
                                     float* workspace, int64_t workspace_floats, void* stream) {
  PG_REQUIRE(src && nsrc >= 1 && nsrc <= PG_MAX_SRC && dY && dW, "pg_stem_wgrad_bf16: bad arguments");
  PG_REQUIRE((K == 3 && stride == 1) || (K == 4 && stride == 2), "pg_stem_wgrad_bf16: only k3s1 / k4s2 (got k%d s%d)", K, stride);
  pg::StemK k;
  PG_REQUIRE(pg::stem_fill(k, src, nsrc, N, Hi, Wi, K, stride, pad) == 0, "pg_stem_wgrad_bf16: sources must not carry aff / mask");
  PG_REQUIRE(k.Ho > 0 && k.Wo > 0 && N > 0, "pg_stem_wgrad_bf16: empty output");
  k.dY = reinterpret_cast<const float*>(dY);
  k.dy_bf16 = dy_is_bf16 ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const int c = k.Ctot;
  int rc;
  if (K == 3) {
    PG_REQUIRE(c <= 36, "pg_stem_wgrad_bf16: k3 supports Cin <= 36 (got %d)", c);
    rc = (c <= 24) ? pg::launch_stem_wgrad<3, 1, 8, 24, 4, 4>(k, dW, workspace, workspace_floats, st, dbias)
                   : pg::launch_stem_wgrad<3, 1, 8, 36, 4, 6>(k, dW, workspace, workspace_floats, st, dbias);
  } else {
    PG_REQUIRE(c <= 72, "pg_stem_wgrad_bf16: k4 supports Cin <= 72 (got %d)", c);
    rc = (c <= 44) ? pg::launch_stem_wgrad<4, 2, 8, 44, 8, 6>(k, dW, workspace, workspace_floats, st, dbias)
                   : pg::launch_stem_wgrad<4, 2, 8, 72, 8, 9>(k, dW, workspace, workspace_floats, st, dbias);
  }
  PG_REQUIRE(rc != -2, "pg_stem_wgrad_bf16: a workspace of at least 64 x %d floats is required", k.npad);
  if (rc != 0 && rc != pg::STEM_RC_BIAS_FUSED) return rc;        // (error message already set by the failing macro)
  pg::last_info() = 7 | (1 << 4) | (1 << 16) | (1 << 30) | (rc == pg::STEM_RC_BIAS_FUSED ? PG_INFO_STEM_BIAS : 0);     // tile code 7 = bf16 stem kernel, scalar X
  PG_LAUNCH_OK("pg_stem_wgrad_bf16");
  return 0;
}
