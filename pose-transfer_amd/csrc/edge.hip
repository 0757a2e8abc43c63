// Edge-layer kernels (gfx950): the 256->3 output convolution of the generator (reference models/networks.py:228:
// ReLU -> Conv2d(k3,p1,bias) -> Tanh) has GEMM-N = 3, which wastes 29/32 of an MFMA tile.  It is re-associated as
//   forward : Y[p][(tap,co)] = W[tap][co][:] . x[p][:]   (a 1x1 convolution with N = 27 -> pg_conv, 84 % tile use)
//             out[q][co]     = tanh(b[co] + sum_tap Y[q+off(tap)][(tap,co)])          (pg_tap_gather, HBM-bound, 28 MB)
//   wgrad   : G[p][(tap,co)] = dY[p-off(tap)][co]                                     (pg_im2col_taps, 28 MB)
//             dW[(tap,co)][ci] = sum_p G[p][(tap,co)] * x[p][ci]                      (pg_conv_wgrad as a 1x1, M = 27)
//   dgrad   : dX[p][ci] = sum_{tap,co} dY[p-off(tap)][co] * W[tap][co][ci]            (pg_small_cout_dgrad: K = 27 only,
//             so this is an HBM-bound streaming kernel: read fwd + write dz = 8 B per element)
#include "common.h"

namespace pg {

// out[n,co,y,x] = act(bias[co] + sum_{r,s} Y[n, y+r-pad, x+s-pad, (r*KW+s)*Co + co]);  Y NHWC with T*Co channels
__global__ __launch_bounds__(256) void tap_gather_kernel(const float* Y, int N, int H, int W, int KH, int KW, int pad,
                                                         int Co, const float* bias, int out_act, float* out, long oN,
                                                         long oC, long oH, long oW) {
  const long total = (long)N * H * W;
  const int CT = KH * KW * Co;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long r0 = i / W;
    const int y = (int)(r0 % H);
    const int n = (int)(r0 / H);
    for (int co = 0; co < Co; ++co) {
      float acc = bias ? bias[co] : 0.f;
      for (int r = 0; r < KH; ++r) {
        const int yy = y + r - pad;
        if (yy < 0 || yy >= H) continue;
        for (int s = 0; s < KW; ++s) {
          const int xx = x + s - pad;
          if (xx < 0 || xx >= W) continue;
          acc += Y[(((long)n * H + yy) * W + xx) * CT + (r * KW + s) * Co + co];
        }
      }
      if (out_act == PG_OUT_TANH) acc = tanhf(acc);
      out[(long)n * oN + (long)co * oC + (long)y * oH + (long)x * oW] = acc;
    }
  }
}

// The 3x3 / 3-channel instance (the generator's output convolution) through LDS: a workgroup owns an 8 x 32 pixel tile,
// stages the 10 x 34 halo of tap rows (27 floats per pixel, contiguous per image row: coalesced 4-byte loads of 918-float
// runs) and gathers from LDS (pixel stride 27 floats: conflict-free).  The direct version above issues 27 loads per lane
// with a 108-byte lane stride (0.36 TB/s at batch 32).
__global__ __launch_bounds__(256) void tap_gather_333_kernel(const float* Y, int N, int H, int W, const float* bias, int out_act,
                                                             float* out, long oN, long oC, long oH, long oW, int pitch) {
  // pitch: floats per pixel row of Y (27 = dense; 64 when the 1x1 contraction ran on the bf16 512 x 64 kernel, round 3)
  constexpr int TH = 8, TW = 32, CT = 27;
  __shared__ float tile[(TH + 2) * (TW + 2) * CT];
  const int n = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int xlo = max(x0 - 1, 0), xhi = min(x0 + TW + 1, W);          // staged columns [xlo, xhi)
  const int run = (xhi - xlo) * CT, lpad = (xlo - (x0 - 1)) * CT;
  if (pitch >= 28 && (pitch & 3) == 0 && (((size_t)Y) & 15) == 0) {
    // padded pixel rows (round 3: pitch 32 / 64 behind the bf16 contraction): one item = a pixel of the halo, its 27 floats arrive
    // as seven 16-byte loads (the 4-byte form below pays an integer division per float and 27 load instructions per pixel)
    for (int i = threadIdx.x; i < (TH + 2) * (TW + 2); i += 256) {
      const int r = i / (TW + 2), c = i - r * (TW + 2);
      const int yy = y0 + r - 1, xx = x0 + c - 1;
      float* tp = tile + i * CT;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) {
#pragma unroll
        for (int k = 0; k < CT; ++k) tp[k] = 0.f;
      } else {
        const float4* src = reinterpret_cast<const float4*>(Y + (((long)n * H + yy) * W + xx) * pitch);
        float4 v[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) v[q] = src[q];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          tp[4 * q] = v[q].x; tp[4 * q + 1] = v[q].y; tp[4 * q + 2] = v[q].z;
          if (q < 6) tp[4 * q + 3] = v[q].w;
        }
      }
    }
  } else
  for (int r = 0; r < TH + 2; ++r) {
    const int yy = y0 + r - 1;
    float* trow = tile + r * (TW + 2) * CT;
    if (yy < 0 || yy >= H) {
      for (int i = threadIdx.x; i < (TW + 2) * CT; i += 256) trow[i] = 0.f;
    } else {
      const float* src = Y + (((long)n * H + yy) * W + xlo) * pitch;
      for (int i = threadIdx.x; i < (TW + 2) * CT; i += 256) {
        const int k = i - lpad;
        const int kp = (pitch == CT) ? k : (k / CT) * pitch + k % CT;
        trow[i] = (k >= 0 && k < run) ? src[kp] : 0.f;
      }
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x = x0 + tx, y = y0 + ty;
  if (x >= W || y >= H) return;
  float acc[3] = {bias ? bias[0] : 0.f, bias ? bias[1] : 0.f, bias ? bias[2] : 0.f};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float* q = tile + ((ty + r) * (TW + 2) + tx + s) * CT + (r * 3 + s) * 3;
      acc[0] += q[0]; acc[1] += q[1]; acc[2] += q[2];
    }
#pragma unroll
  for (int co = 0; co < 3; ++co) {
    float v = acc[co];
    if (out_act == PG_OUT_TANH) v = tanhf(v);
    out[(long)n * oN + (long)co * oC + (long)y * oH + (long)x * oW] = v;
  }
}

// G[n,y,x,(r*KW+s)*C + c] = dY[n,c,y-(r-pad),x-(s-pad)] (0 outside), channels [T*C, Cpad) zero-filled
// OB: G is a bf16 tensor (round 3: the gradient operand of the output convolution's bf16 contractions, Cpad = 64)
template <bool OB>
__global__ __launch_bounds__(256) void im2col_taps_kernel(const float* dY, long yN, long yC, long yH, long yW, int N,
                                                          int H, int W, int KH, int KW, int pad, int C, int Cpad,
                                                          float* G) {
  const long total = (long)N * H * W * Cpad;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % Cpad);
    long r0 = i / Cpad;
    const int x = (int)(r0 % W); r0 /= W;
    const int y = (int)(r0 % H);
    const int n = (int)(r0 / H);
    float v = 0.f;
    if (k < KH * KW * C) {
      const int tap = k / C, c = k - tap * C;
      const int r = tap / KW, s = tap - r * KW;
      const int yy = y - (r - pad), xx = x - (s - pad);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = dY[(long)n * yN + (long)c * yC + (long)yy * yH + (long)xx * yW];
    }
    if constexpr (OB) {
      unsigned r_;
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r_) : "v"(v), "v"(0.f));
      reinterpret_cast<unsigned short*>(G)[i] = (unsigned short)(r_ & 0xffffu);
    } else {
      G[i] = v;
    }
  }
}

struct SmallDgradK {
  const float* dY; long yN, yC, yH, yW;
  int N, H, W, KH, KW, pad, Co, Ctot;
  int stride, Ho, Wo;              // dY is (N,Co,Ho,Wo); input pixel (y,x) pairs with output ((y+pad-r)/stride, (x+pad-s)/stride)
  const float* Wt;                 // packed [KH][KW][Co][Ctot]
  pg_dst_t dst[PG_MAX_SRC];
  int ndst;
  int dstart[PG_MAX_SRC + 1];
};

__global__ __launch_bounds__(256) void small_cout_dgrad_kernel(const SmallDgradK p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];     // KH*KW*Co*Ctot floats
  const int wn = p.KH * p.KW * p.Co * p.Ctot;
  for (int i = threadIdx.x * 4; i < wn; i += 256 * 4)
    *reinterpret_cast<float4*>(&wl[i]) = *reinterpret_cast<const float4*>(&p.Wt[i]);
  __syncthreads();
  const int cpp = p.Ctot >> 2;     // float4 chunks per pixel
  // One wave-iteration covers 64 consecutive float4 chunks.  When a pixel has >= 64 chunks per wave-iteration the
  // pixel index is wave-uniform (made provably so with readfirstlane) and the dY taps become scalar loads.
  const long items = (long)p.N * p.H * p.W * cpp;
  const bool uni = (cpp % 64) == 0;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const int cg = (int)(it % cpp) * 4;
    long pix = it / cpp;
    if (uni) {
      const int lo = __builtin_amdgcn_readfirstlane((int)(pix & 0xffffffff));
      const int hi = __builtin_amdgcn_readfirstlane((int)(pix >> 32));
      pix = ((long)hi << 32) | (unsigned)lo;
    }
    const int x = (int)(pix % p.W);
    const long r0 = pix / p.W;
    const int y = (int)(r0 % p.H);
    const int n = (int)(r0 / p.H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < p.KH; ++r) {
      const int ty = y + p.pad - r;
      if (ty < 0 || ty % p.stride != 0) continue;
      const int yy = ty / p.stride;
      if (yy >= p.Ho) continue;
      for (int s = 0; s < p.KW; ++s) {
        const int tx = x + p.pad - s;
        if (tx < 0 || tx % p.stride != 0) continue;
        const int xx = tx / p.stride;
        if (xx >= p.Wo) continue;
        for (int co = 0; co < p.Co; ++co) {
          const float g = p.dY[(long)n * p.yN + (long)co * p.yC + (long)yy * p.yH + (long)xx * p.yW];
          const float4 w = *reinterpret_cast<const float4*>(&wl[((r * p.KW + s) * p.Co + co) * p.Ctot + cg]);
          acc.x += g * w.x; acc.y += g * w.y; acc.z += g * w.z; acc.w += g * w.w;
        }
      }
    }
    int d = 0;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q) if (q < p.ndst && cg >= p.dstart[q]) d = q;
    const pg_dst_t& ds = p.dst[d];
    const int c = cg - p.dstart[d];
    const long idx = pix * ds.C + c;
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ds.mask) mk = *reinterpret_cast<const float4*>(ds.mask + (long)n * ds.C + c);
    float g4[4] = {acc.x, acc.y, acc.z, acc.w};
    const float m4[4] = {mk.x, mk.y, mk.z, mk.w};
    if (ds.fwd) {
      const float4 f = *reinterpret_cast<const float4*>(ds.fwd + idx);
      const float a = ds.aff ? ds.aff[2 * n] : 1.f, b = ds.aff ? ds.aff[2 * n + 1] : 0.f;
      const float f4[4] = {f.x, f.y, f.z, f.w};
      const float slope = act_slope(ds.act);
#pragma unroll
      for (int e = 0; e < 4; ++e) g4[e] *= act_grad_s((f4[e] * a + b) * m4[e], slope);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) g4[e] *= m4[e];
    float4* o = reinterpret_cast<float4*>(ds.grad + idx);
    if (ds.accumulate) { const float4 old = *o; g4[0] += old.x; g4[1] += old.y; g4[2] += old.z; g4[3] += old.w; }
    *o = make_float4(g4[0], g4[1], g4[2], g4[3]);
  }
}

}  // namespace pg

using namespace pg;

extern "C" int pg_tap_gather_pitch(const float* Y, int32_t pitch, int32_t N, int32_t H, int32_t W, const float* bias,
                                   int32_t out_act, float* out, int64_t oN, int64_t oC, int64_t oH, int64_t oW, void* stream) {
  // the k3 p1 Co = 3 gather over a tap tensor whose pixel rows are `pitch` >= 27 floats apart
  PG_REQUIRE(Y && out && N > 0 && N <= 65535 && pitch >= 27, "pg_tap_gather_pitch: bad arguments");
  PG_KLAUNCH(tap_gather_333_kernel, dim3((W + 31) / 32, (H + 7) / 8, N), dim3(256), 0, (hipStream_t)stream, Y, N, H, W,
                     bias, out_act, out, (long)oN, (long)oC, (long)oH, (long)oW, pitch);
  PG_LAUNCH_OK("pg_tap_gather_pitch");
  return 0;
}

extern "C" int pg_tap_gather(const float* Y, int32_t N, int32_t H, int32_t W, int32_t KH, int32_t KW, int32_t pad,
                             int32_t Co, const float* bias, int32_t out_act, float* out, int64_t oN, int64_t oC,
                             int64_t oH, int64_t oW, void* stream) {
  PG_REQUIRE(Y && out && N > 0 && Co > 0, "pg_tap_gather: bad arguments");
  if (KH == 3 && KW == 3 && pad == 1 && Co == 3 && N <= 65535) {
    PG_KLAUNCH(tap_gather_333_kernel, dim3((W + 31) / 32, (H + 7) / 8, N), dim3(256), 0, (hipStream_t)stream, Y, N, H, W,
                       bias, out_act, out, (long)oN, (long)oC, (long)oH, (long)oW, 27);
    PG_LAUNCH_OK("pg_tap_gather");
    return 0;
  }
  long blocks = ((long)N * H * W + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  PG_KLAUNCH(tap_gather_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, Y, N, H, W, KH, KW, pad, Co,
                     bias, out_act, out, (long)oN, (long)oC, (long)oH, (long)oW);
  PG_LAUNCH_OK("pg_tap_gather");
  return 0;
}

extern "C" int pg_im2col_taps(const float* dY, int64_t yN, int64_t yC, int64_t yH, int64_t yW, int32_t N, int32_t H,
                              int32_t W, int32_t KH, int32_t KW, int32_t pad, int32_t C, int32_t Cpad, float* G,
                              void* stream) {
  PG_REQUIRE(dY && G && N > 0 && Cpad >= KH * KW * C, "pg_im2col_taps: bad arguments");
  long blocks = ((long)N * H * W * Cpad + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  PG_KLAUNCH(im2col_taps_kernel<false>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dY, (long)yN, (long)yC,
                     (long)yH, (long)yW, N, H, W, KH, KW, pad, C, Cpad, G);
  PG_LAUNCH_OK("pg_im2col_taps");
  return 0;
}

// the same with a bf16 output tensor G [N][H][W][Cpad] (bf16 data path: operand of the output convolution's data- and
// weight-gradient contractions)
extern "C" int pg_im2col_taps_bf16(const float* dY, int64_t yN, int64_t yC, int64_t yH, int64_t yW, int32_t N, int32_t H,
                                   int32_t W, int32_t KH, int32_t KW, int32_t pad, int32_t C, int32_t Cpad, void* G_bf16,
                                   void* stream) {
  PG_REQUIRE(dY && G_bf16 && N > 0 && Cpad >= KH * KW * C, "pg_im2col_taps_bf16: bad arguments");
  long blocks = ((long)N * H * W * Cpad + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  PG_KLAUNCH(im2col_taps_kernel<true>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dY, (long)yN, (long)yC,
                     (long)yH, (long)yW, N, H, W, KH, KW, pad, C, Cpad, reinterpret_cast<float*>(G_bf16));
  PG_LAUNCH_OK("pg_im2col_taps_bf16");
  return 0;
}

extern "C" int pg_small_cout_dgrad(const float* dY, int64_t yN, int64_t yC, int64_t yH, int64_t yW, int32_t N, int32_t H,
                                   int32_t W, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t Co,
                                   const float* Wt, const pg_dst_t* dst, int32_t ndst, void* stream) {
  PG_REQUIRE(dY && Wt && dst && ndst >= 1 && ndst <= PG_MAX_SRC && stride >= 1, "pg_small_cout_dgrad: bad arguments");
  SmallDgradK k;
  memset(&k, 0, sizeof(k));
  k.dY = dY; k.yN = yN; k.yC = yC; k.yH = yH; k.yW = yW;
  k.N = N; k.H = H; k.W = W; k.KH = KH; k.KW = KW; k.pad = pad; k.Co = Co; k.Wt = Wt;
  k.stride = stride; k.Ho = (H + 2 * pad - KH) / stride + 1; k.Wo = (W + 2 * pad - KW) / stride + 1;
  int c = 0;
  for (int j = 0; j < ndst; ++j) {
    k.dst[j] = dst[j]; k.dstart[j] = c; c += dst[j].C;
    PG_REQUIRE(dst[j].C % 4 == 0, "pg_small_cout_dgrad: destination channels must be multiples of 4");
  }
  for (int j = ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
  k.ndst = ndst; k.Ctot = c;
  const size_t lds = sizeof(float) * (size_t)KH * KW * Co * c;
  PG_REQUIRE(lds <= 160 * 1024 && (KH * KW * Co * c) % 4 == 0, "pg_small_cout_dgrad: weight tile does not fit LDS");
  if (lds > 64 * 1024)
    PG_HIP(hipFuncSetAttribute((const void*)small_cout_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long blocks = ((long)N * H * W * (c / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;      // each block stages the weights once (27 KB): enough blocks to hide latency
  PG_KLAUNCH(small_cout_dgrad_kernel, dim3((int)blocks), dim3(256), lds, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_small_cout_dgrad");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// First-layer convolutions: few input channels (3+P / P / 3+2P+3) straight from NCHW tensors, 64 output channels.
//   encoder level 0 : nn.Conv2d(cin, 64, k3, p1, bias)      reference models/networks.py:186
//   discriminator   : nn.Conv2d(3+2P+3, 64, k4, s2, bias)   reference models/networks.py:341
// The generic implicit-GEMM gather spends its time on per-element address math for these (K = 9*21 = 189 is tiny and
// ragged).  Here a workgroup stages the INPUT PATCH of its 8x16 output-pixel tile in LDS once per 8-channel chunk and
// the MFMA A-operand is read directly out of the patch (no im2col tile, no bounds checks in the inner loop); the
// weight chunk comes from a [c][tap][co] repack (pg_repack_small_cin) as contiguous float4s.
namespace pg {

struct SmallCinK {
  pg_src_t src[PG_MAX_SRC];
  int nsrc, Ctot;
  int cstart[PG_MAX_SRC + 1];
  int N, Hi, Wi, Ho, Wo, pad;
  const float* Wt;      // repacked [Ctot][K*K][64]
  const float* bias;
  float* out;           // NHWC [N][Ho][Wo][64]
  int tiles_x, tiles_y;
};

template <int K, int S>
__global__ __launch_bounds__(256) void small_cin_conv_kernel(const SmallCinK p) {
  constexpr int TH = 8, TW = 16, CCH = 8, T = K * K;
  constexpr int PH = (TH - 1) * S + K;
  constexpr int PWU = (TW - 1) * S + K;                      // used columns
  constexpr int PWS = (S == 1) ? PWU : 20;                   // row stride (S=2: per-parity half row, padded 17 -> 20)
  constexpr int ROWF = (S == 1) ? PWS : 2 * PWS;             // floats per patch row
  constexpr int PATCH = CCH * PH * ROWF;
  constexpr int KC = CCH * T;                                // K-dim per chunk (72 / 128)
  __shared__ __attribute__((aligned(16))) float smem[PATCH + KC * 64];
  float* patch = smem;
  float* wl = smem + PATCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
  const int wy0 = (wave >> 1) * 4;                           // this wave: output rows wy0..wy0+3, channels wn0..wn0+31
  const int wn0 = (wave & 1) * 32;
  // per-lane pixel offsets inside the patch for the two 32-pixel MFMA row tiles (rows wy0+2i+(l31>>4), col l31&15)
  int pixoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int y = wy0 + 2 * i + (l31 >> 4), x = l31 & 15;
    pixoff[i] = y * S * ROWF + x;                            // S=2: x indexes the per-parity half row
  }
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  for (int c0 = 0; c0 < p.Ctot; c0 += CCH) {
    __syncthreads();
    // ---- stage the input patch of channels c0..c0+7 (zero outside the image / beyond Ctot) and the weight chunk
    //      [(cl*T + tap)][co] (contiguous in the repacked layout).  All loads of the chunk are issued branch-free
    //      (clamped address + select) before the first LDS store: one memory latency per chunk, not one per element.
    constexpr int NP = (CCH * PH * PWU + 255) / 256;
    constexpr int NW = (KC * 16 + 255) / 256;
    float pv[NP];
    float4 wv[NW];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int e = tid + 256 * u;
      const int col = e % PWU;
      const int rr = (e / PWU) % PH;
      const int c = c0 + e / (PWU * PH);
      const int iy = iy0 + rr, ix = ix0 + col;
      const bool ok = (e < CCH * PH * PWU) & (c < p.Ctot) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
      const float* ptr = p.src[0].ptr + (long)n * p.src[0].sN;
      int sC = (int)p.src[0].sC, sH = (int)p.src[0].sH, sW = (int)p.src[0].sW, cb = 0;
#pragma unroll
      for (int q = 1; q < PG_MAX_SRC; ++q) {
        const bool sel = (q < p.nsrc) & (c >= p.cstart[q]);
        ptr = sel ? p.src[q].ptr + (long)n * p.src[q].sN : ptr;
        sC = sel ? (int)p.src[q].sC : sC; sH = sel ? (int)p.src[q].sH : sH; sW = sel ? (int)p.src[q].sW : sW;
        cb = sel ? p.cstart[q] : cb;
      }
      const int off = ok ? (c - cb) * sC + iy * sH + ix * sW : 0;
      pv[u] = ldg32(reinterpret_cast<const char*>(ptr), (long)off * 4);
      if (!ok) pv[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int e = (tid + 256 * u) * 4;
      const bool ok = (e < KC * 64) & (c0 + e / (T * 64) < p.Ctot);
      wv[u] = *reinterpret_cast<const float4*>(p.Wt + (ok ? (long)c0 * T * 64 + e : 0));
      if (!ok) wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int e = tid + 256 * u;
      const int col = e % PWU;
      const int rr = (e / PWU) % PH;
      const int cl = e / (PWU * PH);
      const int dst = (S == 1) ? ((cl * PH + rr) * ROWF + col)
                               : ((cl * PH + rr) * ROWF + (col & 1) * PWS + (col >> 1));
      if (e < CCH * PH * PWU) patch[dst] = pv[u];
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int e = (tid + 256 * u) * 4;
      if (e < KC * 64) *reinterpret_cast<float4*>(&wl[e]) = wv[u];
    }
    __syncthreads();
    // ---- MFMA over the chunk: k = cl*T + r*K + s, pairs (kk, kk+1) on the two lane halves
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
      // compile-time patch offsets of k = kk and k = kk+1
      constexpr int dummy = 0; (void)dummy;
      const int k0 = kk, k1 = kk + 1;
      const int c_0 = k0 / T, t_0 = k0 % T, r_0 = t_0 / K, s_0 = t_0 % K;
      const int c_1 = k1 / T, t_1 = k1 % T, r_1 = t_1 / K, s_1 = t_1 % K;
      const int o0 = (S == 1) ? ((c_0 * PH + r_0) * ROWF + s_0) : ((c_0 * PH + r_0) * ROWF + (s_0 & 1) * PWS + (s_0 >> 1));
      const int o1 = (S == 1) ? ((c_1 * PH + r_1) * ROWF + s_1) : ((c_1 * PH + r_1) * ROWF + (s_1 & 1) * PWS + (s_1 >> 1));
      const int koff = lhi ? o1 : o0;
      const float bv = wl[(kk + lhi) * 64 + wn0 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float av = patch[koff + pixoff[i]];
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: + bias, NHWC store
  const int co = wn0 + l31;
  const float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;          // pixel index inside the 32-pixel tile
      const int oy = oy0 + wy0 + 2 * i + (row >> 4), ox = ox0 + (row & 15);
      if (oy < p.Ho && ox < p.Wo) p.out[(((long)n * p.Ho + oy) * p.Wo + ox) * 64 + co] = acc[i][r] + bias;
    }
}

__global__ void repack_small_cin_kernel(const float* W, int T, int Cout, int Cin, float* Wt) {
  // W packed [tap][co][ci]  ->  Wt [ci][tap][co]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * Cout * Cin) return;
  const int co = i % Cout;
  const int tap = (i / Cout) % T;
  const int ci = i / (Cout * T);
  Wt[i] = W[((long)tap * Cout + co) * Cin + ci];
}

}  // namespace pg

extern "C" int pg_repack_small_cin(const float* W, int32_t KH, int32_t KW, int32_t Cout, int32_t Cin, float* Wt,
                                   void* stream) {
  PG_REQUIRE(W && Wt && Cout > 0 && Cin > 0, "pg_repack_small_cin: bad arguments");
  const int n = KH * KW * Cout * Cin;
  PG_KLAUNCH(pg::repack_small_cin_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, KH * KW, Cout,
                     Cin, Wt);
  PG_LAUNCH_OK("pg_repack_small_cin");
  return 0;
}

extern "C" int pg_small_cin_conv(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K,
                                 int32_t stride, int32_t pad, const float* Wt, const float* bias, float* out,
                                 void* stream) {
  PG_REQUIRE(src && nsrc >= 1 && nsrc <= PG_MAX_SRC && Wt && out, "pg_small_cin_conv: bad arguments");
  PG_REQUIRE((K == 3 && stride == 1) || (K == 4 && stride == 2), "pg_small_cin_conv: only k3s1 / k4s2 (got k%d s%d)", K, stride);
  pg::SmallCinK k;
  memset(&k, 0, sizeof(k));
  int c = 0;
  for (int j = 0; j < nsrc; ++j) { k.src[j] = src[j]; k.cstart[j] = c; c += src[j].C; }
  for (int j = nsrc; j <= PG_MAX_SRC; ++j) k.cstart[j] = c;
  k.nsrc = nsrc; k.Ctot = c;
  k.N = N; k.Hi = Hi; k.Wi = Wi; k.pad = pad;
  k.Ho = (Hi + 2 * pad - K) / stride + 1; k.Wo = (Wi + 2 * pad - K) / stride + 1;
  k.Wt = Wt; k.bias = bias; k.out = out;
  k.tiles_x = (k.Wo + 15) / 16; k.tiles_y = (k.Ho + 7) / 8;
  dim3 grid((unsigned)(k.tiles_x * k.tiles_y * N));
  if (K == 3) PG_KLAUNCH((pg::small_cin_conv_kernel<3, 1>), grid, dim3(256), 0, (hipStream_t)stream, k);
  else PG_KLAUNCH((pg::small_cin_conv_kernel<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_small_cin_conv");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Data-gradient of a first-layer convolution towards a FEW of its input channels, written into an NCHW image:
//   out[n][c][y][x] = sum_{r,s,co : (y+pad-r) % S == 0, (x+pad-s) % S == 0} dY[n][(y+pad-r)/S][(x+pad-s)/S][co] * W[r][s][co][c_off+c]
// (gen_update: d loss / d generated image through the discriminator's stem, reference models/pose_gan.py:95-98 -> autograd
// of networks.py:341; also the chain link of the stacked generator through a stage's first convolution).
// As a pg_conv launch this is GEMM-N = 3 in a 32-wide tile with scalar weight loads: 0.7 ms for 3.2 GFLOP at batch 32.
// Streaming form: 16 lanes per output pixel (4 output channels of dY each: one 256-byte row per tap, coalesced), the
// weights of the <= 4 wanted input channels in LDS as [tap][c][co], a 4-stage cross-lane sum per channel.
namespace pg {

struct SmallCinDgradK {
  const void* dY;      // NHWC [N][Ho][Wo][64], fp32 or (DYB) bf16
  const float* W;      // packed [K*K][64][Cin]
  float* out;
  long oN, oC, oH, oW;
  int N, Ho, Wo, Hi, Wi, K, S, pad, Cin, c_off, nc;
};

// K, S compile-time: the tap loops unroll, and for S = 2 only the K/2 x K/2 taps of the pixel's parity class are visited
// (the generic form spent most of its instructions on runtime divisions and parity tests: 73 us for 4 images).
// DYB (round 6): dY is a bf16 tensor (bf16 STORAGE of the generator: the stage-to-stage chain of the stacked generator).
template <int K, int S, bool DYB = false>
__global__ __launch_bounds__(256) void small_cin_dgrad_kernel(const SmallCinDgradK p) {
  __shared__ __attribute__((aligned(16))) float wl[K * K * 4 * 64];          // [tap][c (4)][co]
  for (int i = threadIdx.x; i < K * K * 4 * 64; i += 256) {
    const int co = i & 63, c = (i >> 6) & 3, tap = i >> 8;
    wl[i] = c < p.nc ? p.W[((long)tap * 64 + co) * p.Cin + p.c_off + c] : 0.f;
  }
  __syncthreads();
  const int l16 = threadIdx.x & 15;
  const int hw = p.Hi * p.Wi;
  const long npix = (long)p.N * hw;
  constexpr int NR = (S == 1) ? K : K / 2;                 // taps per axis that can hit a given pixel
  for (long pix = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix < npix; pix += (long)gridDim.x * 16) {
    const int n = (int)(pix / hw);
    const int rem = (int)(pix - (long)n * hw);
    const int y = rem / p.Wi, x = rem - y * p.Wi;
    const int ry0 = (S == 1) ? 0 : ((y + p.pad) & 1), rx0 = (S == 1) ? 0 : ((x + p.pad) & 1);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 g[NR * NR];
    const float* wt[NR * NR];
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
      for (int b = 0; b < NR; ++b) {
        const int r = ry0 + a * S, s2 = rx0 + b * S;
        const int ty = y + p.pad - r, tx = x + p.pad - s2;          // multiples of S by construction
        const int oy = (S == 1) ? ty : (ty >> 1), ox = (S == 1) ? tx : (tx >> 1);
        const bool ok = (ty >= 0) & (tx >= 0) & (oy < p.Ho) & (ox < p.Wo);
        const long off = ok ? (((long)n * p.Ho + oy) * p.Wo + ox) * 64 + l16 * 4 : 0;
        if constexpr (DYB) {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.dY) + off);
          g[a * NR + b] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                      __uint_as_float(u.y & 0xffff0000u));
        } else {
          g[a * NR + b] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.dY) + off);
        }
        if (!ok) g[a * NR + b] = make_float4(0.f, 0.f, 0.f, 0.f);
        wt[a * NR + b] = wl + (r * K + s2) * 256 + l16 * 4;
      }
#pragma unroll
    for (int q = 0; q < NR * NR; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 w4 = *reinterpret_cast<const float4*>(wt[q] + c * 64);
        acc[c] = fmaf(g[q].x, w4.x, fmaf(g[q].y, w4.y, fmaf(g[q].z, w4.z, fmaf(g[q].w, w4.w, acc[c]))));
      }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    if (l16 < p.nc) {
      const float v = l16 == 0 ? acc[0] : (l16 == 1 ? acc[1] : (l16 == 2 ? acc[2] : acc[3]));
      p.out[(long)n * p.oN + (long)l16 * p.oC + (long)y * p.oH + (long)x * p.oW] = v;
    }
  }
}

}  // namespace pg

// io_flags: bit 0 = dY is bf16
extern "C" int pg_small_cin_dgrad_io(const void* dY, const float* W, int32_t N, int32_t Ho, int32_t Wo, int32_t K, int32_t stride,
                                     int32_t pad, int32_t Hi, int32_t Wi, int32_t Cin, int32_t c_off, int32_t nc, float* out,
                                     int64_t oN, int64_t oC, int64_t oH, int64_t oW, int32_t io_flags, void* stream) {
  PG_REQUIRE(dY && W && out && N > 0 && Ho > 0 && Wo > 0 && Hi > 0 && Wi > 0, "pg_small_cin_dgrad: bad arguments");
  PG_REQUIRE(((K == 3 && stride == 1) || (K == 4 && stride == 2)) && nc >= 1 && nc <= 4 && c_off >= 0 && c_off + nc <= Cin,
             "pg_small_cin_dgrad: k3 s1 or k4 s2, 1..4 channels (got k%d s%d nc=%d c_off=%d Cin=%d)", K, stride, nc, c_off, Cin);
  pg::SmallCinDgradK k;
  k.dY = dY; k.W = W; k.out = out; k.oN = oN; k.oC = oC; k.oH = oH; k.oW = oW;
  k.N = N; k.Ho = Ho; k.Wo = Wo; k.Hi = Hi; k.Wi = Wi; k.K = K; k.S = stride; k.pad = pad; k.Cin = Cin; k.c_off = c_off; k.nc = nc;
  long blocks = ((long)N * Hi * Wi + 15) / 16;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const bool dyb = (io_flags & 1) != 0;
  const dim3 grid((unsigned)blocks);
  hipStream_t st = (hipStream_t)stream;
  if (K == 3) {
    if (dyb) PG_KLAUNCH((pg::small_cin_dgrad_kernel<3, 1, true>), grid, dim3(256), 0, st, k);
    else PG_KLAUNCH((pg::small_cin_dgrad_kernel<3, 1, false>), grid, dim3(256), 0, st, k);
  } else {
    if (dyb) PG_KLAUNCH((pg::small_cin_dgrad_kernel<4, 2, true>), grid, dim3(256), 0, st, k);
    else PG_KLAUNCH((pg::small_cin_dgrad_kernel<4, 2, false>), grid, dim3(256), 0, st, k);
  }
  PG_LAUNCH_OK("pg_small_cin_dgrad");
  return 0;
}
extern "C" int pg_small_cin_dgrad(const float* dY, const float* W, int32_t N, int32_t Ho, int32_t Wo, int32_t K, int32_t stride,
                                  int32_t pad, int32_t Hi, int32_t Wi, int32_t Cin, int32_t c_off, int32_t nc, float* out,
                                  int64_t oN, int64_t oC, int64_t oH, int64_t oW, void* stream) {
  return pg_small_cin_dgrad_io(dY, W, N, Ho, Wo, K, stride, pad, Hi, Wi, Cin, c_off, nc, out, oN, oC, oH, oW, 0, stream);
}
