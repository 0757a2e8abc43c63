// Data-gradient of the generator's 256->3 output convolution (reference models/networks.py:228: ReLU -> Conv2d(k3,p1)
// -> Tanh; autograd of `loss.backward()`, models/pose_gan.py:170) from the im2col'd 3-channel gradient
//     G[pixel][t],  t = (tap, co) < 27, row pitch 32           (pg_im2col_taps, also the weight-gradient operand)
//     dX[pixel][ci] = sum_t G[pixel][t] * Wt[ci][t]             Wt = the weight viewed as [Cin][27 -> 32]
// scattered over the virtual concat with act'(fwd) / dropout mask like pg_conv's data-gradient epilogue.
// K = 27: there is no GEMM here — as a pg_conv launch this was ONE K tile followed by a 4-byte-per-lane scatter
// epilogue (285 us for 537 MB).  This kernel is the streaming form: a wave owns one pixel at a time, a lane owns 4
// consecutive input channels and keeps their 27 x 4 weights in registers for the whole launch; the pixel's 27 gradient
// values arrive with ONE coalesced 128-byte load and are broadcast lane -> SGPR (v_readlane), so the contraction is
// 108 FMAs with a scalar operand per pixel and wave, and every forward value / mask / result moves as 16 bytes per lane.
#include "common.h"

namespace pg {

struct OutDgradK {
  const float* G;          // [npix][32]
  const float* Wt;         // [Ctot][32]
  int npix, ppix, Ctot;
  pg_dst_t dst[PG_MAX_SRC];
  int ndst;
  int dstart[PG_MAX_SRC + 1];
  float* wpart;            // WG = true: per-workgroup partial weight gradients [blocks][Ctot][28] (round 3)
  int g_bf16_pitch;        // 0: G is fp32 with row pitch 32; else G is bf16 with this row pitch (64)
  const float* dpre;       // DIRECT = true: the NCHW [N][3][H][W] gradient itself — the 27 (tap, channel) values of a pixel are
  int H, W;                //   gathered from it, no im2col'd copy exists (round 3)
};

constexpr int ODG_T = 27;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wc99-designator"
__device__ __attribute__((aligned(16))) const float kOnesO[260] = {[0 ... 259] = 1.0f};
#pragma clang diagnostic pop
__device__ __attribute__((aligned(16))) const float kIdentO[4] = {1.0f, 0.0f, 1.0f, 0.0f};

// 4 consecutive elements (element index idx) of an fp32 or bf16 tensor; the branch is uniform per destination
template <bool bf>
__device__ __forceinline__ float4 ld4_dt(const float* base, long idx) {
  if constexpr (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(base + idx);
}

// WG = true (round 3, bf16 STORAGE): the same pass also accumulates the WEIGHT gradient dW[t][ci] = sum_pixels G[pixel][t] *
// x[pixel][ci].  With bf16 storage every destination's `fwd` is the ACTIVATED bf16 operand the forward contraction read
// (relu(z): its sign gives act', its value IS x), so g (27 scalars per pixel, already broadcast) and x (4 channels per
// lane, already loaded) are both in registers: 108 more FMAs per pixel and lane instead of a second kernel that reads
// the three activation tensors again (fp32 kernel before: 0.59 ms + 1.07 GB at batch 32).  Per-lane sums are merged per
// workgroup through LDS and written to a partial buffer that out_conv_wgrad_reduce_kernel adds into dW.
// DG: compute and store the data gradient; BF: every destination's gradient / forward tensor is bf16 (compile-time, so that
// the batched loads stay straight-line code).  Instantiated as <DG, !WG, fp32> (the fp32 paths), <DG, !WG, bf16> and
// <!DG, WG, bf16>: two lean passes beat one fused pass (256 VGPRs, one wave per SIMD: 3.07 ms at batch 32 against
// 0.4 + 0.4 ms).
template <bool DG, bool WG, bool BF, bool DIRECT = false>
__global__ __launch_bounds__(256) void out_conv_dgrad_kernel(const OutDgradK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // DIRECT: lane t < 27 = (tap, channel) = (t / 3, t % 3) reads dpre[n][c][y - (r - 1)][x - (s - 1)] (pg_im2col_taps' formula)
  const int d_c = (lane % 27) % 3, d_tap = (lane % 27) / 3;
  const int d_dy = 1 - d_tap / 3, d_dx = 1 - d_tap % 3;
  typedef float f32x2w __attribute__((ext_vector_type(2)));
  f32x2w aw01[WG ? ODG_T : 1], aw23[WG ? ODG_T : 1];          // channel pairs (packed fp32 FMAs)
  if constexpr (WG) {
#pragma unroll
    for (int t = 0; t < ODG_T; ++t) { aw01[t] = f32x2w{0.f, 0.f}; aw23[t] = f32x2w{0.f, 0.f}; }
  }
  const int cg = lane * 4;
  const bool live = cg < p.Ctot;
  const int cgc = live ? cg : 0;
  // this lane's weights
  float w[DG ? 4 : 1][DG ? ODG_T + 1 : 1];
  if constexpr (DG) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < (ODG_T + 1) / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p.Wt + (long)(cgc + e) * 32 + q * 4);
        w[e][q * 4] = v.x; w[e][q * 4 + 1] = v.y; w[e][q * 4 + 2] = v.z; w[e][q * 4 + 3] = v.w;
      }
  }
  // this lane's destination (constant-index picks: no scratch copy of the kernel argument)
  float* gradp = p.dst[0].grad;
  const float *fwd0 = p.dst[0].fwd, *aff0 = p.dst[0].aff, *mask0 = p.dst[0].mask;
  int C = p.dst[0].C, dact = p.dst[0].act, dacc = p.dst[0].accumulate, cst = 0, dfl = p.dst[0].flags;
#pragma unroll
  for (int q = 1; q < PG_MAX_SRC; ++q)
    if (q < p.ndst && cgc >= p.dstart[q]) {
      gradp = p.dst[q].grad; fwd0 = p.dst[q].fwd; aff0 = p.dst[q].aff; mask0 = p.dst[q].mask;
      C = p.dst[q].C; dact = p.dst[q].act; dacc = p.dst[q].accumulate; cst = p.dstart[q]; dfl = p.dst[q].flags;
    }
  (void)dfl;
  constexpr bool gbf = BF;
  const int c = cgc - cst;
  const float slope = act_slope(dact);
  const bool has_fwd = fwd0 != nullptr, has_aff = aff0 != nullptr && has_fwd, has_mask = mask0 != nullptr;
  const float* const fwdp = has_fwd ? fwd0 : gradp;            // dummy (valid) reads when there is no activation
  constexpr bool fbf = BF;
  const float dslope = has_fwd ? slope : 1.f;                  // slope 1: act' == 1 whatever was read

  // absent mask / affine / accumulation: loads stay unconditional (a load under a divergent branch serialises its
  // latency) but go to a per-lane-constant dummy address that stays in L1/L2
  const float* const maskp = has_mask ? mask0 : kOnesO;
  const float* const affp = has_aff ? aff0 : kIdentO;
  const int affmul = has_aff ? 2 : 0;
  const bool accum = dacc != 0;

  constexpr int U = 4;                                         // consecutive pixels per wave and step (loads in flight)
  const int stride = gridDim.x * 4 * U;
  for (int base0 = (blockIdx.x * 4 + wave) * U; base0 < p.npix; base0 += stride) {
    const int base = __builtin_amdgcn_readfirstlane(base0);
    float gl[U];
    float4 f[U], m[U], old[U];
    float2 ab[U];
    long idx[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = base + u < p.npix;                               // wave-uniform
      const int pix = ok[u] ? base + u : base;
      const int n = pix / p.ppix;
      idx[u] = (long)pix * C + c;
      if constexpr (DIRECT) {
        const int rem = pix - n * p.ppix;
        const int y = rem / p.W, x = rem - y * p.W;            // wave-uniform
        const int yy = y + d_dy, xx = x + d_dx;
        const bool in = (yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W);
        const float v = p.dpre[((long)(n * 3 + d_c) * p.H + (in ? yy : y)) * p.W + (in ? xx : x)];
        gl[u] = in ? v : 0.f;
      } else if (p.g_bf16_pitch)                               // (wave-uniform) bf16 rows of the bf16 data path
        gl[u] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p.G)[(long)pix * p.g_bf16_pitch + (lane & 31)] << 16);
      else
        gl[u] = p.G[(long)pix * 32 + (lane & 31)];             // one 128-byte row, lanes 32..63 mirror it
      f[u] = ld4_dt<fbf>(fwdp, has_fwd ? idx[u] : (long)c);
      m[u] = *reinterpret_cast<const float4*>(has_mask ? maskp + (long)n * C + c : maskp + (c & 255));
      ab[u] = *reinterpret_cast<const float2*>(affp + affmul * n);
      if constexpr (DG) old[u] = ld4_dt<gbf>(gradp, accum ? idx[u] : (long)c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (DG) {
        // packed fp32 FMAs (v_pk_fma_f32: two channels per instruction, the broadcast gradient value in both halves): the
        // loop is VALU-bound (27 readlanes + 108 FMAs per pixel and wave), 54 packed instead of 108 scalar FMAs
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < ODG_T; ++t) {
          const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl[u]), t));
          const f32x2 gg = {g, g};
          const f32x2 w01 = {w[0][t], w[1][t]}, w23 = {w[2][t], w[3][t]};
          a01 = __builtin_elementwise_fma(gg, w01, a01);
          a23 = __builtin_elementwise_fma(gg, w23, a23);
        }
        acc[0] = a01[0]; acc[1] = a01[1]; acc[2] = a23[0]; acc[3] = a23[1];
      }
      const float f4[4] = {f[u].x, f[u].y, f[u].z, f[u].w}, m4[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
      const float o4[4] = {old[u].x, old[u].y, old[u].z, old[u].w};
      if constexpr (WG) {
        if (ok[u]) {                                           // wave-uniform
          const f32x2w x01 = {f4[0], f4[1]}, x23 = {f4[2], f4[3]};
#pragma unroll
          for (int t = 0; t < ODG_T; ++t) {
            const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl[u]), t));
            const f32x2w gg = {g, g};
            aw01[t] = __builtin_elementwise_fma(gg, x01, aw01[t]);
            aw23[t] = __builtin_elementwise_fma(gg, x23, aw23[t]);
          }
        }
      }
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(f4[e], ab[u].x, ab[u].y) * m4[e];
        r[e] = fmaf(acc[e] * m4[e], act_grad_s(z, dslope), accum ? o4[e] : 0.f);
      }
      if (DG && live && ok[u]) {
        if constexpr (gbf) {
          uint2 pk;
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.x) : "v"(r[0]), "v"(r[1]));
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.y) : "v"(r[2]), "v"(r[3]));
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(gradp) + idx[u]) = pk;
        } else {
          *reinterpret_cast<float4*>(gradp + idx[u]) = make_float4(r[0], r[1], r[2], r[3]);
        }
      }
    }
  }
  if constexpr (WG) {
    // merge the four waves' sums (a lane's 4 x 27 values per round through LDS), then one partial block per workgroup
    __shared__ float red[64 * 4 * 28];
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < ODG_T; ++t) {
            float* q = red + (lane * 4 + e) * 28 + t;
            *q = (wv == 0 ? 0.f : *q) + (e < 2 ? aw01[t][e] : aw23[t][e - 2]);
          }
      }
      __syncthreads();
    }
    float* dstp = p.wpart + (long)blockIdx.x * p.Ctot * 28;
    for (int i = threadIdx.x; i < p.Ctot * 28; i += 256) dstp[i] = red[i];
  }
}

// dW[t][ci] += sum over workgroup partials [nb][Ctot][28]; blockIdx.y = slice of the partial list (float atomics into dW: one
// serial walk over 1024 partials per thread took 160 us at batch 4)
__global__ __launch_bounds__(256) void out_conv_wgrad_reduce_kernel(const float* part, int nb, int Ctot, float* dW) {
  const int i = blockIdx.x * 256 + threadIdx.x;       // (ci, t)
  if (i >= Ctot * 28) return;
  const int ci = i / 28, t = i - ci * 28;
  if (t >= ODG_T) return;
  const int per = (nb + (int)gridDim.y - 1) / (int)gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nb, b0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += part[(long)b * Ctot * 28 + i]; s1 += part[(long)(b + 1) * Ctot * 28 + i];
    s2 += part[(long)(b + 2) * Ctot * 28 + i]; s3 += part[(long)(b + 3) * Ctot * 28 + i];
  }
  for (; b < b1; ++b) s0 += part[(long)b * Ctot * 28 + i];
  if (b1 > b0) atomicAdd(&dW[(long)t * Ctot + ci], (s0 + s1) + (s2 + s3));
}

}  // namespace pg

extern "C" int pg_out_conv_dgrad(const float* G, const float* Wt, int32_t N, int32_t H, int32_t W,
                                 const pg_dst_t* dst, int32_t ndst, void* stream) {
  PG_REQUIRE(G && Wt && dst && ndst >= 1 && ndst <= PG_MAX_SRC && N > 0 && H > 0 && W > 0, "pg_out_conv_dgrad: bad arguments");
  pg::OutDgradK k;
  memset(&k, 0, sizeof(k));
  k.G = G; k.Wt = Wt;
  int c = 0;
  for (int j = 0; j < ndst; ++j) {
    k.dst[j] = dst[j]; k.dstart[j] = c; c += dst[j].C;
    PG_REQUIRE(dst[j].C % 4 == 0 && ((size_t)dst[j].grad & 15) == 0 && ((size_t)dst[j].fwd & 15) == 0 &&
               ((size_t)dst[j].mask & 15) == 0, "pg_out_conv_dgrad: destinations need C %% 4 == 0 and 16-byte alignment");
  }
  for (int j = ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
  k.ndst = ndst; k.Ctot = c;
  PG_REQUIRE(c <= 256, "pg_out_conv_dgrad: at most 256 input channels (one wave per pixel), got %d", c);
  PG_REQUIRE((double)N * H * W < 2147483648.0 / 32, "pg_out_conv_dgrad: too many pixels");
  k.npix = N * H * W; k.ppix = H * W;
  long blocks = (k.npix + 15) / 16;
  if (blocks > 256 * 12) blocks = 256 * 12;     // 108 weight registers per lane are loaded once per workgroup
  int nbf = 0;
  for (int j = 0; j < ndst; ++j) {
    const int want = PG_DST_GRAD_BF16 | (dst[j].fwd ? PG_DST_FWD_BF16 : 0);
    if (dst[j].flags != 0) { PG_REQUIRE((dst[j].flags & want) == want, "pg_out_conv_dgrad: a destination mixes fp32 and bf16 tensors"); ++nbf; }
  }
  PG_REQUIRE(nbf == 0 || nbf == ndst, "pg_out_conv_dgrad: destinations must be all fp32 or all bf16");
  if (nbf) PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  else PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_out_conv_dgrad");
  return 0;
}

// Data gradient AND weight gradient of the output convolution (bf16 STORAGE on the bf16 data path), two streaming passes
// over the same descriptors: every destination's `fwd` must be the ACTIVATED bf16 operand of the forward pass (no aff /
// mask on it), gradients are bf16, dW is [27][Ctot] fp32 (accumulated), `workspace` holds workspace_floats >= 64 * Ctot * 28
// floats of per-workgroup partials.
extern "C" int pg_out_conv_bwd_direct(const float* G, int32_t g_is_dpre, const float* Wt, int32_t N, int32_t H, int32_t W,
                                      const pg_dst_t* dst, int32_t ndst, float* dW, float* workspace, int64_t workspace_floats,
                                      void* wg_stream, void* stream);
extern "C" int pg_out_conv_dgrad_wgrad(const float* G, const float* Wt, int32_t N, int32_t H, int32_t W, const pg_dst_t* dst,
                                       int32_t ndst, float* dW, float* workspace, int64_t workspace_floats, void* stream) {
  return pg_out_conv_bwd_direct(G, 0, Wt, N, H, W, dst, ndst, dW, workspace, workspace_floats, nullptr, stream);
}

// g_is_dpre = 1: `G` is the NCHW (N,3,H,W) gradient wrt the pre-tanh output itself; the kernels gather each pixel's 27 (tap,
// channel) values from it (no pg_im2col_taps pass, no [pixel][32] tensor).  wg_stream: stream of the weight-gradient pass (NULL:
// `stream`); the caller orders it after the producer of G.
extern "C" int pg_out_conv_bwd_direct(const float* G, int32_t g_is_dpre, const float* Wt, int32_t N, int32_t H, int32_t W,
                                      const pg_dst_t* dst, int32_t ndst, float* dW, float* workspace, int64_t workspace_floats,
                                      void* wg_stream, void* stream) {
  PG_REQUIRE(G && Wt && dst && dW && workspace && ndst >= 1 && ndst <= PG_MAX_SRC && N > 0 && H > 0 && W > 0,
             "pg_out_conv_dgrad_wgrad: bad arguments");
  pg::OutDgradK k;
  memset(&k, 0, sizeof(k));
  k.G = G; k.Wt = Wt;
  if (g_is_dpre) { k.dpre = G; k.H = H; k.W = W; }
  int c = 0;
  for (int j = 0; j < ndst; ++j) {
    k.dst[j] = dst[j]; k.dstart[j] = c; c += dst[j].C;
    PG_REQUIRE(dst[j].C % 4 == 0 && ((size_t)dst[j].grad & 15) == 0 && ((size_t)dst[j].fwd & 15) == 0 && dst[j].fwd != nullptr &&
               dst[j].aff == nullptr && dst[j].mask == nullptr && dst[j].flags == (PG_DST_GRAD_BF16 | PG_DST_FWD_BF16),
               "pg_out_conv_dgrad_wgrad: destinations need C %% 4 == 0, 16-byte alignment, bf16 tensors and an activated forward operand");
  }
  for (int j = ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
  k.ndst = ndst; k.Ctot = c;
  PG_REQUIRE(c <= 256, "pg_out_conv_dgrad_wgrad: at most 256 input channels (one wave per pixel), got %d", c);
  PG_REQUIRE((double)N * H * W < 2147483648.0 / 32, "pg_out_conv_dgrad_wgrad: too many pixels");
  k.npix = N * H * W; k.ppix = H * W;
  long blocks = (k.npix + 15) / 16;
  if (blocks > 256 * 12) blocks = 256 * 12;
  hipStream_t st = (hipStream_t)stream, wst = wg_stream ? (hipStream_t)wg_stream : st;
  if (g_is_dpre) PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, k);
  else PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, true, false>), dim3((unsigned)blocks), dim3(256), 0, st, k);
  PG_LAUNCH_OK("pg_out_conv_dgrad_wgrad (data gradient)");
  blocks = (k.npix + 15) / 16;
  long cap = workspace_floats / ((long)c * 28);
  if (cap > 1024) cap = 1024;
  PG_REQUIRE(cap >= 64, "pg_out_conv_dgrad_wgrad: workspace too small");
  if (blocks > cap) blocks = cap;
  k.wpart = workspace;
  if (g_is_dpre) PG_KLAUNCH((pg::out_conv_dgrad_kernel<false, true, true, true>), dim3((unsigned)blocks), dim3(256), 0, wst, k);
  else PG_KLAUNCH((pg::out_conv_dgrad_kernel<false, true, true, false>), dim3((unsigned)blocks), dim3(256), 0, wst, k);
  PG_LAUNCH_OK("pg_out_conv_dgrad_wgrad (weight gradient)");
  PG_KLAUNCH(pg::out_conv_wgrad_reduce_kernel, dim3((c * 28 + 255) / 256, 16), dim3(256), 0, wst, workspace, (int)blocks, c, dW);
  PG_LAUNCH_OK("pg_out_conv_dgrad_wgrad (reduce)");
  return 0;
}

// Weight gradient of the output convolution alone (bf16 data path, round 3): dW[27][Ctot] += sum_pixels G[pixel][t] * x[pixel][ci]
// with G the bf16 im2col'd gradient (row pitch g_pitch, pg_im2col_taps_bf16) and x the ACTIVATED bf16 operands of the forward
// pass, passed as the `fwd` tensors of `dst` (grad / aff / mask unused).  The data gradient of the same layer runs as a
// bf16 contraction (pg_conv with the padded weight).
extern "C" int pg_out_conv_wgrad_bf16(const void* G_bf16, int32_t g_pitch, int32_t N, int32_t H, int32_t W, const pg_dst_t* dst,
                                      int32_t ndst, float* dW, float* workspace, int64_t workspace_floats, void* stream) {
  PG_REQUIRE(G_bf16 && dst && dW && workspace && ndst >= 1 && ndst <= PG_MAX_SRC && N > 0 && H > 0 && W > 0 && g_pitch >= 32,
             "pg_out_conv_wgrad_bf16: bad arguments");
  pg::OutDgradK k;
  memset(&k, 0, sizeof(k));
  k.G = reinterpret_cast<const float*>(G_bf16); k.Wt = nullptr; k.g_bf16_pitch = g_pitch;
  int c = 0;
  for (int j = 0; j < ndst; ++j) {
    k.dst[j] = dst[j]; k.dstart[j] = c; c += dst[j].C;
    k.dst[j].grad = const_cast<float*>(dst[j].fwd);          // never written (DG = false); keeps the dummy addresses valid
    k.dst[j].accumulate = 0;
    PG_REQUIRE(dst[j].C % 4 == 0 && ((size_t)dst[j].fwd & 15) == 0 && dst[j].fwd != nullptr && dst[j].aff == nullptr &&
               dst[j].mask == nullptr && (dst[j].flags & PG_DST_FWD_BF16),
               "pg_out_conv_wgrad_bf16: sources need C %% 4 == 0, 16-byte alignment and an activated bf16 operand");
  }
  for (int j = ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
  k.ndst = ndst; k.Ctot = c;
  PG_REQUIRE(c <= 256, "pg_out_conv_wgrad_bf16: at most 256 input channels (one wave per pixel), got %d", c);
  PG_REQUIRE((double)N * H * W < 2147483648.0 / 64, "pg_out_conv_wgrad_bf16: too many pixels");
  k.npix = N * H * W; k.ppix = H * W;
  long blocks = (k.npix + 15) / 16;
  long cap = workspace_floats / ((long)c * 28);
  if (cap > 1024) cap = 1024;
  PG_REQUIRE(cap >= 64, "pg_out_conv_wgrad_bf16: workspace too small");
  if (blocks > cap) blocks = cap;
  k.wpart = workspace;
  PG_KLAUNCH((pg::out_conv_dgrad_kernel<false, true, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_out_conv_wgrad_bf16");
  PG_KLAUNCH(pg::out_conv_wgrad_reduce_kernel, dim3((c * 28 + 255) / 256, 16), dim3(256), 0, (hipStream_t)stream, workspace,
                     (int)blocks, c, dW);
  PG_LAUNCH_OK("pg_out_conv_wgrad_bf16 (reduce)");
  return 0;
}
