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
};

constexpr int ODG_T = 27;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wc99-designator"
__device__ __attribute__((aligned(16))) const float kOnesO[260] = {[0 ... 259] = 1.0f};
#pragma clang diagnostic pop
__device__ __attribute__((aligned(16))) const float kIdentO[4] = {1.0f, 0.0f, 1.0f, 0.0f};

__global__ __launch_bounds__(256) void out_conv_dgrad_kernel(const OutDgradK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = lane * 4;
  const bool live = cg < p.Ctot;
  const int cgc = live ? cg : 0;
  // this lane's weights
  float w[4][ODG_T + 1];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int q = 0; q < (ODG_T + 1) / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(p.Wt + (long)(cgc + e) * 32 + q * 4);
      w[e][q * 4] = v.x; w[e][q * 4 + 1] = v.y; w[e][q * 4 + 2] = v.z; w[e][q * 4 + 3] = v.w;
    }
  // this lane's destination (constant-index picks: no scratch copy of the kernel argument)
  float* gradp = p.dst[0].grad;
  const float *fwd0 = p.dst[0].fwd, *aff0 = p.dst[0].aff, *mask0 = p.dst[0].mask;
  int C = p.dst[0].C, dact = p.dst[0].act, dacc = p.dst[0].accumulate, cst = 0;
#pragma unroll
  for (int q = 1; q < PG_MAX_SRC; ++q)
    if (q < p.ndst && cgc >= p.dstart[q]) {
      gradp = p.dst[q].grad; fwd0 = p.dst[q].fwd; aff0 = p.dst[q].aff; mask0 = p.dst[q].mask;
      C = p.dst[q].C; dact = p.dst[q].act; dacc = p.dst[q].accumulate; cst = p.dstart[q];
    }
  const int c = cgc - cst;
  const float slope = act_slope(dact);
  const bool has_fwd = fwd0 != nullptr, has_aff = aff0 != nullptr && has_fwd, has_mask = mask0 != nullptr;
  const float* const fwdp = has_fwd ? fwd0 : gradp;            // dummy (valid) reads when there is no activation
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
      gl[u] = p.G[(long)pix * 32 + (lane & 31)];               // one 128-byte row, lanes 32..63 mirror it
      f[u] = *reinterpret_cast<const float4*>(has_fwd ? fwdp + idx[u] : fwdp + c);
      m[u] = *reinterpret_cast<const float4*>(has_mask ? maskp + (long)n * C + c : maskp + (c & 255));
      ab[u] = *reinterpret_cast<const float2*>(affp + affmul * n);
      old[u] = *reinterpret_cast<const float4*>(accum ? gradp + idx[u] : gradp + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < ODG_T; ++t) {
        const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl[u]), t));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(g, w[e][t], acc[e]);
      }
      const float f4[4] = {f[u].x, f[u].y, f[u].z, f[u].w}, m4[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
      const float o4[4] = {old[u].x, old[u].y, old[u].z, old[u].w};
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(f4[e], ab[u].x, ab[u].y) * m4[e];
        r[e] = fmaf(acc[e] * m4[e], act_grad_s(z, dslope), accum ? o4[e] : 0.f);
      }
      if (live && ok[u]) *reinterpret_cast<float4*>(gradp + idx[u]) = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
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
  hipLaunchKernelGGL(pg::out_conv_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_out_conv_dgrad");
  return 0;
}
