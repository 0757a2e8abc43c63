// Deformable skip connection kernels (reference utils/pose_transform.py:16-92), gfx950.
//
//  * mask_pyramid: cv2.resize(mask_HWT,(w,h)) INTER_LINEAR on device — removes the reference's
//    device->host->device round trip (pose_transform.py:84), 4x per generator forward.
//  * warp_mask_max fwd/bwd: repeat x T, affine_grid, grid_sample(bilinear, zeros), * mask, max over T fused into one
//    pass: the (N,T,C,h,w) tensor (10x blow-up, 168 MB/img at level 0) never exists.  HBM-bound: algorithmic bytes
//    = read feat + write out + masks (66.4 MB/img at 256^2).
//    Layout is NHWC, so the 4 bilinear taps of a pixel are 4 contiguous channel runs (coalesced float4 per lane);
//    transforms whose mask is 0 at a pixel contribute an exact 0 to the max and are skipped without touching memory.
//    Coordinates follow the reference's fp32 operation order (fp contraction off): SURVEY.md App. A.2.
#include "common.h"

namespace pg {

constexpr int MAXT = 32;

template <typename TIn>
__global__ __launch_bounds__(256) void mask_pyramid_kernel(const TIn* m, int N, int T, int H0, int W0, int h, int w,
                                                           float* out) {
  const long total = (long)N * h * w * T;
  const double ry = (double)H0 / (double)h, rx = (double)W0 / (double)w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = (int)(i % T);
    long r = i / T;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int n = (int)(r / h);
    const double sy = ((double)y + 0.5) * ry - 0.5, sx = ((double)x + 0.5) * rx - 0.5;
    double fy0 = floor(sy), fx0 = floor(sx);
    double fy = sy - fy0, fx = sx - fx0;
    int y0 = (int)fy0, x0 = (int)fx0;
    if (y0 < 0) fy = 0.0;
    if (x0 < 0) fx = 0.0;
    int y1 = y0 + 1, x1 = x0 + 1;
    y0 = min(max(y0, 0), H0 - 1); y1 = min(max(y1, 0), H0 - 1);
    x0 = min(max(x0, 0), W0 - 1); x1 = min(max(x1, 0), W0 - 1);
    const TIn* b = m + ((long)n * T + t) * H0 * W0;
    const double top = (double)b[(long)y0 * W0 + x0] * (1.0 - fx) + (double)b[(long)y0 * W0 + x1] * fx;
    const double bot = (double)b[(long)y1 * W0 + x0] * (1.0 - fx) + (double)b[(long)y1 * W0 + x1] * fx;
    out[i] = (float)(top * (1.0 - fy) + bot * fy);
  }
}

struct Theta { float t00, t01, t02, t10, t11, t12; };

// normalize_transforms + per-level translation rescale, in the reference's evaluation order
// (pose_transform.py:48-58, 72-76).  No FMA contraction, no re-association.
__device__ __forceinline__ Theta make_theta(const float* wr, int h, int w, int H0, int W0) {
#pragma clang fp contract(off)
  const float mulh = (float)((double)H0 / (double)h);
  const float mulw = (float)((double)W0 / (double)w);
  const float a0 = wr[0], a1 = wr[1], a2 = wr[2] / mulh;
  const float b0 = wr[3], b1 = wr[4], b2 = wr[5] / mulw;
  const float fh = (float)h, fw = (float)w;
  Theta t;
  t.t00 = a0;
  t.t01 = (a1 * fw) / fh;
  t.t02 = ((((a2 * 2.0f) / fh) + t.t00) + t.t01) - 1.0f;
  t.t10 = (b0 * fh) / fw;
  t.t11 = b1;
  t.t12 = ((((b2 * 2.0f) / fw) + t.t10) + t.t11) - 1.0f;
  return t;
}

struct Taps { int x0, y0; float w00, w01, w10, w11; };

__device__ __forceinline__ Taps make_taps(const Theta& t, int i, int j, int h, int w, int align_corners) {
#pragma clang fp contract(off)
  const float fh = (float)h, fw = (float)w;
  float xs, ys;
  if (align_corners) {
    xs = w > 1 ? (((float)j * 2.0f) / (fw - 1.0f)) - 1.0f : 0.0f;
    ys = h > 1 ? (((float)i * 2.0f) / (fh - 1.0f)) - 1.0f : 0.0f;
  } else {
    xs = ((((float)j * 2.0f) + 1.0f) / fw) - 1.0f;
    ys = ((((float)i * 2.0f) + 1.0f) / fh) - 1.0f;
  }
  const float gx = ((t.t00 * xs) + (t.t01 * ys)) + t.t02;
  const float gy = ((t.t10 * xs) + (t.t11 * ys)) + t.t12;
  float ix, iy;
  if (align_corners) {
    ix = ((gx + 1.0f) / 2.0f) * (fw - 1.0f);
    iy = ((gy + 1.0f) / 2.0f) * (fh - 1.0f);
  } else {
    ix = (((gx + 1.0f) * fw) - 1.0f) / 2.0f;
    iy = (((gy + 1.0f) * fh) - 1.0f) / 2.0f;
  }
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float fx = ix - x0f, fy = iy - y0f;
  Taps r;
  // clamp before the int conversion: the "no point" transform puts coordinates ~1e3 px outside
  r.x0 = (int)fminf(fmaxf(x0f, -4.0f), (float)w + 4.0f);
  r.y0 = (int)fminf(fmaxf(y0f, -4.0f), (float)h + 4.0f);
  r.w00 = (1.0f - fx) * (1.0f - fy);
  r.w01 = fx * (1.0f - fy);
  r.w10 = (1.0f - fx) * fy;
  r.w11 = fx * fy;
  return r;
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* feat, const float* aff, const float* warps,
                                                       const float* masks, int T, int C, int h, int w, int H0, int W0,
                                                       int align, float* out, uint8_t* amax) {
  __shared__ Theta th[MAXT];
  const int n = blockIdx.y;
  if (threadIdx.x < T) th[threadIdx.x] = make_theta(warps + ((long)n * T + threadIdx.x) * 8, h, w, H0, W0);
  __syncthreads();
  const int cpp = C >> 2;                  // float4 chunks per pixel
  const long items = (long)h * w * cpp;
  const float a = aff ? aff[2 * n] : 1.f, b = aff ? aff[2 * n + 1] : 0.f;
  const float* fb = feat + (long)n * h * w * C;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const int cc = (int)(it % cpp) * 4;
    const int pix = (int)(it / cpp);
    const int i = pix / w, j = pix - i * w;
    const float* mp = masks + ((long)n * h * w + pix) * T;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {255, 255, 255, 255};
    for (int t = 0; t < T; ++t) {
      const float m = mp[t];
      float cand[4] = {0.f, 0.f, 0.f, 0.f};
      int id = 255;
      if (m != 0.f) {
#pragma clang fp contract(off)
        const Taps tp = make_taps(th[t], i, j, h, w, align);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const float wg[4] = {tp.w00, tp.w01, tp.w10, tp.w11};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int xx = tp.x0 + (k & 1), yy = tp.y0 + (k >> 1);
          if (xx >= 0 && xx < w && yy >= 0 && yy < h) {
            const float4 v = *reinterpret_cast<const float4*>(fb + ((long)yy * w + xx) * C + cc);
            s[0] = s[0] + ((v.x * a) + b) * wg[k];
            s[1] = s[1] + ((v.y * a) + b) * wg[k];
            s[2] = s[2] + ((v.z * a) + b) * wg[k];
            s[3] = s[3] + ((v.w * a) + b) * wg[k];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) cand[e] = s[e] * m;
        id = t;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (cand[e] > best[e]) { best[e] = cand[e]; bi[e] = id; }
    }
    const long o = ((long)n * h * w + pix) * C + cc;
    *reinterpret_cast<float4*>(out + o) = make_float4(best[0], best[1], best[2], best[3]);
    if (amax) *reinterpret_cast<uchar4*>(amax + o) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1],
                                                                  (unsigned char)bi[2], (unsigned char)bi[3]);
  }
}

__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* gout, const uint8_t* amax, const float* warps,
                                                       const float* masks, int T, int C, int h, int w, int H0, int W0,
                                                       int align, float* dfeat) {
  __shared__ Theta th[MAXT];
  const int n = blockIdx.y;
  if (threadIdx.x < T) th[threadIdx.x] = make_theta(warps + ((long)n * T + threadIdx.x) * 8, h, w, H0, W0);
  __syncthreads();
  // One lane per (pixel, channel): every element has exactly ONE selected transform (the forward argmax), so a lane
  // computes the taps of its own transform and issues its four corner atomics once — 4 atomic instructions per 64
  // elements with all lanes active, instead of looping the wave over the T transforms with ~1/T of the lanes live.
  // Lanes of one pixel that share a transform hit consecutive addresses (coalesced within the atomic).
  const long items = (long)h * w * C;
  float* db = dfeat + (long)n * h * w * C;
  const long nb = (long)n * h * w;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const int pix = (int)(it / C);
    const int c = (int)(it - (long)pix * C);
    const int i = pix / w, j = pix - i * w;
    const long o = (nb + pix) * C + c;
    const int t = amax[o];
    const float g = gout[o];
    const bool live = t < T;                       // 255: no transform won (all masks zero) -> no gradient
    const int tt = live ? t : 0;
    const float m = masks[(nb + pix) * T + tt];
    const Taps tp = make_taps(th[tt], i, j, h, w, align);
    const float gm = live ? g * m : 0.f;
    const float wg[4] = {tp.w00, tp.w01, tp.w10, tp.w11};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = tp.x0 + (k & 1), yy = tp.y0 + (k >> 1);
      const float v = gm * wg[k];
      // ReLU'd consumers make about half of the gradients exact zeros
      if (v != 0.f && xx >= 0 && xx < w && yy >= 0 && yy < h) atomicAdd(db + ((long)yy * w + xx) * C + c, v);
    }
  }
}


// Key-point coordinates -> Gaussian heat-maps (reference utils/pose_utils.py:79-86, the step right before the path:
// SURVEY.md §8f row 1).  numpy evaluates exp(-((yy-y)^2 + (xx-x)^2) / (2 sigma^2)) in float64 and stores float32; so
// does this kernel.  A key-point with either coordinate == -1 (MISSING_VALUE) gives a zero map.
__global__ __launch_bounds__(256) void cords_to_map_kernel(const float* cords, int P, int H, int W, double inv2s2,
                                                           float* out, long oN, long oC, long oH, long oW) {
  const int n = blockIdx.z, c = blockIdx.y;
  const float cy = cords[((long)n * P + c) * 2], cx = cords[((long)n * P + c) * 2 + 1];
  const bool missing = cy == -1.f || cx == -1.f;
  float* o = out + (long)n * oN + (long)c * oC;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    const double dy = (double)y - (double)cy, dx = (double)x - (double)cx;
    o[(long)y * oH + (long)x * oW] = missing ? 0.f : (float)exp(-(dy * dy + dx * dx) * inv2s2);
  }
}

}  // namespace pg

using namespace pg;

extern "C" int pg_cords_to_map(const float* cords, int32_t N, int32_t P, int32_t H, int32_t W, float sigma, float* out,
                               int64_t oN, int64_t oC, int64_t oH, int64_t oW, void* stream) {
  PG_REQUIRE(cords && out && N > 0 && P > 0 && H > 0 && W > 0 && sigma > 0.f, "pg_cords_to_map: bad arguments");
  int bx = (H * W + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(cords_to_map_kernel, dim3(bx, P, N), dim3(256), 0, (hipStream_t)stream, cords, P, H, W,
                     1.0 / (2.0 * (double)sigma * (double)sigma), out, (long)oN, (long)oC, (long)oH, (long)oW);
  PG_LAUNCH_OK("pg_cords_to_map");
  return 0;
}


extern "C" int pg_mask_pyramid(const void* masks, int32_t is_f64, int32_t N, int32_t T, int32_t H0, int32_t W0,
                               int32_t h, int32_t w, float* out, void* stream) {
  PG_REQUIRE(masks && out && N > 0 && T > 0 && h > 0 && w > 0, "pg_mask_pyramid: bad arguments");
  const long total = (long)N * h * w * T;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (is_f64)
    hipLaunchKernelGGL(mask_pyramid_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const double*)masks, N, T, H0, W0, h, w, out);
  else
    hipLaunchKernelGGL(mask_pyramid_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float*)masks, N, T, H0, W0, h, w, out);
  PG_LAUNCH_OK("pg_mask_pyramid");
  return 0;
}

static int warp_bwd_grid(int C, int h, int w) {
  long b = ((long)h * w * C + 255) / 256;
  if (b > 8192) b = 8192;
  return (int)b;
}

static int warp_grid(int C, int h, int w) {
  const long items = (long)h * w * (C / 4);
  long b = (items + 255) / 256;
  if (b > 2048) b = 2048;
  return (int)b;
}

extern "C" int pg_warp_mask_max_fwd(const float* feat, const float* aff, const float* warps, const float* lvl_masks,
                                    int32_t N, int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0,
                                    int32_t align_corners, float* out, uint8_t* argmax, void* stream) {
  PG_REQUIRE(feat && warps && lvl_masks && out, "pg_warp_mask_max_fwd: null pointer");
  PG_REQUIRE(T >= 1 && T <= MAXT && C % 4 == 0 && N > 0, "pg_warp_mask_max_fwd: need T<=32, C%%4==0 (T=%d C=%d)", T, C);
  hipLaunchKernelGGL(warp_fwd_kernel, dim3(warp_grid(C, h, w), N), dim3(256), 0, (hipStream_t)stream, feat, aff, warps,
                     lvl_masks, T, C, h, w, H0, W0, align_corners, out, argmax);
  PG_LAUNCH_OK("pg_warp_mask_max_fwd");
  return 0;
}

extern "C" int pg_warp_mask_max_bwd(const float* gout, const uint8_t* argmax, const float* warps,
                                    const float* lvl_masks, int32_t N, int32_t T, int32_t C, int32_t h, int32_t w,
                                    int32_t H0, int32_t W0, int32_t align_corners, float* dfeat, void* stream) {
  PG_REQUIRE(gout && argmax && warps && lvl_masks && dfeat, "pg_warp_mask_max_bwd: null pointer");
  PG_REQUIRE(T >= 1 && T <= MAXT && C % 4 == 0 && N > 0, "pg_warp_mask_max_bwd: need T<=32, C%%4==0");
  hipLaunchKernelGGL(warp_bwd_kernel, dim3(warp_bwd_grid(C, h, w), N), dim3(256), 0, (hipStream_t)stream, gout, argmax,
                     warps, lvl_masks, T, C, h, w, H0, W0, align_corners, dfeat);
  PG_LAUNCH_OK("pg_warp_mask_max_bwd");
  return 0;
}
