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
#include <cstdlib>

#include "common.h"

namespace pg {

constexpr int MAXT = 32;

// bf16 STORAGE (round 3, bf16 data path): 4 consecutive channels of an fp32 (16 B) or bf16 (8 B) NHWC tensor at a BYTE offset
template <bool BF>
__device__ __forceinline__ float4 wld4(const char* base, size_t byte_off) {
  if constexpr (BF) {
    const uint2 u = *reinterpret_cast<const uint2*>(base + byte_off);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  } else {
    return *reinterpret_cast<const float4*>(base + byte_off);
  }
}
__device__ __forceinline__ unsigned wpack_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// V consecutive channels (V = 4: 16 B fp32 / 8 B bf16; V = 8: bf16 only, 16 B)
template <bool BF, int V>
__device__ __forceinline__ void wldv(const char* base, size_t byte_off, float (&v)[V]) {
  if constexpr (V == 8) {
    static_assert(BF, "8 channels per lane: bf16 tensors only");
    const uint4 u = *reinterpret_cast<const uint4*>(base + byte_off);
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
  } else {
    const float4 f = wld4<BF>(base, byte_off);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  }
}
template <bool BF, int V>
__device__ __forceinline__ void wstv(char* base, size_t byte_off, const float (&v)[V]);

template <bool BF>
__device__ __forceinline__ void wst4(char* base, size_t byte_off, float4 v) {
  if constexpr (BF) {
    uint2 u;
    u.x = wpack_bf16(v.x, v.y); u.y = wpack_bf16(v.z, v.w);
    *reinterpret_cast<uint2*>(base + byte_off) = u;
  } else {
    *reinterpret_cast<float4*>(base + byte_off) = v;
  }
}

// cv2.resize(mask_{H0 x W0 x T}, (w, h)), INTER_LINEAR (utils/pose_transform.py:84-87), planes (N, T, H0, W0) in,
// channel-last (N, h, w, T) out.  Round 4: one workgroup = 64 consecutive output pixels of one row; a lane keeps ONE x (its
// column taps and weights are computed once, in double as the oracle does) and walks the T planes four at a time, so every plane
// is read along x (the first version mapped consecutive lanes to consecutive t: ten 4-byte reads from ten planes 256 KB apart,
// 0.7 TB/s on the identity level); the 64 x T results leave through LDS as one contiguous run.
template <typename TIn>
__global__ __launch_bounds__(256) void mask_pyramid_kernel(const TIn* m, int N, int T, int H0, int W0, int h, int w,
                                                           float* out) {
  __shared__ float tile[64 * 33];
  const int n = blockIdx.z, y = blockIdx.y, xb = blockIdx.x * 64;
  const int xl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int x = min(xb + xl, w - 1);
  const double ry = (double)H0 / (double)h, rx = (double)W0 / (double)w;
  const double sy = ((double)y + 0.5) * ry - 0.5, sx = ((double)x + 0.5) * rx - 0.5;
  double fy0 = floor(sy), fx0 = floor(sx);
  double fy = sy - fy0, fx = sx - fx0;
  int y0 = (int)fy0, x0 = (int)fx0;
  if (y0 < 0) fy = 0.0;
  if (x0 < 0) fx = 0.0;
  int y1 = y0 + 1, x1 = x0 + 1;
  y0 = min(max(y0, 0), H0 - 1); y1 = min(max(y1, 0), H0 - 1);
  x0 = min(max(x0, 0), W0 - 1); x1 = min(max(x1, 0), W0 - 1);
  const long o00 = (long)y0 * W0 + x0, o01 = (long)y0 * W0 + x1, o10 = (long)y1 * W0 + x0, o11 = (long)y1 * W0 + x1;
  const int nx = min(64, w - xb);
  for (int t0 = 0; t0 < T; t0 += 32) {             // T <= 32 per LDS pass (MAXT)
    const int tn = min(32, T - t0);
    for (int t = tl; t < tn; t += 4) {
      const TIn* b = m + ((long)n * T + t0 + t) * H0 * W0;
      const double top = (double)b[o00] * (1.0 - fx) + (double)b[o01] * fx;
      const double bot = (double)b[o10] * (1.0 - fx) + (double)b[o11] * fx;
      tile[xl * 33 + t] = (float)(top * (1.0 - fy) + bot * fy);
    }
    __syncthreads();
    float* o = out + (((long)n * h + y) * w + xb) * T + t0;
    for (int i = threadIdx.x; i < nx * tn; i += 256) {
      const int px = i / tn, t = i - px * tn;
      o[(long)px * T + t] = tile[px * 33 + t];
    }
    __syncthreads();
  }
}

// Bounding box of the non-zero pixels of every (sample, transform) mask plane (N, T, H0, W0): out[(n*T+t)*4 ..] = ymin, ymax, xmin,
// xmax; an empty plane gives ymax < ymin.  One workgroup per plane, coalesced scan (84 MB at batch 32: ~20 us).
template <typename TIn>
__global__ __launch_bounds__(1024) void mask_bbox_kernel(const TIn* m, int H0, int W0, int* out) {
  __shared__ int red[4][1024];
  const TIn* b = m + (long)blockIdx.x * H0 * W0;
  int y0 = H0, y1 = -1, x0 = W0, x1 = -1;
  const int total = H0 * W0;
  // four elements per lane and step (16-byte loads when the plane allows it): 320 planes of 256 KB are read in ~25 us
  if (sizeof(TIn) == 4 && (W0 & 3) == 0 && (((size_t)b) & 15) == 0) {
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (int i = threadIdx.x; i < total / 4; i += 1024) {
      const float4 v = b4[i];
      if (v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f) {
        const int e = i * 4, y = e / W0, x = e - y * W0;          // W0 % 4 == 0: the four share a row
        y0 = min(y0, y); y1 = max(y1, y);
        const int xa = v.x != 0.f ? x : (v.y != 0.f ? x + 1 : (v.z != 0.f ? x + 2 : x + 3));
        const int xb = v.w != 0.f ? x + 3 : (v.z != 0.f ? x + 2 : (v.y != 0.f ? x + 1 : x));
        x0 = min(x0, xa); x1 = max(x1, xb);
      }
    }
  } else {
    for (int i = threadIdx.x; i < total; i += 1024) {
      if (b[i] != (TIn)0) {
        const int y = i / W0, x = i - y * W0;
        y0 = min(y0, y); y1 = max(y1, y); x0 = min(x0, x); x1 = max(x1, x);
      }
    }
  }
  red[0][threadIdx.x] = y0; red[1][threadIdx.x] = y1; red[2][threadIdx.x] = x0; red[3][threadIdx.x] = x1;
  __syncthreads();
  for (int s_ = 512; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) {
      red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + s_]);
      red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + s_]);
      red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + s_]);
      red[3][threadIdx.x] = max(red[3][threadIdx.x], red[3][threadIdx.x + s_]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) out[(long)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
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

// un-floored source coordinates of output pixel (i, j) under theta — the reference's fp32 evaluation order
__device__ __forceinline__ void make_coords(const Theta& t, int i, int j, int h, int w, int align_corners, float& ix, float& iy) {
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
  if (align_corners) {
    ix = ((gx + 1.0f) / 2.0f) * (fw - 1.0f);
    iy = ((gy + 1.0f) / 2.0f) * (fh - 1.0f);
  } else {
    ix = (((gx + 1.0f) * fw) - 1.0f) / 2.0f;
    iy = (((gy + 1.0f) * fh) - 1.0f) / 2.0f;
  }
}

__device__ __forceinline__ Taps make_taps(const Theta& t, int i, int j, int h, int w, int align_corners) {
#pragma clang fp contract(off)
  float ix, iy;
  make_coords(t, i, j, h, w, align_corners, ix, iy);
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

// The same with the normalised grid coordinates of the column / row given (per-image tables: the two divisions of
// make_coords depend on j or i alone).  Identical operations in identical order -> identical results.
__device__ __forceinline__ float warp_norm_coord(int k, int n, int align_corners) {
#pragma clang fp contract(off)
  const float fn = (float)n;
  return align_corners ? (n > 1 ? (((float)k * 2.0f) / (fn - 1.0f)) - 1.0f : 0.0f) : ((((float)k * 2.0f) + 1.0f) / fn) - 1.0f;
}
__device__ __forceinline__ Taps make_taps_xy(const Theta& t, float xs, float ys, int h, int w, int align_corners) {
#pragma clang fp contract(off)
  const float fh = (float)h, fw = (float)w;
  const float gx = ((t.t00 * xs) + (t.t01 * ys)) + t.t02;
  const float gy = ((t.t10 * xs) + (t.t11 * ys)) + t.t12;
  float ix, iy;
  if (align_corners) { ix = ((gx + 1.0f) / 2.0f) * (fw - 1.0f); iy = ((gy + 1.0f) / 2.0f) * (fh - 1.0f); }
  else { ix = (((gx + 1.0f) * fw) - 1.0f) / 2.0f; iy = (((gy + 1.0f) * fh) - 1.0f) / 2.0f; }
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float fx = ix - x0f, fy = iy - y0f;
  Taps r;
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

template <bool BF, int V>
__device__ __forceinline__ void wstv(char* base, size_t byte_off, const float (&v)[V]) {
  if constexpr (V == 8) {
    uint4 u = make_uint4(wpack_bf16(v[0], v[1]), wpack_bf16(v[2], v[3]), wpack_bf16(v[4], v[5]), wpack_bf16(v[6], v[7]));
    *reinterpret_cast<uint4*>(base + byte_off) = u;
  } else {
    wst4<BF>(base, byte_off, make_float4(v[0], v[1], v[2], v[3]));
  }
}

// ---- backward.  Gather form: an input pixel (X, Y) collects from the output pixels whose bilinear footprint under some
// transform covers it.  The affine map is inverted per (sample, transform): the pre-image of the 2 x 2 neighbourhood of
// (X, Y) is a parallelogram, its bounding box holds <= GATHER_CAP integer points for every "narrow" transform (identity
// 9, scale 0.8 / 30 degrees 16).  Phase 1 (one lane per (input pixel, transform)): walk the box, keep the candidates with a
// non-zero mask whose taps (evaluated by the SAME make_taps as the forward pass) hit (X, Y) in an LDS list (output pixel,
// mask x bilinear weight).  Phase 2 (lanes = 4 channels of a pixel): run the lists, add g where the forward arg-max
// selected that transform.  No atomics, deterministic, one 16-byte store per 4 elements, no zero-fill of the destination.
// "Wide" transforms (a strongly shrinking limb fit; rare) take the scatter kernel below with float atomics.
constexpr int GATHER_MAXDIM = 1024;                                  // per-image coordinate tables of the gather kernel
constexpr int GATHER_OVF_MAX = 65536;                                // tiles the fast kernel can hand to the full-capacity kernel
static __device__ int g_gather_ovf[1 + GATHER_OVF_MAX];               // [0] = count, then (sample << 20 | tile)
constexpr int GATHER_FLAT = 48;                                      // list capacity per input pixel (see the kernel)
constexpr int GATHER_CAP = 16, GATHER_PIX = 32, GATHER_T = 10;      // T <= 10 on the gather path (10 limb transforms / 1)

struct WarpInv { float jx, jy, ix_, iy_, cx, cy, ej, ei; int narrow; };

__device__ __forceinline__ WarpInv invert_warp(const Theta& th, int h, int w, int align) {
  float x00, y00, x10, y10, x01, y01;
  make_coords(th, 0, 0, h, w, align, x00, y00);
  make_coords(th, 0, 1, h, w, align, x10, y10);        // j + 1
  make_coords(th, 1, 0, h, w, align, x01, y01);        // i + 1
  const float a = x10 - x00, b = x01 - x00, c = y10 - y00, d = y01 - y00;     // (ix, iy) = [a b; c d] (j, i) + (x00, y00)
  const float det = a * d - b * c;
  WarpInv r;
  r.narrow = 0;
  r.cx = x00; r.cy = y00;
  r.jx = r.jy = r.ix_ = r.iy_ = r.ej = r.ei = 0.f;
  if (!(fabsf(det) > 1e-12f) || !isfinite(det) || !isfinite(x00) || !isfinite(y00)) return r;
  r.jx = d / det; r.jy = -b / det;                      // j = jx (x - cx) + jy (y - cy)
  r.ix_ = -c / det; r.iy_ = a / det;                    // i = ix_ (x - cx) + iy_ (y - cy)
  r.ej = fabsf(r.jx) + fabsf(r.jy) + 0.02f;             // half extents of the pre-image of [X-1, X+1] x [Y-1, Y+1]
  r.ei = fabsf(r.ix_) + fabsf(r.iy_) + 0.02f;
  const float nj = floorf(2.f * r.ej) + 1.f, ni = floorf(2.f * r.ei) + 1.f;
  r.narrow = (nj * ni <= (float)GATHER_CAP) ? 1 : 0;
  return r;
}

// Round-2 forward.  PMC / ISA of the kernel above: it is VALU-bound, not memory-bound — every one of the C/4 lanes of a pixel
// recomputes the transform's sampling coordinates (four IEEE divisions, floor, clamps: ~90 instructions) for every transform
// whose mask is non-zero, next to 64 multiply/adds of actual sampling (0.23 of the HBM rate at batch 32).  Here a workgroup
// owns 64 consecutive pixels: a pre-pass computes the taps of every (pixel, transform) pair ONCE (64 T pairs over 256
// threads; the column / row terms of the normalised grid come from per-image tables, so the pre-pass has no division) into
// LDS — four clamped byte offsets, four weights (zeroed outside the image) and the mask value — and the sampling pass reads
// them back: per (lane, active transform) two ds_read_b128 + one b32, four unconditional float4 loads, 64 multiply/adds.
// Arithmetic per output element is unchanged (same operations in the same order, adding a +-0 for a zero-weight tap).
struct WarpTap { int o[4]; float w[4]; };          // byte offsets into the sample's feature map (x C x 4 applied), weights

template <bool IB, bool OB, int V = 4>
__global__ __launch_bounds__(256) void warp_fwd3_kernel(const void* feat, const float* aff, const float* warps,
                                                        const float* masks, int T, int C, int h, int w, int H0, int W0,
                                                        int align, int TP, void* out, uint8_t* amax, int relu_out) {
  // IB / OB: the feature map / the output are bf16 tensors (bf16 STORAGE); relu_out: store max(out, 0) — the decoder reads
  // the warped skip only through its ReLU, and relu(x) > 0 <=> x > 0 keeps the backward's activation derivative
  constexpr int ESI = IB ? 2 : 4, ESO = OB ? 2 : 4;
#ifdef PG_TIMING_EXPERIMENTS
  const int fdbg = relu_out >> 8;    // PG_DEBUG_WARP_FWD (timing build only; results are wrong): 1 = no sampling pass, 2 = no pre-pass work
  relu_out &= 0xff;
#else
  constexpr int fdbg = 0;
#endif
  // TP = pixels per tile (host: 64, fewer on small maps so that the grid still fills the chip)
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  Theta* th = reinterpret_cast<Theta*>(wsm);                   // [MAXT]
  float* xs_t = reinterpret_cast<float*>(wsm + MAXT * sizeof(Theta));       // [w] normalised column coordinate
  float* ys_t = xs_t + w;                                                   // [h]
  float* mval = ys_t + h;                                                   // [TP * T]
  WarpTap* taps = reinterpret_cast<WarpTap*>(wsm + ((MAXT * sizeof(Theta) + (size_t)(w + h + TP * T) * 4 + 15) / 16) * 16);
  const int n = blockIdx.y, tid = threadIdx.x;
  if (tid < T) th[tid] = make_theta(warps + ((long)n * T + tid) * 8, h, w, H0, W0);
  {
#pragma clang fp contract(off)
    const float fh = (float)h, fw = (float)w;
    for (int j = tid; j < w; j += 256)
      xs_t[j] = align ? (w > 1 ? (((float)j * 2.0f) / (fw - 1.0f)) - 1.0f : 0.0f) : ((((float)j * 2.0f) + 1.0f) / fw) - 1.0f;
    for (int i = tid; i < h; i += 256)
      ys_t[i] = align ? (h > 1 ? (((float)i * 2.0f) / (fh - 1.0f)) - 1.0f : 0.0f) : ((((float)i * 2.0f) + 1.0f) / fh) - 1.0f;
  }
  // V channels per lane (round 3: 8 on bf16 tensors — 16-byte accesses, half the lanes / instructions per pixel)
  const int cpp = C / V;
  const int ppp = 256 / cpp;                                   // pixels per pass (host: cpp divides 256)
  const float a = aff ? aff[2 * n] : 1.f, b = aff ? aff[2 * n + 1] : 0.f;
  const char* fb = reinterpret_cast<const char*>(feat) + (size_t)n * h * w * C * ESI;
  const int npix = h * w;
  const int cc = (tid % cpp) * V, lp = tid / cpp;
  for (int p0 = blockIdx.x * TP; p0 < npix; p0 += gridDim.x * TP) {
    __syncthreads();
    // ---- pre-pass: (pixel, transform) pairs, mask values read as one contiguous run
    const float* mrow = masks + ((long)n * npix + p0) * T;
    for (int q = tid; q < TP * T; q += 256) {
      const int pl = q / T, t = q - pl * T;
      const int pix = p0 + pl;
      float m = 0.f;
      if (pix < npix) m = mrow[q];
      if (fdbg & 2) m = 0.f;
      mval[q] = m;
      if (m != 0.f) {
#pragma clang fp contract(off)
        const int i = pix / w, j = pix - i * w;
        const Theta tt = th[t];
        const float fh = (float)h, fw = (float)w;
        const float xs = xs_t[j], ys = ys_t[i];
        const float gx = ((tt.t00 * xs) + (tt.t01 * ys)) + tt.t02;
        const float gy = ((tt.t10 * xs) + (tt.t11 * ys)) + tt.t12;
        float ix, iy;
        if (align) { ix = ((gx + 1.0f) / 2.0f) * (fw - 1.0f); iy = ((gy + 1.0f) / 2.0f) * (fh - 1.0f); }
        else { ix = (((gx + 1.0f) * fw) - 1.0f) / 2.0f; iy = (((gy + 1.0f) * fh) - 1.0f) / 2.0f; }
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float fx = ix - x0f, fy = iy - y0f;
        const int x0 = (int)fminf(fmaxf(x0f, -4.0f), (float)w + 4.0f);
        const int y0 = (int)fminf(fmaxf(y0f, -4.0f), (float)h + 4.0f);
        const float wg[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
        WarpTap tp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
          const bool ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h);
          tp.o[k] = (min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)) * C * ESI;
          tp.w[k] = ok ? wg[k] : 0.f;
        }
        taps[q] = tp;
      }
    }
    __syncthreads();
    // ---- sampling pass: lane = (pixel of the pass, V channels)
    if (fdbg & 1) continue;
    for (int pl = lp; pl < TP; pl += ppp) {
      const int pix = p0 + pl;
      if (pix >= npix) break;
      float best[V];
      int bi[V];
#pragma unroll
      for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 255; }
      // A transform whose mask is 0 here contributes the candidate (0, "no transform").  Only the FIRST such transform can
      // change (best, arg): afterwards best >= 0 and `0 > best` stays false — so the later ones cost one LDS read and a branch
      // instead of 3 V instructions each (PMC, round 3: the kernel is instruction-bound, 1.3 issue-active SIMD cycles per
      // cycle; ~9 of the 10 transforms are masked out at a typical pixel).
      bool zero_done = false;
      for (int t = 0; t < T; ++t) {
        const float m = mval[pl * T + t];
        if (m != 0.f) {
#pragma clang fp contract(off)
          const WarpTap tp = taps[pl * T + t];
          float v[4][V];
#pragma unroll
          for (int k = 0; k < 4; ++k) wldv<IB, V>(fb, (size_t)(unsigned)tp.o[k] + cc * ESI, v[k]);
          float s[V];
#pragma unroll
          for (int e = 0; e < V; ++e) s[e] = 0.f;
          if constexpr (IB && OB) {
            // bf16 STORAGE: the result is rounded to bf16 anyway, so the deferred affine is applied once per pixel instead of
            // once per tap (sum_k ((v_k a + b) w_k) = a sum_k v_k w_k + b sum_k w_k) and the taps are FMAs: 1 instruction per
            // (tap, channel) instead of 4 — the kernel is instruction-bound.  The fp32 instantiations keep the reference's order.
            const float wsum = (tp.w[0] + tp.w[1]) + (tp.w[2] + tp.w[3]);
            const float am = a * m, bm = b * wsum * m;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int e = 0; e < V; ++e) s[e] = __builtin_fmaf(v[k][e], tp.w[k], s[e]);
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float cand = __builtin_fmaf(s[e], am, bm);
              if (cand > best[e]) { best[e] = cand; bi[e] = t; }
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int e = 0; e < V; ++e) s[e] = s[e] + ((v[k][e] * a) + b) * tp.w[k];
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float cand = s[e] * m;
              if (cand > best[e]) { best[e] = cand; bi[e] = t; }
            }
          }
        } else if (!zero_done) {
          zero_done = true;
#pragma unroll
          for (int e = 0; e < V; ++e)
            if (0.f > best[e]) { best[e] = 0.f; bi[e] = 255; }
        }
      }
      const long o = ((long)n * npix + pix) * C + cc;
      if (relu_out) {
#pragma unroll
        for (int e = 0; e < V; ++e) best[e] = fmaxf(best[e], 0.f);
      }
      wstv<OB, V>(reinterpret_cast<char*>(out), (size_t)o * ESO, best);
      if (amax) {
        unsigned pk[V / 4];
#pragma unroll
        for (int q = 0; q < V / 4; ++q)
          pk[q] = (unsigned)(bi[4 * q] & 0xff) | ((unsigned)(bi[4 * q + 1] & 0xff) << 8) | ((unsigned)(bi[4 * q + 2] & 0xff) << 16) |
                  ((unsigned)(bi[4 * q + 3] & 0xff) << 24);
        if constexpr (V == 8) *reinterpret_cast<uint2*>(amax + o) = make_uint2(pk[0], pk[1]);      // one 8-byte store per lane
        else *reinterpret_cast<unsigned*>(amax + o) = pk[0];
      }
    }
  }
}

// Round-5 forward for bf16 STORAGE (8 channels per lane).  What bounds the kernel above (level 0, batch 32: 250 us = 2.7 TB/s over
// the algorithmic bytes) — found by elimination: its stores and mask reads alone, in a plain streaming kernel, take
// 80 - 95 us (tools/micro/store_patterns.hip); a first barrier-free version of this kernel with ~25 % more VALU work per pixel
// ran in the SAME 250 us, and trimming that work by 15 % moved nothing — so neither bytes, nor barriers, nor instruction issue.
// It is the chain of DEPENDENT memory round trips per wave and tile: mask values -> (taps ->) gathers of the first active
// transform -> gathers of the second (95 % of the waves hold a pixel with two) -> stores, whose completion the NEXT tile's
// first wait has to sit out as well, because gfx950 counts loads and stores in one in-order vmcnt.  At 5 waves per SIMD and
// 1 - 2 us per round trip under load that is the measured time.  This version shortens the chain instead of widening it:
//   * no LDS tap table and no per-tile barrier: the C/8 lanes of a pixel fetch one mask value each (one coalesced 4-byte load
//     per lane and round; __ballot turns them into the pixel's set of active transforms), every lane walks ITS pixel's active
//     set and evaluates the taps itself (~40 VALU instructions per active transform next to ~100 of sampling);
//   * the next tile's mask values are fetched under this tile's gathers (- 11 % on their own);
//   * the tile's results stay in registers and are stored AFTER the next tile's first gathers have been issued: the wait for
//     those gathers is then vmcnt(<number of stores>), and no wait in the loop stands behind a fresh store.
// Candidate order, arithmetic and the position of the "no transform" candidate are those of warp_fwd3_kernel: results are
// BIT-equal to it (tests/test_gpu_round5.py::test_warp_forward_v5_equals_v3).
// Host guarantees: C / 8 a power of two in 8 ... 64, T <= 2 C / 8, h w C / 8 a multiple of 64 (every lane of every wave owns a pixel).
template <bool HAS_AMAX>
__global__ __launch_bounds__(256) void warp_fwd5_kernel(const void* feat, const float* aff, const float* warps, const float* masks,
                                                        int T, int C, int h, int w, int H0, int W0, int align, void* out,
                                                        uint8_t* amax, int relu_out) {
  constexpr int V = 8;
  typedef float f2 __attribute__((ext_vector_type(2)));        // v_pk_fma_f32: two channels per instruction
  __shared__ Theta th[MAXT];
  extern __shared__ __attribute__((aligned(16))) char wsm[];
  float* xs_t = reinterpret_cast<float*>(wsm);                 // [w] normalised column coordinate, [h] row coordinate: the two
  float* ys_t = xs_t + w;                                      // IEEE divisions per pixel happen once per workgroup
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  if (tid < T) th[tid] = make_theta(warps + ((long)n * T + tid) * 8, h, w, H0, W0);
  const float fh = (float)h, fw = (float)w;
  {
#pragma clang fp contract(off)
    for (int j = tid; j < w; j += 256)
      xs_t[j] = align ? (w > 1 ? (((float)j * 2.0f) / (fw - 1.0f)) - 1.0f : 0.0f) : ((((float)j * 2.0f) + 1.0f) / fw) - 1.0f;
    for (int i = tid; i < h; i += 256)
      ys_t[i] = align ? (h > 1 ? (((float)i * 2.0f) / (fh - 1.0f)) - 1.0f : 0.0f) : ((((float)i * 2.0f) + 1.0f) / fh) - 1.0f;
  }
  __syncthreads();
  const int wsh = (w & (w - 1)) == 0 ? 31 - __builtin_clz(w) : -1;
  const int cpp = C / V;                                       // lanes per pixel
  const int cshift = 31 - __builtin_clz(cpp);
  const int npix = h * w;
  const int items = npix * cpp;                                // host: < 2^31, a multiple of 64
  const float a = aff ? aff[2 * n] : 1.f, b = aff ? aff[2 * n + 1] : 0.f;
  const char* fb = reinterpret_cast<const char*>(feat) + (size_t)n * npix * C * 2;
  char* ob = reinterpret_cast<char*>(out) + (size_t)n * npix * C * 2;
  uint8_t* ab = HAS_AMAX ? amax + (size_t)n * npix * C : nullptr;
  const int sub = lane & (cpp - 1), grp = lane - sub;          // my index in the pixel's lane group, the group's first lane
  const unsigned long long gmask = cpp >= 64 ? ~0ull : ((1ull << cpp) - 1ull);
  const unsigned tmask = T >= 32 ? 0xffffffffu : ((1u << T) - 1u);
  const int cc = sub * V;
  const float* mbase = masks + (long)n * npix * T;
  const int stride = gridDim.x * 256;
  // one mask value per lane and round (T <= 2 x lanes per pixel: two rounds, both loads issued before the first wait)
  auto load_masks = [&](int base_, float (&mv_)[2]) {
    const int pix_ = (base_ + lane) >> cshift;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int t = r * cpp + sub;
      mv_[r] = (t < T) ? mbase[(long)pix_ * T + t] : 0.f;
    }
  };
  // taps of transform t at normalised coordinates (xs, ys): four clamped byte offsets, four weights (zero outside the image)
  auto make_taps4 = [&](int t, float xs, float ys, int (&o4)[4], float (&w4)[4]) {
#pragma clang fp contract(off)
    const Theta tt = th[t];
    const float gx = ((tt.t00 * xs) + (tt.t01 * ys)) + tt.t02;
    const float gy = ((tt.t10 * xs) + (tt.t11 * ys)) + tt.t12;
    float ix, iy;
    if (align) { ix = ((gx + 1.0f) / 2.0f) * (fw - 1.0f); iy = ((gy + 1.0f) / 2.0f) * (fh - 1.0f); }
    else { ix = (((gx + 1.0f) * fw) - 1.0f) / 2.0f; iy = (((gy + 1.0f) * fh) - 1.0f) / 2.0f; }
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float fx = ix - x0f, fy = iy - y0f;
    const int x0 = (int)fminf(fmaxf(x0f, -4.0f), (float)w + 4.0f);
    const int y0 = (int)fminf(fmaxf(y0f, -4.0f), (float)h + 4.0f);
    const float wg[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
      const bool ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h);
      o4[k] = (min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)) * C * 2 + cc * 2;
      w4[k] = ok ? wg[k] : 0.f;
    }
  };
  float best[V];
  unsigned bip[2];                                             // arg-max bytes, packed as they are stored
  auto zero_candidate = [&]() {
#pragma unroll
    for (int e = 0; e < V; ++e)
      if (0.f > best[e]) { best[e] = 0.f; bip[e >> 2] |= 0xffu << (8 * (e & 3)); }
  };
  auto consume = [&](const uint4 (&raw)[4], const float (&w4)[4], float m, int t) {
    f2 s[V / 2];
#pragma unroll
    for (int e = 0; e < V / 2; ++e) s[e] = (f2){0.f, 0.f};
    float wsum, am, bm;
    {
#pragma clang fp contract(off)
      wsum = (w4[0] + w4[1]) + (w4[2] + w4[3]);
      am = a * m;
      bm = b * wsum * m;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned u[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
#pragma unroll
      for (int e = 0; e < V / 2; ++e)
        s[e] = __builtin_elementwise_fma((f2){__uint_as_float(u[e] << 16), __uint_as_float(u[e] & 0xffff0000u)}, (f2){w4[k], w4[k]}, s[e]);
    }
    const unsigned tsplat = (unsigned)t * 0x01010101u;
#pragma unroll
    for (int e = 0; e < V / 2; ++e) {
      const f2 cand = __builtin_elementwise_fma(s[e], (f2){am, am}, (f2){bm, bm});
      const float cv[2] = {cand.x, cand.y};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ch = 2 * e + q;
        const bool gt = cv[q] > best[ch];
        best[ch] = gt ? cv[q] : best[ch];
        const unsigned sel = gt ? 0xffu << (8 * (ch & 3)) : 0u;
        bip[ch >> 2] = (bip[ch >> 2] & ~sel) | (tsplat & sel);
      }
    }
  };
  // results of the previous tile, stored under this tile's first gathers.  Before the first tile: zeros to this lane's OWN first
  // destination (its real values follow from the same lane in program order) — the stores stay unconditional, so the compiler's
  // vmcnt for the gathers is a constant
  uint4 pend_o = make_uint4(0u, 0u, 0u, 0u);
  uint2 pend_a = make_uint2(0u, 0u);
  const int base0 = blockIdx.x * 256 + (tid - lane);           // per WAVE: items is a multiple of 64, not of 256
  int pend_off = ((base0 + lane) >> cshift) * C + cc;
  float mv[2];
  if (base0 < items) load_masks(base0, mv);
  for (int base = base0; base < items; base += stride) {
    const int pix = (base + lane) >> cshift;
    const float* mrow = mbase + (long)pix * T;
    // ---- the pixel's active transforms: bit t <=> mask != 0
    unsigned act = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const unsigned long long bal = __ballot(mv[r] != 0.f);
      if (r * cpp < 32) act |= (unsigned)((bal >> grp) & gmask) << (r * cpp);
    }
    act &= tmask;
    const unsigned inact = ~act & tmask;
    const int z = inact ? __builtin_ctz(inact) : 64;           // the first masked-out transform: where the candidate (0, none) enters
    if (base + stride < items) load_masks(base + stride, mv);  // the NEXT tile's mask values, under this tile's gathers
    const int i = wsh >= 0 ? pix >> wsh : pix / w, j = pix - i * w;
    const float xs = xs_t[j], ys = ys_t[i];
#pragma unroll
    for (int e = 0; e < V; ++e) best[e] = -INFINITY;
    bip[0] = bip[1] = 0xffffffffu;
    bool zero_done = false;
    unsigned bits = act;
    // ---- first active transform: gathers issued, THEN the previous tile's stores, then the wait
    const bool has = bits != 0;
    uint4 raw[4];
    float w4[4], m = 0.f;
    int t = 0;
    if (has) {
      t = __builtin_ctz(bits);
      bits &= bits - 1;
      if (t > z) { zero_done = true; zero_candidate(); }
      int o4[4];
      make_taps4(t, xs, ys, o4, w4);
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[k] = *reinterpret_cast<const uint4*>(fb + (unsigned)o4[k]);
      m = mrow[t];
    }
    *reinterpret_cast<uint4*>(ob + (size_t)(unsigned)pend_off * 2) = pend_o;
    if (HAS_AMAX) *reinterpret_cast<uint2*>(ab + (unsigned)pend_off) = pend_a;
    if (has) consume(raw, w4, m, t);
    while (bits) {
      t = __builtin_ctz(bits);
      bits &= bits - 1;
      if (!zero_done && t > z) { zero_done = true; zero_candidate(); }
      int o4[4];
      make_taps4(t, xs, ys, o4, w4);
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[k] = *reinterpret_cast<const uint4*>(fb + (unsigned)o4[k]);
      m = mrow[t];
      consume(raw, w4, m, t);
    }
    if (!zero_done && z < T) zero_candidate();
    if (relu_out) {
#pragma unroll
      for (int e = 0; e < V; ++e) best[e] = fmaxf(best[e], 0.f);
    }
    pend_o = make_uint4(wpack_bf16(best[0], best[1]), wpack_bf16(best[2], best[3]), wpack_bf16(best[4], best[5]), wpack_bf16(best[6], best[7]));
    pend_a = make_uint2(bip[0], bip[1]);
    pend_off = pix * C + cc;
  }
  if (base0 < items) {
    *reinterpret_cast<uint4*>(ob + (size_t)(unsigned)pend_off * 2) = pend_o;
    if (HAS_AMAX) *reinterpret_cast<uint2*>(ab + (unsigned)pend_off) = pend_a;
  }
}

// LIST = false (the launch over all tiles): per-pixel candidate lists of GATHER_FLAT = 48 entries — 12 KB per workgroup instead of
// the 40 KB of the worst case 10 x 16 (PMC: 2.4 waves per SIMD, 65 % of the wave cycles waiting; 1.06 -> 0.83 ms per pass at
// batch 32).  A tile in which a pixel's list would overflow (several masks overlapping at a strongly minified spot) is not written
// but appended to g_gather_ovf; LIST = true (a small second launch with the worst-case capacity) walks that list.
template <bool GB, bool DB, int V = 4, bool LIST = false>
__global__ __launch_bounds__(256) void warp_bwd_gather_kernel(const void* gout, const uint8_t* amax, const float* warps,
                                                              const float* masks, int T, int C, int h, int w, int H0, int W0,
                                                              int align, void* dfeat, const int* bbox, int det) {
  // GB / DB: the incoming gradient / the written input gradient are bf16 tensors (bf16 STORAGE)
  // det (PG_DETERMINISTIC): every pixel's list is sorted by (output pixel, transform) before it is summed — the append order is
  // the arrival order of LDS atomics, i.e. the fp32 summation order would differ from run to run in the last bits
  constexpr int ESG = GB ? 2 : 4, ESD = DB ? 2 : 4;
  constexpr int FLAT = LIST ? GATHER_T * GATHER_CAP : GATHER_FLAT;
#ifndef PG_GATHER_PB
#define PG_GATHER_PB 2
#endif
  constexpr int PB = PG_GATHER_PB;          // list entries per batch of the gather phase (independent loads in flight).
                                            // Round 5, last day: what this kernel needs is RESIDENT WAVES, not loads per wave — it is bound by dependent memory round
                                            // trips like the forward.  4 entries per batch = 100 VGPRs = 4 waves per SIMD: 712 us per pass at batch 32; 3 = 84 VGPRs = 5
                                            // waves: 667; 2 with 32-bit offsets = 80 VGPRs and the coordinate tables in dynamic LDS (22 KB per workgroup): 6 waves, 597;
                                            // + two candidate items per lane and round instead of four (68 VGPRs, 7 waves): 571.  (8 entries: 155 VGPRs, 924 us.)
#ifdef PG_TIMING_EXPERIMENTS
  const int wdbg = align >> 8;       // PG_DEBUG_WARP_BWD (timing build only; results are wrong): 1 = no phase 2, 2 = no phase 1
  align &= 0xff;
#else
  constexpr int wdbg = 0;
#endif
  __shared__ Theta th[MAXT];
  __shared__ WarpInv inv[MAXT];
  __shared__ int e_pix[GATHER_PIX][FLAT];                 // output pixel | transform << 24
  __shared__ float e_w[GATHER_PIX][FLAT];                 // mask x bilinear weight
  __shared__ int e_cnt[GATHER_PIX];
  __shared__ int e_ovf;
  __shared__ int pr_n;                                    // accepted (pixel, transform) pairs of the tile:
  __shared__ int4 pr[GATHER_PIX * GATHER_T];              //   pixel | t << 8, i0, j0 | nj << 16, candidates (<= GATHER_CAP)
  __shared__ int px_x[GATHER_PIX], px_y[GATHER_PIX];
  extern __shared__ __attribute__((aligned(16))) char gsm[];       // normalised grid coordinate per column / row: (w + h) floats
  float* const xs_t = reinterpret_cast<float*>(gsm);               // (dynamic: 2 KB at 256^2 instead of 8 KB static — a sixth workgroup per CU)
  float* const ys_t = xs_t + w;
  const int nent = LIST ? min(g_gather_ovf[0], GATHER_OVF_MAX) : 1;
  for (int ent = LIST ? (int)blockIdx.x : 0; ent < nent; ent += LIST ? (int)gridDim.x : 1) {
  const int n = LIST ? (g_gather_ovf[1 + ent] >> 20) : (int)blockIdx.y;
  if (LIST) __syncthreads();                                // the previous entry's tables are consumed
  // bbox (optional, pg_mask_bbox): per (sample, transform) the bounding box of the non-zero FULL-resolution mask.  At this level a
  // mask pixel can be non-zero only inside the box scaled to the level and grown by the bilinear down-sampling's reach; a
  // (pixel, transform) pair whose candidate box misses it is skipped before any mask load — the limb masks cover a few per cent
  // of the image, so phase 1 (16 strided mask loads + tap evaluation per pair) runs for ~2 of the 10 transforms.
  __shared__ int mb[MAXT][4];
  if (threadIdx.x < T) {
    th[threadIdx.x] = make_theta(warps + ((long)n * T + threadIdx.x) * 8, h, w, H0, W0);
    inv[threadIdx.x] = invert_warp(th[threadIdx.x], h, w, align);
    int y0 = 0, y1 = h - 1, x0 = 0, x1 = w - 1;
    if (bbox != nullptr) {
      const int* b = bbox + ((long)n * T + threadIdx.x) * 4;       // ymin, ymax, xmin, xmax at (H0, W0); empty: ymax < ymin
      const float sy = (float)h / (float)H0, sx = (float)w / (float)W0;
      y0 = (int)floorf((float)(b[0] - 1) * sy) - 2; y1 = (int)ceilf((float)(b[1] + 1) * sy) + 2;
      x0 = (int)floorf((float)(b[2] - 1) * sx) - 2; x1 = (int)ceilf((float)(b[3] + 1) * sx) + 2;
      if (b[1] < b[0] || b[3] < b[2]) { y0 = h + 8; y1 = -8; x0 = w + 8; x1 = -8; }
    }
    mb[threadIdx.x][0] = y0; mb[threadIdx.x][1] = y1; mb[threadIdx.x][2] = x0; mb[threadIdx.x][3] = x1;
  }
  for (int k = threadIdx.x; k < w; k += 256) xs_t[k] = warp_norm_coord(k, w, align);
  for (int k = threadIdx.x; k < h; k += 256) ys_t[k] = warp_norm_coord(k, h, align);
  const long nb = (long)n * h * w;
  const uint8_t* const amax_n = amax + nb * C;
  const char* const gout_n = reinterpret_cast<const char*>(gout) + (size_t)(nb * C) * ESG;
  // persistent workgroups (round 3): the per-sample set-up above (ten inverted transforms, w + h table entries) cost more
  // than the 32 pixels of work behind it when every tile was its own workgroup (65536 workgroups at 256^2, batch 32)
  const int ntiles = LIST ? (g_gather_ovf[1 + ent] & 0xfffff) + 1 : (h * w + GATHER_PIX - 1) / GATHER_PIX;
  for (int tile = LIST ? ntiles - 1 : (int)blockIdx.x; tile < ntiles; tile += LIST ? 1 : (int)gridDim.x) {
  __syncthreads();                                           // the previous tile's lists are consumed
  if (threadIdx.x < GATHER_PIX) e_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) e_ovf = 0;
  __syncthreads();
  const int P0 = tile * GATHER_PIX;
  // ---- phase 1 (round 4: two stages).  One lane per (input pixel, transform) evaluating up to 16 candidates serially kept ~2 of
  // 10 lanes busy (the limb masks' bounding boxes reject the rest) with 80 registers of per-lane candidate arrays: 300 of the
  // 480 us of the level-0 launch at batch 32 (timing build, PG_DEBUG_WARP_BWD).  Now: (a) every pair computes its pre-image box
  // and the rejections — transform not narrow, box outside the map, box outside the mask's bounding box — and the accepted pairs
  // are compacted into `pr`; (b) one lane per (accepted pair, candidate): mask value, bilinear footprint test, list append.
  // The order of a pixel's list entries was already decided by LDS atomics across transforms; it still is.
  if (threadIdx.x == 0) pr_n = 0;
  if (threadIdx.x < GATHER_PIX) {
    const int P = P0 + (int)threadIdx.x;
    const int Y = P / w;
    px_y[threadIdx.x] = Y; px_x[threadIdx.x] = P - Y * w;
  }
  __syncthreads();
  if (!(wdbg & 2)) {
    const int p = threadIdx.x & (GATHER_PIX - 1), P = P0 + p;
    const int X = px_x[p], Y = px_y[p];
    for (int t = threadIdx.x / GATHER_PIX; t < T; t += 256 / GATHER_PIX) {
      const WarpInv v = inv[t];
      if (!v.narrow || P >= h * w) continue;
      const float dx = (float)X - v.cx, dy = (float)Y - v.cy;
      const float jc = v.jx * dx + v.jy * dy, ic = v.ix_ * dx + v.iy_ * dy;
      if (!(jc + v.ej >= 0.f && jc - v.ej <= (float)w && ic + v.ei >= 0.f && ic - v.ei <= (float)h)) continue;
      const int j0 = max((int)ceilf(jc - v.ej), 0), j1 = min((int)floorf(jc + v.ej), w - 1);
      const int i0 = max((int)ceilf(ic - v.ei), 0), i1 = min((int)floorf(ic + v.ei), h - 1);
      const int nj = j1 - j0 + 1, tot = nj * (i1 - i0 + 1);
      if (nj <= 0 || tot <= 0) continue;
      if (i1 < mb[t][0] || i0 > mb[t][1] || j1 < mb[t][2] || j0 > mb[t][3]) continue;      // every candidate has a zero mask
      const int k = atomicAdd(&pr_n, 1);
      pr[k] = make_int4(p | (t << 8), i0, j0 | (nj << 16), min(tot, GATHER_CAP));
    }
  }
  __syncthreads();
  {
    // (tried: one word per candidate written by stage (a) so that stage (b) is dense — the divergent store loop costs more than
    //  the idle lanes: level 0 at batch 32 372 -> 408 us.  Timing build at that level: stage (a) 53 us, stage (b) 95, gather 178.)
    const int nitems = (wdbg & 4) ? 0 : pr_n * GATHER_CAP;      // (timing build: 4 = no candidate stage)
    constexpr int U = 2;                                   // items per lane and round: the mask loads go out as one batch
    for (int it0 = threadIdx.x; it0 < nitems; it0 += 256 * U) {
      float mv[U];
      int pp[U], tt[U], ci[U], cj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = it0 + u * 256;
        const bool in = it < nitems;
        const int4 d = pr[in ? (it >> 4) : 0];
        const int e = it & (GATHER_CAP - 1), nj = d.z >> 16;
        const bool val = in && e < d.w;
        const int di = (int)(((float)e + 0.5f) * __frcp_rn((float)nj));       // e / nj, e < 16 (exact)
        pp[u] = d.x & 0xff; tt[u] = d.x >> 8;
        ci[u] = d.y + di; cj[u] = (d.z & 0xffff) + e - di * nj;
        mv[u] = val ? masks[(nb + (long)ci[u] * w + cj[u]) * T + tt[u]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (mv[u] == 0.f) continue;
        const Taps tp = make_taps_xy(th[tt[u]], xs_t[cj[u]], ys_t[ci[u]], h, w, align);      // division-free (tables)
        const int kx = px_x[pp[u]] - tp.x0, ky = px_y[pp[u]] - tp.y0;
        if ((unsigned)kx <= 1u && (unsigned)ky <= 1u) {
          const float wk = ky ? (kx ? tp.w11 : tp.w10) : (kx ? tp.w01 : tp.w00);
          const int at = atomicAdd(&e_cnt[pp[u]], 1);
          if (at < FLAT) { e_pix[pp[u]][at] = (ci[u] * w + cj[u]) | (tt[u] << 24); e_w[pp[u]][at] = mv[u] * wk; }
          else e_ovf = 1;                                  // (LIST: FLAT = T x CAP, cannot happen)
        }
      }
    }
  }
  __syncthreads();
  const int cq = C / V;
  if (!LIST && e_ovf) {       // rare: hand the tile to the full-capacity launch (host: N * tiles <= GATHER_OVF_MAX)
    if (threadIdx.x == 0) {
      const int slot = atomicAdd(&g_gather_ovf[0], 1);
      if (slot < GATHER_OVF_MAX) g_gather_ovf[1 + slot] = (n << 20) | tile;
    }
    continue;
  }
  if (det) {
    if (threadIdx.x < GATHER_PIX) {
      const int p = threadIdx.x, cnt = min(e_cnt[p], FLAT);
      for (int a = 1; a < cnt; ++a) {             // insertion sort (<= 48 entries, usually < 8)
        const int key = e_pix[p][a];
        const float wv = e_w[p][a];
        int b = a - 1;
        while (b >= 0 && e_pix[p][b] > key) { e_pix[p][b + 1] = e_pix[p][b]; e_w[p][b + 1] = e_w[p][b]; --b; }
        e_pix[p][b + 1] = key; e_w[p][b + 1] = wv;
      }
    }
    __syncthreads();
  }
  // ---- phase 2: lanes = V channels of an input pixel; entries in batches of four (independent loads in flight)
  if (wdbg & 1) continue;
  for (int q = threadIdx.x; q < GATHER_PIX * cq; q += 256) {
    const int p = q / cq, c4 = (q - p * cq) * V;
    const int P = P0 + p;
    if (P >= h * w) continue;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    const int cnt = e_cnt[p];
    for (int e0 = 0; e0 < cnt; e0 += PB) {
      unsigned o[PB]; float wt[PB]; int tt[PB]; unsigned am[PB][V / 4]; float g[PB][V];      // o: element offset inside the sample (host: h w C < 2^31)
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const bool val = e0 + u < cnt;
        const int ee = val ? e0 + u : e0;
        const int pk = e_pix[p][ee];
        tt[u] = val ? (pk >> 24) : 256;          // padding entries match no arg-max byte (255 = "no transform won" is a byte value)
        wt[u] = val ? e_w[p][ee] : 0.f;
        o[u] = (unsigned)(pk & 0xffffff) * (unsigned)C + (unsigned)c4;
#pragma unroll
        for (int qd = 0; qd < V / 4; ++qd) am[u][qd] = *reinterpret_cast<const unsigned*>(amax_n + o[u] + 4 * qd);
        wldv<GB, V>(gout_n, (size_t)o[u] * ESG, g[u]);
      }
#pragma unroll
      for (int u = 0; u < PB; ++u)
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const int ab = (int)((am[u][e >> 2] >> (8 * (e & 3))) & 0xffu);
          acc[e] += ab == tt[u] ? g[u][e] * wt[u] : 0.f;
        }
    }
    wstv<DB, V>(reinterpret_cast<char*>(dfeat), (size_t)((nb + P) * C + c4) * ESD, acc);
  }
  }   // tile loop
  }   // LIST: entry loop
}

// Scatter form (float atomics), `wide_only`: only the elements whose selected transform is not "narrow" — the complement
// of the gather kernel; workgroups of a sample without wide transforms leave at once.
template <bool GB, bool DB>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const void* gout_, const uint8_t* amax, const float* warps,
                                                       const float* masks, int T, int C, int h, int w, int H0, int W0,
                                                       int align, void* dfeat_, int wide_only) {
  __shared__ Theta th[MAXT];
  __shared__ int wide[MAXT];
  __shared__ int any_wide;
  const int n = blockIdx.y;
  if (threadIdx.x == 0) any_wide = 0;
  __syncthreads();
  if (threadIdx.x < T) {
    th[threadIdx.x] = make_theta(warps + ((long)n * T + threadIdx.x) * 8, h, w, H0, W0);
    wide[threadIdx.x] = wide_only ? !invert_warp(th[threadIdx.x], h, w, align).narrow : 1;
    if (wide[threadIdx.x]) any_wide = 1;
  }
  __syncthreads();
  if (!any_wide) return;
  // One lane per (pixel, channel): every element has exactly ONE selected transform (the forward argmax), so a lane
  // computes the taps of its own transform and issues its four corner atomics once.
  const long items = (long)h * w * C;
  const long nb = (long)n * h * w;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    const int pix = (int)(it / C);
    const int c = (int)(it - (long)pix * C);
    const int i = pix / w, j = pix - i * w;
    const long o = (nb + pix) * C + c;
    const int t = amax[o];
    if (t >= T || !wide[t]) continue;                // 255: no transform won (all masks zero) -> no gradient
    float g;
    if constexpr (GB) g = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(gout_)[o] << 16);
    else g = reinterpret_cast<const float*>(gout_)[o];
    const float m = masks[(nb + pix) * T + t];
    const Taps tp = make_taps(th[t], i, j, h, w, align);
    const float gm = g * m;
    const float wg[4] = {tp.w00, tp.w01, tp.w10, tp.w11};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = tp.x0 + (k & 1), yy = tp.y0 + (k >> 1);
      const float v = gm * wg[k];
      if (v != 0.f && xx >= 0 && xx < w && yy >= 0 && yy < h) {
        const long e = (nb + (long)yy * w + xx) * C + c;
        if constexpr (DB) {
          // packed bf16 atomic: this element's half carries the value, the neighbour's half adds +0
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          const unsigned pk = (c & 1) ? (wpack_bf16(0.f, v)) : (wpack_bf16(v, 0.f));
          auto* addr = reinterpret_cast<__attribute__((address_space(1))) bf16x2*>(
              reinterpret_cast<unsigned long long>(reinterpret_cast<unsigned short*>(dfeat_) + (e & ~1l)));
          __builtin_amdgcn_global_atomic_fadd_v2bf16(addr, __builtin_bit_cast(bf16x2, pk));
        } else {
          atomicAdd(reinterpret_cast<float*>(dfeat_) + e, v);
        }
      }
    }
  }
}


// PG_DETERMINISTIC form of the scatter above (round 6): one lane owns ONE channel of a sample and walks the output
// pixels in order, adding its taps to dfeat with plain read-modify-writes — every destination element has a single writer and a
// fixed summation order, so two runs are bit-equal (the float-atomic form adds in arrival order).  Serial over h x w per lane:
// slow by design, and only reached by the transforms the gather kernel leaves out (shrinking by more than ~0.6) or by shapes
// outside the gather form; workgroups of a sample without such a transform leave at once.
template <bool GB, bool DB>
__global__ __launch_bounds__(64) void warp_bwd_det_kernel(const void* gout_, const uint8_t* amax, const float* warps,
                                                          const float* masks, int T, int C, int h, int w, int H0, int W0,
                                                          int align, void* dfeat_, int wide_only) {
  __shared__ Theta th[MAXT];
  __shared__ int wide[MAXT];
  __shared__ int any_wide;
  const int n = blockIdx.y;
  if (threadIdx.x == 0) any_wide = 0;
  __syncthreads();
  if (threadIdx.x < T) {
    th[threadIdx.x] = make_theta(warps + ((long)n * T + threadIdx.x) * 8, h, w, H0, W0);
    wide[threadIdx.x] = wide_only ? !invert_warp(th[threadIdx.x], h, w, align).narrow : 1;
    if (wide[threadIdx.x]) any_wide = 1;
  }
  __syncthreads();
  if (!any_wide) return;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const long nb = (long)n * h * w;
  for (int pix = 0; pix < h * w; ++pix) {
    const int i = pix / w, j = pix - i * w;
    const long o = (nb + pix) * C + c;
    const int t = amax[o];
    if (t >= T || !wide[t]) continue;
    float g;
    if constexpr (GB) g = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(gout_)[o] << 16);
    else g = reinterpret_cast<const float*>(gout_)[o];
    const float m = masks[(nb + pix) * T + t];
    const Taps tp = make_taps(th[t], i, j, h, w, align);
    const float gm = g * m;
    const float wg[4] = {tp.w00, tp.w01, tp.w10, tp.w11};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = tp.x0 + (k & 1), yy = tp.y0 + (k >> 1);
      const float v = gm * wg[k];
      if (v != 0.f && xx >= 0 && xx < w && yy >= 0 && yy < h) {
        const long e = (nb + (long)yy * w + xx) * C + c;
        if constexpr (DB) {
          unsigned short* q = reinterpret_cast<unsigned short*>(dfeat_) + e;
          const float cur = __uint_as_float((unsigned)(*q) << 16);
          *q = (unsigned short)(wpack_bf16(cur + v, 0.f) & 0xffffu);      // one bf16 rounding per add, as the packed atomic
        } else {
          reinterpret_cast<float*>(dfeat_)[e] += v;
        }
      }
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
  PG_KLAUNCH(cords_to_map_kernel, dim3(bx, P, N), dim3(256), 0, (hipStream_t)stream, cords, P, H, W,
                     1.0 / (2.0 * (double)sigma * (double)sigma), out, (long)oN, (long)oC, (long)oH, (long)oW);
  PG_LAUNCH_OK("pg_cords_to_map");
  return 0;
}


extern "C" int pg_mask_pyramid(const void* masks, int32_t is_f64, int32_t N, int32_t T, int32_t H0, int32_t W0,
                               int32_t h, int32_t w, float* out, void* stream) {
  PG_REQUIRE(masks && out && N > 0 && T > 0 && h > 0 && w > 0, "pg_mask_pyramid: bad arguments");
  PG_REQUIRE(h <= 65535 && N <= 65535, "pg_mask_pyramid: grid limits");
  const dim3 grid((w + 63) / 64, h, N);
  if (is_f64)
    PG_KLAUNCH(mask_pyramid_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const double*)masks, N, T, H0, W0, h, w, out);
  else
    PG_KLAUNCH(mask_pyramid_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
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

// io_flags: bit 0 = `feat` is bf16, bit 1 = `out` is bf16, bit 2 = store relu(out) (bf16 STORAGE on the bf16 data path)
extern "C" int pg_warp_mask_max_fwd_io(const void* feat, const float* aff, const float* warps, const float* lvl_masks,
                                       int32_t N, int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0,
                                       int32_t align_corners, void* out, uint8_t* argmax, int32_t io_flags, void* stream) {
  PG_REQUIRE(feat && warps && lvl_masks && out, "pg_warp_mask_max_fwd: null pointer");
  PG_REQUIRE(T >= 1 && T <= MAXT && C % 4 == 0 && N > 0, "pg_warp_mask_max_fwd: need T<=32, C%%4==0 (T=%d C=%d)", T, C);
  static const bool v1 = getenv("PG_WARP_FWD_V1") != nullptr;      // ablation switch: the round-1 per-lane walk
  const bool ib = io_flags & 1, ob = io_flags & 2;
  static const bool no_v8 = getenv("PG_WARP_NO_V8") != nullptr;      // ablation switch: 4 channels per lane on bf16 tensors too
  const bool v8 = ib && ob && C % 8 == 0 && 256 % (C / 8) == 0 && !no_v8;
  const int cpp = v8 ? C / 8 : C / 4;
  const size_t lds = ((MAXT * sizeof(Theta) + (size_t)(w + h + 64 * T) * 4 + 15) / 16) * 16 + (size_t)64 * T * sizeof(WarpTap);
  int relu = (io_flags & 4) ? 1 : 0;
#ifdef PG_TIMING_EXPERIMENTS
  { static const int fd = getenv("PG_DEBUG_WARP_FWD") ? atoi(getenv("PG_DEBUG_WARP_FWD")) : 0; relu |= fd << 8; }
#endif
  if (v1 || cpp > 256 || 256 % cpp != 0 || lds > 64 * 1024 || (double)h * w * C * 4.0 >= 2147483648.0) {
    PG_REQUIRE(io_flags == 0, "pg_warp_mask_max_fwd: bf16 storage needs the tiled kernel (C / 4 must divide 256)");
    PG_KLAUNCH(warp_fwd_kernel, dim3(warp_grid(C, h, w), N), dim3(256), 0, (hipStream_t)stream, (const float*)feat, aff, warps,
                       lvl_masks, T, C, h, w, H0, W0, align_corners, (float*)out, argmax);
  } else {
    int tp = 64;
    const int ppp = 256 / cpp;
    while (tp > 8 && tp / 2 >= ppp && (((long)h * w + tp - 1) / tp) * N < 1024) tp /= 2;
    long tiles = ((long)h * w + tp - 1) / tp;
    // ~8192 workgroups per launch, each walking several tiles (round 4 sweep at batch 32, level 0: 32768 one-tile workgroups
    // 265 us, 8192: 250, 4096: 257, 2048: 300 — the set-up per workgroup is small, the kernel needs many loads in flight)
    static const long wcap = getenv("PG_WARP_FWD_WGS") ? atol(getenv("PG_WARP_FWD_WGS")) : 8192;
    const long cap = wcap / N > 32 ? wcap / N : 32;
    if (tiles > cap) tiles = cap;
    const dim3 grid((unsigned)tiles, N);
    hipStream_t st = (hipStream_t)stream;
#define PGW_FWD(IB_, OB_)                                                                                                     \
  PG_KLAUNCH((warp_fwd3_kernel<IB_, OB_>), grid, dim3(256), lds, st, feat, aff, warps, lvl_masks, T, C, h, w, H0, W0, \
                     align_corners, tp, out, argmax, relu)
    const bool no_v5 = getenv("PG_WARP_FWD_V3") != nullptr;            // ablation switch (read per call: a test flips it): the LDS-tap-table kernel on bf16 storage too
    const int cpp8 = C / 8;
    if (v8 && !no_v5 && !(relu >> 8) && cpp8 >= 8 && cpp8 <= 64 && (cpp8 & (cpp8 - 1)) == 0 && T <= 2 * cpp8 && ((long)h * w * cpp8) % 64 == 0 &&
        (double)h * w * C * 2.0 < 2147483648.0 && w + h <= 8192) {
      const char* wenv = getenv("PG_WARP_FWD5_WGS");                    // workgroups per launch (read per call: the test walks several tiles per workgroup)
      const long wcap5 = wenv ? atol(wenv) : 8192;
      long wgs = ((long)h * w * cpp8 + 255) / 256;
      const long cap5 = wcap5 / N > 32 ? wcap5 / N : 32;
      if (wgs > cap5) wgs = cap5;
      if (argmax)
        PG_KLAUNCH(warp_fwd5_kernel<true>, dim3((unsigned)wgs, N), dim3(256), (size_t)(w + h) * 4, st, feat, aff, warps, lvl_masks, T, C, h, w, H0,
                   W0, align_corners, out, argmax, relu);
      else
        PG_KLAUNCH(warp_fwd5_kernel<false>, dim3((unsigned)wgs, N), dim3(256), (size_t)(w + h) * 4, st, feat, aff, warps, lvl_masks, T, C, h, w, H0,
                   W0, align_corners, out, argmax, relu);
    } else if (v8)
      PG_KLAUNCH((warp_fwd3_kernel<true, true, 8>), grid, dim3(256), lds, st, feat, aff, warps, lvl_masks, T, C, h, w, H0, W0,
                 align_corners, tp, out, argmax, relu);
    else if (ib && ob) PGW_FWD(true, true);
    else if (ib) PGW_FWD(true, false);
    else if (ob) PGW_FWD(false, true);
    else PGW_FWD(false, false);
#undef PGW_FWD
  }
  PG_LAUNCH_OK("pg_warp_mask_max_fwd");
  return 0;
}
extern "C" int pg_warp_mask_max_fwd(const float* feat, const float* aff, const float* warps, const float* lvl_masks,
                                    int32_t N, int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0,
                                    int32_t align_corners, float* out, uint8_t* argmax, void* stream) {
  return pg_warp_mask_max_fwd_io(feat, aff, warps, lvl_masks, N, T, C, h, w, H0, W0, align_corners, out, argmax, 0, stream);
}

extern "C" int pg_mask_bbox(const void* masks, int32_t is_f64, int32_t N, int32_t T, int32_t H0, int32_t W0, int32_t* bbox,
                            void* stream) {
  PG_REQUIRE(masks && bbox && N > 0 && T > 0 && H0 > 0 && W0 > 0, "pg_mask_bbox: bad arguments");
  if (is_f64)
    PG_KLAUNCH(mask_bbox_kernel<double>, dim3((unsigned)(N * T)), dim3(1024), 0, (hipStream_t)stream, (const double*)masks, H0, W0,
               (int*)bbox);
  else
    PG_KLAUNCH(mask_bbox_kernel<float>, dim3((unsigned)(N * T)), dim3(1024), 0, (hipStream_t)stream, (const float*)masks, H0, W0,
               (int*)bbox);
  PG_LAUNCH_OK("pg_mask_bbox");
  return 0;
}

// io_flags: bit 0 = `gout` is bf16, bit 1 = `dfeat` is bf16.  bbox: NULL, or the [N][T][4] boxes of pg_mask_bbox over the
// full-resolution masks the level masks were made from.
extern "C" int pg_warp_mask_max_bwd_bbox(const void* gout, const uint8_t* argmax, const float* warps, const float* lvl_masks,
                                         const int32_t* bbox, int32_t N, int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0,
                                         int32_t W0, int32_t align_corners, void* dfeat, int32_t io_flags, void* stream) {
  PG_REQUIRE(gout && argmax && warps && lvl_masks && dfeat, "pg_warp_mask_max_bwd: null pointer");
  PG_REQUIRE(T >= 1 && T <= MAXT && C % 4 == 0 && N > 0, "pg_warp_mask_max_bwd: need T<=32, C%%4==0");
  static const bool no_gather = getenv("PG_WARP_BWD_SCATTER") != nullptr;       // ablation: round-1 scatter kernel only
  const bool gb = io_flags & 1, db = io_flags & 2;
  hipStream_t st = (hipStream_t)stream;
#define PGW_BWD(KERNEL, GRID, ...)                                                                   \
  do {                                                                                               \
    if (gb && db) PG_KLAUNCH((KERNEL<true, true>), GRID, dim3(256), 0, st, __VA_ARGS__);     \
    else if (gb) PG_KLAUNCH((KERNEL<true, false>), GRID, dim3(256), 0, st, __VA_ARGS__);     \
    else if (db) PG_KLAUNCH((KERNEL<false, true>), GRID, dim3(256), 0, st, __VA_ARGS__);     \
    else PG_KLAUNCH((KERNEL<false, false>), GRID, dim3(256), 0, st, __VA_ARGS__);            \
  } while (0)
#define PGW_DET(GRID, ...)                                                                              \
  do {                                                                                                   \
    if (gb && db) PG_KLAUNCH((warp_bwd_det_kernel<true, true>), GRID, dim3(64), 0, st, __VA_ARGS__);      \
    else if (gb) PG_KLAUNCH((warp_bwd_det_kernel<true, false>), GRID, dim3(64), 0, st, __VA_ARGS__);      \
    else if (db) PG_KLAUNCH((warp_bwd_det_kernel<false, true>), GRID, dim3(64), 0, st, __VA_ARGS__);      \
    else PG_KLAUNCH((warp_bwd_det_kernel<false, false>), GRID, dim3(64), 0, st, __VA_ARGS__);             \
  } while (0)
  const long all_tiles = (long)N * (((long)h * w + GATHER_PIX - 1) / GATHER_PIX);
  if (T <= GATHER_T && !no_gather && (long)h * w < (1l << 24) && (double)h * w * C < 2147483648.0 && h <= GATHER_MAXDIM && w <= GATHER_MAXDIM && all_tiles <= GATHER_OVF_MAX &&
      N < 2048) {
    // gather kernel OVERWRITES dfeat (narrow transforms), then the scatter kernel adds the wide ones
    static const int gcap = getenv("PG_WARP_BWD_TILES") ? atoi(getenv("PG_WARP_BWD_TILES")) : 256;     // workgroups per sample
    int gtiles = (h * w + GATHER_PIX - 1) / GATHER_PIX;
    if (gtiles > gcap) gtiles = gcap;
    static const bool no_v8b = getenv("PG_WARP_NO_V8") != nullptr;
    int ac_g = align_corners;                            // the gather launches' copy (timing build: + experiment bits)
#ifdef PG_TIMING_EXPERIMENTS
    { static const int wd = getenv("PG_DEBUG_WARP_BWD") ? atoi(getenv("PG_DEBUG_WARP_BWD")) : 0; ac_g |= wd << 8; }
#endif
    const int det_g = deterministic() ? 1 : 0;           // PG_DETERMINISTIC: sorted candidate lists here, the ordered single-writer kernel for the wide transforms below
    static void* ovf_dev = nullptr;                      // g_gather_ovf: tiles whose 48-entry lists overflowed
    if (ovf_dev == nullptr) PG_REQUIRE(hipGetSymbolAddress(&ovf_dev, HIP_SYMBOL(g_gather_ovf)) == hipSuccess, "pg_warp_mask_max_bwd: symbol");
    PG_MEMSET_ASYNC(ovf_dev, 0, 4, st);
    const size_t glds = (size_t)(w + h) * 4;
    const dim3 g1(gtiles, N), g2(64);                    // second launch: worst-case capacity over the overflow list (usually empty)
    if (gb && db && C % 8 == 0 && !no_v8b) {
      PG_KLAUNCH((warp_bwd_gather_kernel<true, true, 8, false>), g1, dim3(256), glds, st, gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0,
                 ac_g, dfeat, (const int*)bbox, det_g);
      PG_KLAUNCH((warp_bwd_gather_kernel<true, true, 8, true>), g2, dim3(256), glds, st, gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0,
                 ac_g, dfeat, (const int*)bbox, det_g);
    } else {
#define PGW_GATHER(GBv, DBv)                                                                                                    \
  do {                                                                                                                          \
    PG_KLAUNCH((warp_bwd_gather_kernel<GBv, DBv, 4, false>), g1, dim3(256), glds, st, gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0, \
               ac_g, dfeat, (const int*)bbox, det_g);                                                                         \
    PG_KLAUNCH((warp_bwd_gather_kernel<GBv, DBv, 4, true>), g2, dim3(256), glds, st, gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0,  \
               ac_g, dfeat, (const int*)bbox, det_g);                                                                         \
  } while (0)
      if (gb && db) PGW_GATHER(true, true);
      else if (gb) PGW_GATHER(true, false);
      else if (db) PGW_GATHER(false, true);
      else PGW_GATHER(false, false);
#undef PGW_GATHER
    }
    PG_LAUNCH_OK("pg_warp_mask_max_bwd (gather)");
    // (a sample without wide transforms costs one early-exiting workgroup round: keep that grid small)
    if (deterministic())      // ordered single-writer form instead of float atomics (bit-repeatable; slow, rare)
      PGW_DET(dim3((C + 63) / 64, N), gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0, align_corners, dfeat, 1);
    else
      PGW_BWD(warp_bwd_kernel, dim3(min(warp_bwd_grid(C, h, w), 256), N), gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0,
              align_corners, dfeat, 1);
    PG_LAUNCH_OK("pg_warp_mask_max_bwd (wide transforms)");
    return 0;
  }
  PG_MEMSET_ASYNC(dfeat, 0, (db ? 2 : 4) * (size_t)N * h * w * C, st);
  if (deterministic())
    PGW_DET(dim3((C + 63) / 64, N), gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0, align_corners, dfeat, 0);
  else
    PGW_BWD(warp_bwd_kernel, dim3(warp_bwd_grid(C, h, w), N), gout, argmax, warps, lvl_masks, T, C, h, w, H0, W0, align_corners, dfeat, 0);
#undef PGW_DET
#undef PGW_BWD
  PG_LAUNCH_OK("pg_warp_mask_max_bwd");
  return 0;
}
// test aid: how many tiles the LAST gather-form backward handed to its full-capacity second launch (synchronises the device)
extern "C" int pg_debug_warp_gather_overflows(int32_t* count) {
  PG_REQUIRE(count != nullptr, "pg_debug_warp_gather_overflows: null pointer");
  PG_REQUIRE(hipDeviceSynchronize() == hipSuccess, "pg_debug_warp_gather_overflows: synchronize");
  int c = 0;
  PG_REQUIRE(hipMemcpyFromSymbol(&c, HIP_SYMBOL(pg::g_gather_ovf), sizeof(int), 0, hipMemcpyDeviceToHost) == hipSuccess,
             "pg_debug_warp_gather_overflows: copy");
  *count = c;
  return 0;
}
extern "C" int pg_warp_mask_max_bwd_io(const void* gout, const uint8_t* argmax, const float* warps,
                                       const float* lvl_masks, int32_t N, int32_t T, int32_t C, int32_t h, int32_t w,
                                       int32_t H0, int32_t W0, int32_t align_corners, void* dfeat, int32_t io_flags, void* stream) {
  return pg_warp_mask_max_bwd_bbox(gout, argmax, warps, lvl_masks, nullptr, N, T, C, h, w, H0, W0, align_corners, dfeat, io_flags,
                                   stream);
}
extern "C" int pg_warp_mask_max_bwd(const float* gout, const uint8_t* argmax, const float* warps,
                                    const float* lvl_masks, int32_t N, int32_t T, int32_t C, int32_t h, int32_t w,
                                    int32_t H0, int32_t W0, int32_t align_corners, float* dfeat, void* stream) {
  return pg_warp_mask_max_bwd_io(gout, argmax, warps, lvl_masks, N, T, C, h, w, H0, W0, align_corners, dfeat, 0, stream);
}
