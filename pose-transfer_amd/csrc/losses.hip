// Loss kernels for the Deformable-GAN step (reference models/pose_gan.py), gfx950.  All HBM-bound or tiny.
//  * gan_logloss: sigmoid + -mean log(o+1e-7) / -mean log(1-o+1e-7) + gradient wrt the logits (pose_gan.py:90-98,140-160)
//  * l1_loss:     nn.L1Loss value + sign gradient (pose_gan.py:66,105)
//  * vgg conv1_1: Feature_Extractor('block1_conv2') with the view-not-permute pre-process (utils/pose_utils.py:312-338)
//  * nn_loss:     nearest-neighbour L1 over an area x area window, value + gradient in one pass — the reference's 25x
//                 materialisation (419 MB/img) never exists (pose_gan.py:173-199)
#include "common.h"
#include <cstdlib>

namespace pg {

__global__ __launch_bounds__(256) void gan_logloss_kernel(const float* x, long count, int mode, float scale,
                                                          float* loss, float* dx, float* sig) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const float v = x[i];
    const float o = 1.f / (1.f + expf(-v));
    if (sig) sig[i] = o;
    const float so = o * (1.f - o);   // d sigmoid / dx
    if (mode == 0) {
      acc += -logf(o + 1e-7f);
      if (dx) dx[i] = scale * (-so / (o + 1e-7f));
    } else {
      acc += -logf((1.f - o) + 1e-7f);
      if (dx) dx[i] = scale * (so / ((1.f - o) + 1e-7f));
    }
  }
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, scale * t);
}

__global__ __launch_bounds__(256) void l1_loss_kernel(const float* p, const float* t, long count, float scale,
                                                      float* loss, float* g, int accumulate) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const float d = p[i] - t[i];
    acc += fabsf(d);
    if (g) {
      const float s = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
      g[i] = accumulate ? g[i] + s : s;
    }
  }
  const float tt = block_sum_256(acc, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, scale * tt);
}

__global__ __launch_bounds__(256) void tanh_bwd_kernel(float* g, const float* out, long count) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
    const float o = out[i];
    g[i] = g[i] * (1.f - o * o);
  }
}

__constant__ float kVggMean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float kVggStd[3] = {0.229f, 0.224f, 0.225f};

// 16 lanes per pixel, 4 output channels per lane.  x NCHW [N][3][H][W] -> feat NHWC [N][H][W][64].
__global__ __launch_bounds__(256) void vgg_fwd_kernel(const float* x, const float* w, const float* b, int N, int H,
                                                      int W, float* feat) {
  __shared__ float ws[64 * 27 + 64];
  for (int i = threadIdx.x; i < 64 * 27; i += 256) ws[i] = w[i];
  if (threadIdx.x < 64) ws[64 * 27 + threadIdx.x] = b[threadIdx.x];
  __syncthreads();
  const long npix = (long)N * H * W;
  const int lane16 = threadIdx.x & 15;
  for (long pix = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pix < npix; pix += (long)gridDim.x * 16) {
    const int n = (int)(pix / ((long)H * W));
    const int rem = (int)(pix - (long)n * H * W);
    const int y = rem / W, xx = rem - y * W;
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int yy = y + r - 1, xs = xx + s - 1;
          float v = 0.f;
          if (yy >= 0 && yy < H && xs >= 0 && xs < W) {
            const long k = ((long)c * H + yy) * W + xs;   // flat per-sample offset: mean/std index = k % 3
            const int q = (int)(k % 3);
            v = (x[(long)n * 3 * H * W + k] - kVggMean[q]) / kVggStd[q];
          }
          in[c * 9 + r * 3 + s] = v;
        }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = lane16 * 4 + e;
      float acc = ws[64 * 27 + co];
#pragma unroll
      for (int q = 0; q < 27; ++q) acc += in[q] * ws[co * 27 + q];
      o[e] = acc > 0.f ? acc : 0.f;
    }
    *reinterpret_cast<float4*>(feat + pix * 64 + lane16 * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Round 2: the same arithmetic, tiled.  The kernel above evaluates the input normalisation ((x - mean[k%3]) / std[k%3], a
// 64-bit modulo and an IEEE division) 27 times in EVERY one of the 16 lanes of a pixel and reads its 108 weights from LDS
// per pixel: 2.1 ms for 2 M pixels (0.03 of the HBM rate of its 256 B/pixel output).  Here a workgroup owns a 16 x 16 pixel
// tile: the normalised 18 x 18 x 3 input patch is built ONCE in LDS (4 values per thread), a lane keeps the 27 weights of
// its 4 output channels in registers for all the tiles it walks, and a pixel costs 27 broadcast LDS reads + 108 FMAs per
// lane.  Same operation order per output (bias, then q = (c, r, s) ascending).
__global__ __launch_bounds__(256) void vgg_fwd_tile_kernel(const float* x, const float* w, const float* b, int N, int H,
                                                           int W, int tiles_x, int tiles_y, float* feat) {
  __shared__ float patch[3 * 18 * 18];
  const int tid = threadIdx.x, lane16 = tid & 15, px = tid >> 4;
  float wr[4][27], br[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    br[e] = b[lane16 * 4 + e];
#pragma unroll
    for (int q = 0; q < 27; ++q) wr[e][q] = w[(lane16 * 4 + e) * 27 + q];
  }
  const int ntiles = tiles_x * tiles_y * N;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int bb = t;
    const int tx = bb % tiles_x; bb /= tiles_x;
    const int ty = bb % tiles_y;
    const int n = bb / tiles_y;
    const int y0 = ty * 16, x0 = tx * 16;
    __syncthreads();
    for (int e = tid; e < 3 * 18 * 18; e += 256) {
      const int c = e / 324, rr = (e / 18) % 18, cc = e % 18;
      const int yy = y0 + rr - 1, xs = x0 + cc - 1;
      float v = 0.f;
      if (yy >= 0 && yy < H && xs >= 0 && xs < W) {
        const long k = ((long)c * H + yy) * W + xs;     // flat per-sample offset: mean/std index = k % 3 (the reference's broadcast)
        const int q = (int)(k % 3);
        v = (x[(long)n * 3 * H * W + k] - kVggMean[q]) / kVggStd[q];
      }
      patch[e] = v;
    }
    __syncthreads();
    const int ox = x0 + px;
#pragma unroll 2
    for (int row = 0; row < 16; ++row) {
      const int oy = y0 + row;
      float in[27];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s2 = 0; s2 < 3; ++s2) in[c * 9 + r * 3 + s2] = patch[c * 324 + (row + r) * 18 + px + s2];
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = br[e];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc += in[q] * wr[e][q];
        o[e] = acc > 0.f ? acc : 0.f;
      }
      if (oy < H && ox < W)
        *reinterpret_cast<float4*>(feat + (((long)n * H + oy) * W + ox) * 64 + lane16 * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// gout[n][ci][y][x] += ( sum_{r,s,co} dF[n, y+1-r, x+1-s, co] * w[co][ci][r][s] ) / std[k%3]
__global__ __launch_bounds__(256) void vgg_dgrad_kernel(const float* dfeat, const float* w, int N, int H, int W,
                                                        float* gout) {
  __shared__ float ws[64 * 27];
  for (int i = threadIdx.x; i < 64 * 27; i += 256) ws[i] = w[i];
  __syncthreads();
  const long npix = (long)N * H * W;
  const int lane16 = threadIdx.x & 15;
  const long iters = (npix + (long)gridDim.x * 16 - 1) / ((long)gridDim.x * 16);
  for (long itn = 0; itn < iters; ++itn) {
    const long pix = (itn * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 4);
    const bool pv = pix < npix;
    const int n = pv ? (int)(pix / ((long)H * W)) : 0;
    const int rem = pv ? (int)(pix - (long)n * H * W) : 0;
    const int y = rem / W, xx = rem - y * W;
    float acc[3] = {0.f, 0.f, 0.f};
    if (pv) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int yy = y + 1 - r, xs = xx + 1 - s;
          if (yy < 0 || yy >= H || xs < 0 || xs >= W) continue;
          const float4 d = *reinterpret_cast<const float4*>(dfeat + (((long)n * H + yy) * W + xs) * 64 + lane16 * 4);
          const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = lane16 * 4 + e;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] += dv[e] * ws[co * 27 + c * 9 + r * 3 + s];
          }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    }
    if (pv && lane16 < 3) {
      const int c = lane16;
      const long k = ((long)c * H + y) * W + xx;
      const float v = (c == 0 ? acc[0] : (c == 1 ? acc[1] : acc[2])) / kVggStd[(int)(k % 3)];
      gout[(long)n * 3 * H * W + k] += v;
    }
  }
}

// LPP lanes per pixel (C = 4*LPP channels).  P, G: NHWC.
template <int LPP>
__global__ __launch_bounds__(256) void nn_loss_kernel(const float* P, const float* G, int N, int H, int W, int area,
                                                      float scale, int relu_mask, float* loss, float* dP) {
  __shared__ float red[4];
  constexpr int C = LPP * 4;
  constexpr int PPB = 256 / LPP;
  const long npix = (long)N * H * W;
  const int sub = threadIdx.x % LPP;
  const int half = area / 2;
  float lacc = 0.f;
  const long iters = (npix + (long)gridDim.x * PPB - 1) / ((long)gridDim.x * PPB);
  for (long itn = 0; itn < iters; ++itn) {
    const long pix = (itn * gridDim.x + blockIdx.x) * PPB + threadIdx.x / LPP;
    const bool pv = pix < npix;
    const int n = pv ? (int)(pix / ((long)H * W)) : 0;
    const int rem = pv ? (int)(pix - (long)n * H * W) : 0;
    const int y = rem / W, x = rem - y * W;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pv) p = *reinterpret_cast<const float4*>(P + pix * C + sub * 4);
    float best = INFINITY;
    float4 gbest = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int di = 0; di < area; ++di)
      for (int dj = 0; dj < area; ++dj) {
        const int yy = y + di - half, xx = x + dj - half;
        float4 g = make_float4(-10000.f, -10000.f, -10000.f, -10000.f);   // ConstantPad2d(..., -10000), pose_gan.py:176
        if (pv && yy >= 0 && yy < H && xx >= 0 && xx < W)
          g = *reinterpret_cast<const float4*>(G + (((long)n * H + yy) * W + xx) * C + sub * 4);
        float d = (fabsf(g.x - p.x) + fabsf(g.y - p.y)) + (fabsf(g.z - p.z) + fabsf(g.w - p.w));
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (d < best) { best = d; gbest = g; }
      }
    if (pv) {
      if (sub == 0) lacc += best;
      if (dP) {
        float4 o;
        auto sg = [&](float pp, float gg) {
          float s = pp > gg ? scale : (pp < gg ? -scale : 0.f);
          if (relu_mask && !(pp > 0.f)) s = 0.f;
          return s;
        };
        o.x = sg(p.x, gbest.x); o.y = sg(p.y, gbest.y); o.z = sg(p.z, gbest.z); o.w = sg(p.w, gbest.w);
        *reinterpret_cast<float4*>(dP + pix * C + sub * 4) = o;
      }
    }
  }
  const float t = block_sum_256(lacc, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, scale * t);
}

static int ew_blocks(long count) {
  long b = (count + 256 * 4 - 1) / (256 * 4);
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// Round 2, C = 64 (the VGG block1 feature loss of configs[3]): one workgroup = an 8 x 8 tile of predicted pixels, FOUR lanes
// per pixel (16 channels each).  The kernel above gives a pixel 16 lanes, pays a 4-stage cross-lane sum for every one of
// the area^2 offsets (18 VALU per lane and offset: VALU-bound at 1.2 ms for 2 M pixels) and re-reads G from L2 area^2
// times.  Here the (8 + 2 half)^2 ground-truth tile sits in LDS with a pixel pitch of 68 floats (272 B: the float4 reads of a
// 16-lane group — 4 pixels x 4 channel quarters — fall into disjoint bank groups), a lane keeps its 16 predicted channels in
// registers, and an offset costs 4 ds_read_b128 + 32 VALU + a 2-stage cross-lane sum.  39 KB of LDS per workgroup keeps
// four workgroups per CU resident, which is what hides the LDS latency (a 16 x 16 tile with one lane per pixel was tried:
// 109 KB, one wave per SIMD, every read's latency exposed — no faster than the kernel above).  Offsets are walked in the
// same (di, dj) order with the same strict '<', so ties resolve as before; the winning offset is re-read for the gradient.
template <int AREA>
__global__ __launch_bounds__(256) void nn_loss_tile64_kernel(const float* P, const float* G, int N, int H, int W, float scale,
                                                             int relu_mask, int tiles_x, int tiles_y, float* loss, float* dP) {
  constexpr int HALF = AREA / 2, TS = 8 + 2 * HALF, PITCH = 68;
  __shared__ __attribute__((aligned(16))) float gt[TS * TS * PITCH + 4];     // tile + 4 floats of reduction scratch
  float* const red = gt + TS * TS * PITCH;
  const int tid = threadIdx.x;
  int bb = blockIdx.x;
  const int tx = bb % tiles_x; bb /= tiles_x;
  const int ty = bb % tiles_y;
  const int n = bb / tiles_y;
  const int y0 = ty * 8, x0 = tx * 8;
  // ---- stage the ground-truth tile (ConstantPad2d(-10000) outside the image, reference pose_gan.py:176); branch-free
  //      loads in batches so that a batch costs one memory latency
  constexpr int NE = TS * TS * 16, U = 4;
  for (int e0 = tid; e0 < NE; e0 += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 256 * u;
      const int c4 = e & 15, pp = e >> 4;
      const int rr = pp / TS, cc = pp - rr * TS;
      const int yy = y0 + rr - HALF, xx = x0 + cc - HALF;
      const bool ok = (e < NE) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
      const long off = ok ? (((long)n * H + yy) * W + xx) * 64 + c4 * 4 : 0;
      v[u] = *reinterpret_cast<const float4*>(G + off);
      if (!ok) v[u] = make_float4(-10000.f, -10000.f, -10000.f, -10000.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 256 * u;
      if (e < NE) *reinterpret_cast<float4*>(gt + (e >> 4) * PITCH + (e & 15) * 4) = v[u];
    }
  }
  const int sub = tid & 3, lp = tid >> 2, ly = lp >> 3, lx = lp & 7;
  const int y = y0 + ly, x = x0 + lx;
  const bool pv = y < H && x < W;
  const long pix = ((long)n * H + (pv ? y : 0)) * W + (pv ? x : 0);
  float4 p[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) p[c] = *reinterpret_cast<const float4*>(P + pix * 64 + sub * 16 + c * 4);
  __syncthreads();
  float best = INFINITY;
  int bo = 0;
  for (int di = 0; di < AREA; ++di)
#pragma unroll
    for (int dj = 0; dj < AREA; ++dj) {
      const float* gp = gt + ((ly + di) * TS + lx + dj) * PITCH + sub * 16;
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 g = *reinterpret_cast<const float4*>(gp + c * 4);
        d0 += fabsf(g.x - p[c].x); d1 += fabsf(g.y - p[c].y); d2 += fabsf(g.z - p[c].z); d3 += fabsf(g.w - p[c].w);
      }
      float d = (d0 + d1) + (d2 + d3);
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      if (d < best) { best = d; bo = di * AREA + dj; }
    }
  float lacc = 0.f;
  if (pv) {
    if (sub == 0) lacc = best;
    if (dP) {
      const int di = bo / AREA, dj = bo - di * AREA;
      const float* gp = gt + ((ly + di) * TS + lx + dj) * PITCH + sub * 16;
      auto sg = [&](float pp, float gg) {
        float s = pp > gg ? scale : (pp < gg ? -scale : 0.f);
        if (relu_mask && !(pp > 0.f)) s = 0.f;
        return s;
      };
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 g = *reinterpret_cast<const float4*>(gp + c * 4);
        float4 o;
        o.x = sg(p[c].x, g.x); o.y = sg(p[c].y, g.y); o.z = sg(p[c].z, g.z); o.w = sg(p[c].w, g.w);
        *reinterpret_cast<float4*>(dP + pix * 64 + sub * 16 + c * 4) = o;
      }
    }
  }
  const float t = block_sum_256(lacc, red);
  if (tid == 0 && loss) atomicAdd(loss, scale * t);
}

template <int AREA>
static int launch_nn_tile64(const float* P, const float* G, int N, int H, int W, float scale, int relu_mask, float* loss, float* dP,
                            hipStream_t st) {
  const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
  PG_KLAUNCH(nn_loss_tile64_kernel<AREA>, dim3((unsigned)(tiles_x * tiles_y * N)), dim3(256), 0, st, P, G, N, H, W, scale,
                     relu_mask, tiles_x, tiles_y, loss, dP);
  return 0;
}

}  // namespace pg

using namespace pg;

extern "C" int pg_gan_logloss(const float* logits, int64_t count, int32_t mode, float scale, float* loss,
                              float* dlogits, float* sig, void* stream) {
  PG_REQUIRE(logits && count > 0 && (mode == 0 || mode == 1), "pg_gan_logloss: bad arguments");
  PG_KLAUNCH(gan_logloss_kernel, dim3(deterministic() ? 1 : ew_blocks(count)), dim3(256), 0, (hipStream_t)stream, logits, (long)count,
                     mode, scale, loss, dlogits, sig);
  PG_LAUNCH_OK("pg_gan_logloss");
  return 0;
}

extern "C" int pg_l1_loss(const float* pred, const float* target, int64_t count, float scale, float* loss, float* gout,
                          int32_t accumulate, void* stream) {
  PG_REQUIRE(pred && target && count > 0, "pg_l1_loss: bad arguments");
  PG_KLAUNCH(l1_loss_kernel, dim3(deterministic() ? 1 : ew_blocks(count)), dim3(256), 0, (hipStream_t)stream, pred, target,
                     (long)count, scale, loss, gout, accumulate);
  PG_LAUNCH_OK("pg_l1_loss");
  return 0;
}

extern "C" int pg_tanh_bwd(float* g, const float* out, int64_t count, void* stream) {
  PG_REQUIRE(g && out && count > 0, "pg_tanh_bwd: bad arguments");
  PG_KLAUNCH(tanh_bwd_kernel, dim3(ew_blocks(count)), dim3(256), 0, (hipStream_t)stream, g, out, (long)count);
  PG_LAUNCH_OK("pg_tanh_bwd");
  return 0;
}

extern "C" int pg_vgg_conv1_relu_fwd(const float* x, const float* w, const float* b, int32_t N, int32_t H, int32_t W,
                                     float* feat, void* stream) {
  PG_REQUIRE(x && w && b && feat && N > 0, "pg_vgg_conv1_relu_fwd: bad arguments");
  static const bool v1 = getenv("PG_VGG_FWD_V1") != nullptr;       // ablation switch: the round-1 per-pixel kernel
  if (v1) {
    const long npix = (long)N * H * W;
    long blocks = (npix + 15) / 16;
    if (blocks > 8192) blocks = 8192;
    PG_KLAUNCH(vgg_fwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, w, b, N, H, W, feat);
  } else {
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    long blocks = (long)tiles_x * tiles_y * N;
    if (blocks > 2048) blocks = 2048;
    PG_KLAUNCH(vgg_fwd_tile_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, w, b, N, H, W, tiles_x,
                       tiles_y, feat);
  }
  PG_LAUNCH_OK("pg_vgg_conv1_relu_fwd");
  return 0;
}

extern "C" int pg_vgg_conv1_dgrad(const float* dfeat, const float* w, int32_t N, int32_t H, int32_t W, float* gout,
                                  void* stream) {
  PG_REQUIRE(dfeat && w && gout && N > 0, "pg_vgg_conv1_dgrad: bad arguments");
  const long npix = (long)N * H * W;
  long blocks = (npix + 15) / 16;
  if (blocks > 8192) blocks = 8192;
  PG_KLAUNCH(vgg_dgrad_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dfeat, w, N, H, W, gout);
  PG_LAUNCH_OK("pg_vgg_conv1_dgrad");
  return 0;
}

extern "C" int pg_nn_loss(const float* P, const float* G, int32_t N, int32_t H, int32_t W, int32_t C, int32_t area,
                          float scale, int32_t relu_mask, float* loss, float* dP, void* stream) {
  PG_REQUIRE(P && G && N > 0 && area >= 1 && (area & 1), "pg_nn_loss: bad arguments (area must be odd)");
  const long npix = (long)N * H * W;
  hipStream_t st = (hipStream_t)stream;
#define PG_NN(LPP)                                                                                     \
  {                                                                                                    \
    long blocks = (npix + (256 / LPP) - 1) / (256 / LPP);                                              \
    if (blocks > 8192) blocks = 8192;                                                                  \
    PG_KLAUNCH(nn_loss_kernel<LPP>, dim3((int)blocks), dim3(256), 0, st, P, G, N, H, W, area,  \
                       scale, relu_mask, loss, dP);                                                    \
  }
  if (C == 64 && (area == 3 || area == 5 || area == 7) && !env().nn_loss_v1) {
    int rc = area == 3 ? launch_nn_tile64<3>(P, G, N, H, W, scale, relu_mask, loss, dP, st)
           : area == 5 ? launch_nn_tile64<5>(P, G, N, H, W, scale, relu_mask, loss, dP, st)
                       : launch_nn_tile64<7>(P, G, N, H, W, scale, relu_mask, loss, dP, st);
    if (rc) return rc;
    PG_LAUNCH_OK("pg_nn_loss");
    return 0;
  }
  switch (C) {
    case 4: PG_NN(1); break;
    case 8: PG_NN(2); break;
    case 16: PG_NN(4); break;
    case 32: PG_NN(8); break;
    case 64: PG_NN(16); break;
    case 128: PG_NN(32); break;
    case 256: PG_NN(64); break;
    default: PG_FAIL(1, "pg_nn_loss: C=%d unsupported (need 4,8,...,256)", C);
  }
#undef PG_NN
  PG_LAUNCH_OK("pg_nn_loss");
  return 0;
}
