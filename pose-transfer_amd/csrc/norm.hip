// Per-sample normalisation kernels, gfx950.
// Reference: nn.InstanceNorm3d(1, eps=1e-3, affine=True) applied on x.unsqueeze(1)
// (models/networks.py:159,166-169) == LayerNorm over (C,H,W) per sample with ONE scalar gamma/beta.
//
// The normalisation itself is never materialised: `finalize` emits the per-sample affine (a_n, b_n) that consumers
// (pg_conv / pg_conv_wgrad / pg_warp_mask_max_fwd prologues) apply on load.  These kernels are the HBM-bound
// reductions around it: one read of the activation for the statistics; backward reads (dz, y) once for the two
// per-sample sums and once more to rewrite dz -> dy in place.
#include "common.h"
#include <cstdlib>

namespace pg {

__global__ __launch_bounds__(256) void norm_stats_kernel(const float* y, long L, double* sums) {
  __shared__ double red[8];
  const int n = blockIdx.y;
  const float* b = y + (long)n * L;
  float s = 0.f, q = 0.f;
  const long L4 = L >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(b)[i];
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0)
    for (long i = (L4 << 2) + threadIdx.x; i < L; i += 256) { const float v = b[i]; s += v; q += v * v; }
  double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w] = ds; red[4 + w] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* slot = sums + ((long)n * PG_STAT_SLOTS + (blockIdx.x % PG_STAT_SLOTS)) * 2;
    atomicAdd(&slot[0], red[0] + red[1] + red[2] + red[3]);
    atomicAdd(&slot[1], red[4] + red[5] + red[6] + red[7]);
  }
}

__global__ void norm_finalize_kernel(const double* sums, const float* gamma, const float* beta, int N, long L,
                                     float eps, float* mr, float* aff) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < PG_STAT_SLOTS; ++k) {     // partial sums are spread over slots to keep atomics off one address
    s1 += sums[((long)n * PG_STAT_SLOTS + k) * 2];
    s2 += sums[((long)n * PG_STAT_SLOTS + k) * 2 + 1];
  }
  const double mean = s1 / (double)L;
  double var = s2 / (double)L - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double g = (double)gamma[0], b = (double)beta[0];
  mr[2 * n] = (float)mean;
  mr[2 * n + 1] = (float)rstd;
  aff[2 * n] = (float)(g * rstd);
  aff[2 * n + 1] = (float)(b - g * mean * rstd);
}

__device__ __forceinline__ unsigned pack_bf16_rne(float lo, float hi) {      // RNE, lo -> bits 0..15 (as pg_materialise_bf16)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// 4 consecutive elements of an fp32 (T = float) or bf16 (T = unsigned short: bf16 STORAGE, round 3) tensor
template <typename T>
__device__ __forceinline__ float4 ld4(const T* base, long i4) {
  if constexpr (sizeof(T) == 4) {
    return reinterpret_cast<const float4*>(base)[i4];
  } else {
    const uint2 u = reinterpret_cast<const uint2*>(base)[i4];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
}
template <typename T>
__device__ __forceinline__ void st4(T* base, long i4, float4 v) {
  if constexpr (sizeof(T) == 4) {
    reinterpret_cast<float4*>(base)[i4] = v;
  } else {
    uint2 u;
    u.x = pack_bf16_rne(v.x, v.y); u.y = pack_bf16_rne(v.z, v.w);
    reinterpret_cast<uint2*>(base)[i4] = u;
  }
}
template <typename T>
__device__ __forceinline__ float ld1(const T* base, long i) {
  if constexpr (sizeof(T) == 4) return base[i];
  else return __uint_as_float((unsigned)base[i] << 16);
}

template <typename TD, typename TY>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const TD* dz, const TY* y, const float* mr, long L,
                                                              double* bsums) {
  __shared__ double red[8];
  const int n = blockIdx.y;
  const float mean = mr[2 * n], rstd = mr[2 * n + 1];
  const TD* bd = dz + (long)n * L;
  const TY* by = y + (long)n * L;
  float s = 0.f, q = 0.f;
  const long L4 = L >> 2;
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < L4; i += 4 * stride) {           // eight independent loads in flight per lane
    float4 d[4], v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      d[u] = ld4(bd, i + u * stride);
      v[u] = ld4(by, i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += (d[u].x + d[u].y) + (d[u].z + d[u].w);
      q += (d[u].x * ((v[u].x - mean) * rstd) + d[u].y * ((v[u].y - mean) * rstd)) +
           (d[u].z * ((v[u].z - mean) * rstd) + d[u].w * ((v[u].w - mean) * rstd));
    }
  }
  for (; i < L4; i += stride) {
    const float4 d = ld4(bd, i);
    const float4 v = ld4(by, i);
    s += (d.x + d.y) + (d.z + d.w);
    q += (d.x * ((v.x - mean) * rstd) + d.y * ((v.y - mean) * rstd)) +
         (d.z * ((v.z - mean) * rstd) + d.w * ((v.w - mean) * rstd));
  }
  if (blockIdx.x == 0)
    for (long i = (L4 << 2) + threadIdx.x; i < L; i += 256) { s += ld1(bd, i); q += ld1(bd, i) * ((ld1(by, i) - mean) * rstd); }
  double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w] = ds; red[4 + w] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&bsums[2 * n], red[0] + red[1] + red[2] + red[3]);
    atomicAdd(&bsums[2 * n + 1], red[4] + red[5] + red[6] + red[7]);
  }
}

template <typename TD, typename TY>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(TD* dz, const TY* y, const float* mr,
                                                             const double* bsums, const float* gamma, int N, long L,
                                                             float* dgamma, float* dbeta, unsigned short* dy_bf16) {
  const int n = blockIdx.y;
  const float mean = mr[2 * n], rstd = mr[2 * n + 1];
  const float g = gamma[0];
  const float m1 = (float)(bsums[2 * n] / (double)L);
  const float m2 = (float)(bsums[2 * n + 1] / (double)L);
  const float k = g * rstd;
  TD* bd = dz + (long)n * L;
  const TY* by = y + (long)n * L;
  const long L4 = L >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < L4; i += (long)gridDim.x * 256) {
    float4 d = ld4(bd, i);
    const float4 v = ld4(by, i);
    d.x = k * (d.x - m1 - ((v.x - mean) * rstd) * m2);
    d.y = k * (d.y - m1 - ((v.y - mean) * rstd) * m2);
    d.z = k * (d.z - m1 - ((v.z - mean) * rstd) * m2);
    d.w = k * (d.w - m1 - ((v.w - mean) * rstd) * m2);
    st4(bd, i, d);
    if (dy_bf16) {           // the bf16 operand copy the data- / weight-gradient contractions read (bf16 data path)
      uint2 pk;
      pk.x = pack_bf16_rne(d.x, d.y); pk.y = pack_bf16_rne(d.z, d.w);
      reinterpret_cast<uint2*>(dy_bf16 + (long)n * L)[i] = pk;
    }
  }
  if (blockIdx.x == 0)
    for (long i = (L4 << 2) + threadIdx.x; i < L; i += 256) {
      const float r = k * (ld1(bd, i) - m1 - ((ld1(by, i) - mean) * rstd) * m2);
      if constexpr (sizeof(TD) == 4) bd[i] = r;
      else bd[i] = (unsigned short)(pack_bf16_rne(r, 0.f) & 0xffffu);
      if (dy_bf16) dy_bf16[(long)n * L + i] = (unsigned short)(pack_bf16_rne(r, 0.f) & 0xffffu);
    }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    double sg = 0.0, sb = 0.0;
    for (int i = 0; i < N; ++i) { sb += bsums[2 * i]; sg += bsums[2 * i + 1]; }
    if (dgamma) atomicAdd(dgamma, (float)sg);
    if (dbeta) atomicAdd(dbeta, (float)sb);
  }
}

// norm_bwd_reduce ends every workgroup with two double atomics on ONE 64-byte line (bsums [N][2]); same-address
// atomics serialise (measured on the bias gradient: ~90 ns per workgroup), so that kernel gets few, deep workgroups
static int norm_blocks_bwd(long L, int N) {
  long b = (L / 4 + 256 * 16 - 1) / (256 * 16);
  const long cap = N >= 4 ? 64 : N == 3 ? 96 : N == 2 ? 128 : 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

static int norm_blocks(long L) {
  // streaming passes on this chip prefer MANY workgroups with one 16-byte chunk per lane over few deep ones (swept at batch
  // 32: 8 chunks per lane / 512 workgroups per sample 2.04 ms, 1 / 8192: 1.87 ms for the norm-backward apply passes of a step)
  static const int per = getenv("PG_NORM_PER") ? atoi(getenv("PG_NORM_PER")) : 1;
  static const int cap = getenv("PG_NORM_CAP") ? atoi(getenv("PG_NORM_CAP")) : 8192;
  long b = (L / 4 + 256 * per - 1) / (256 * per);
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace pg

using namespace pg;

extern "C" int pg_norm_stats(const float* y, int32_t N, int64_t L, double* sums, void* stream) {
  PG_REQUIRE(y && sums && N > 0 && L > 0, "pg_norm_stats: bad arguments");
  PG_REQUIRE(L % 4 == 0, "pg_norm_stats: per-sample length must be a multiple of 4 (float4 path)");
  hipLaunchKernelGGL(norm_stats_kernel, dim3(norm_blocks(L), N), dim3(256), 0, (hipStream_t)stream, y, (long)L, sums);
  PG_LAUNCH_OK("pg_norm_stats");
  return 0;
}

extern "C" int pg_norm_finalize(const double* sums, const float* gamma, const float* beta, int32_t N, int64_t L,
                                float eps, float* mr, float* aff, void* stream) {
  PG_REQUIRE(sums && gamma && beta && mr && aff && N > 0, "pg_norm_finalize: bad arguments");
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, gamma, beta, N,
                     (long)L, eps, mr, aff);
  PG_LAUNCH_OK("pg_norm_finalize");
  return 0;
}

// io_flags: bit 0 = dz is bf16, bit 1 = y is bf16 (bf16 STORAGE on the bf16 data path; sums stay double)
extern "C" int pg_norm_bwd_reduce_ex(const void* dz, const void* y, const float* mr, int32_t N, int64_t L, double* bsums,
                                     int32_t io_flags, void* stream) {
  PG_REQUIRE(dz && y && mr && bsums && N > 0 && L > 0 && L % 4 == 0 && io_flags >= 0 && io_flags <= 3, "pg_norm_bwd_reduce: bad arguments");
  const dim3 grid(norm_blocks_bwd(L, N), N);
  typedef unsigned short bf;
  hipStream_t st = (hipStream_t)stream;
  switch (io_flags) {
    case 0: hipLaunchKernelGGL((norm_bwd_reduce_kernel<float, float>), grid, dim3(256), 0, st, (const float*)dz, (const float*)y, mr, (long)L, bsums); break;
    case 1: hipLaunchKernelGGL((norm_bwd_reduce_kernel<bf, float>), grid, dim3(256), 0, st, (const bf*)dz, (const float*)y, mr, (long)L, bsums); break;
    case 2: hipLaunchKernelGGL((norm_bwd_reduce_kernel<float, bf>), grid, dim3(256), 0, st, (const float*)dz, (const bf*)y, mr, (long)L, bsums); break;
    default: hipLaunchKernelGGL((norm_bwd_reduce_kernel<bf, bf>), grid, dim3(256), 0, st, (const bf*)dz, (const bf*)y, mr, (long)L, bsums); break;
  }
  PG_LAUNCH_OK("pg_norm_bwd_reduce");
  return 0;
}
extern "C" int pg_norm_bwd_reduce(const float* dz, const float* y, const float* mr, int32_t N, int64_t L,
                                  double* bsums, void* stream) {
  return pg_norm_bwd_reduce_ex(dz, y, mr, N, L, bsums, 0, stream);
}

// io_flags as in pg_norm_bwd_reduce_ex; with a bf16 dz the in-place result IS the bf16 operand of the layer's gradient
// contractions (dy_bf16 = NULL then)
extern "C" int pg_norm_bwd_apply_io(void* dz, const void* y, const float* mr, const double* bsums, const float* gamma,
                                    int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, int32_t io_flags,
                                    void* stream) {
  PG_REQUIRE(dz && y && mr && bsums && gamma && N > 0 && L > 0 && L % 4 == 0 && io_flags >= 0 && io_flags <= 3, "pg_norm_bwd_apply: bad arguments");
  const dim3 grid(norm_blocks(L), N);
  typedef unsigned short bf;
  hipStream_t st = (hipStream_t)stream;
  switch (io_flags) {
    case 0: hipLaunchKernelGGL((norm_bwd_apply_kernel<float, float>), grid, dim3(256), 0, st, (float*)dz, (const float*)y, mr, bsums, gamma, N, (long)L, dgamma, dbeta, dy_bf16); break;
    case 1: hipLaunchKernelGGL((norm_bwd_apply_kernel<bf, float>), grid, dim3(256), 0, st, (bf*)dz, (const float*)y, mr, bsums, gamma, N, (long)L, dgamma, dbeta, dy_bf16); break;
    case 2: hipLaunchKernelGGL((norm_bwd_apply_kernel<float, bf>), grid, dim3(256), 0, st, (float*)dz, (const bf*)y, mr, bsums, gamma, N, (long)L, dgamma, dbeta, dy_bf16); break;
    default: hipLaunchKernelGGL((norm_bwd_apply_kernel<bf, bf>), grid, dim3(256), 0, st, (bf*)dz, (const bf*)y, mr, bsums, gamma, N, (long)L, dgamma, dbeta, dy_bf16); break;
  }
  PG_LAUNCH_OK("pg_norm_bwd_apply");
  return 0;
}
extern "C" int pg_norm_bwd_apply_ex(float* dz, const float* y, const float* mr, const double* bsums, const float* gamma,
                                    int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, void* stream) {
  return pg_norm_bwd_apply_io(dz, y, mr, bsums, gamma, N, L, dgamma, dbeta, dy_bf16, 0, stream);
}

extern "C" int pg_norm_bwd_apply(float* dz, const float* y, const float* mr, const double* bsums, const float* gamma,
                                 int32_t N, int64_t L, float* dgamma, float* dbeta, void* stream) {
  return pg_norm_bwd_apply_ex(dz, y, mr, bsums, gamma, N, L, dgamma, dbeta, nullptr, stream);
}
