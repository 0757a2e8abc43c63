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
// One 16-byte chunk of an fp32 (T = float: 4 elements) or bf16 (T = unsigned short: 8 elements; bf16 STORAGE, round 3) tensor
template <typename T> struct Chunk { static constexpr int N = 16 / (int)sizeof(T); };
template <typename T>
__device__ __forceinline__ void ldc(const T* base, long ic, float (&v)[Chunk<T>::N]) {
  const uint4 u = reinterpret_cast<const uint4*>(base)[ic];
  if constexpr (sizeof(T) == 4) {
    v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
  } else {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
  }
}
template <typename T>
__device__ __forceinline__ void stc(T* base, long ic, const float (&v)[Chunk<T>::N]) {
  uint4 u;
  if constexpr (sizeof(T) == 4) {
    u = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
  } else {
    u = make_uint4(pack_bf16_rne(v[0], v[1]), pack_bf16_rne(v[2], v[3]), pack_bf16_rne(v[4], v[5]), pack_bf16_rne(v[6], v[7]));
  }
  reinterpret_cast<uint4*>(base)[ic] = u;
}
template <typename T>
__device__ __forceinline__ float ld1(const T* base, long i) {
  if constexpr (sizeof(T) == 4) return base[i];
  else return __uint_as_float((unsigned)base[i] << 16);
}

// dz and y share the storage type (fp32 or bf16): every lane moves 16-byte chunks of both
// sums_mode 2 recovers sum(r * xhat) as (S2 - beta S1) / gamma from the ACTIVATED, bf16-rounded operand: ill-conditioned when
// |gamma| is small against |beta| (and undefined at gamma == 0).  Such a layer takes the plain reduce pass over (dz, y) instead:
// pg_norm_bwd_reduce_guard runs this kernel with `guard` = (gamma, beta) and exits at once when the layer is well-conditioned;
// the apply kernel evaluates the same predicate and reads whichever sums are valid.
__device__ __forceinline__ bool norm_gamma_small(float g, float b) { return fabsf(g) < 1e-3f + 0.02f * fabsf(b); }

template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const T* dz, const T* y, const float* mr, long L,
                                                              double* bsums, const float* guard_gamma = nullptr,
                                                              const float* guard_beta = nullptr) {
  constexpr int V = Chunk<T>::N;
  __shared__ double red[8];
  if (guard_gamma != nullptr && !norm_gamma_small(guard_gamma[0], guard_beta[0])) return;
  const int n = blockIdx.y;
  const float mean = mr[2 * n], rstd = mr[2 * n + 1];
  const T* bd = dz + (long)n * L;
  const T* by = y + (long)n * L;
  float s = 0.f, q = 0.f;
  const long LC = L / V;
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < LC; i += 4 * stride) {           // eight independent 16-byte loads in flight per lane
    float d[4][V], v[4][V];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ldc(bd, i + u * stride, d[u]); ldc(by, i + u * stride, v[u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < V; ++e) { s += d[u][e]; q += d[u][e] * ((v[u][e] - mean) * rstd); }
  }
  for (; i < LC; i += stride) {
    float d[V], v[V];
    ldc(bd, i, d); ldc(by, i, v);
#pragma unroll
    for (int e = 0; e < V; ++e) { s += d[e]; q += d[e] * ((v[e] - mean) * rstd); }
  }
  if (blockIdx.x == 0)
    for (long i = LC * V + threadIdx.x; i < L; i += 256) { s += ld1(bd, i); q += ld1(bd, i) * ((ld1(by, i) - mean) * rstd); }
  double ds = wave_sum_d((double)s), dq = wave_sum_d((double)q);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w] = ds; red[4 + w] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&bsums[2 * n], red[0] + red[1] + red[2] + red[3]);
    atomicAdd(&bsums[2 * n + 1], red[4] + red[5] + red[6] + red[7]);
  }
}

// The two per-sample sums (sum dz, sum dz * xhat) of sample n.  MODE 0: bsums [N][2] from norm_bwd_reduce_kernel.  MODE 1 / 2
// (round 4): [N][PG_STAT_SLOTS][2] accumulated by the epilogue that wrote dz — (sum r, sum r * y_raw) or (sum r, sum r * x_act);
// see pg_norm_bwd_apply_v2 in include/posegan_hip.h.
template <int MODE>
__device__ __forceinline__ void bwd_sums(const double* bsums, int n, float mean, float rstd, float g, float b, double& s1, double& s2,
                                         const double* bsums0 = nullptr) {
  if constexpr (MODE == 0) { s1 = bsums[2 * n]; s2 = bsums[2 * n + 1]; return; }
  if constexpr (MODE == 2)
    if (bsums0 != nullptr && norm_gamma_small(g, b)) { s1 = bsums0[2 * n]; s2 = bsums0[2 * n + 1]; return; }
  double a1 = 0.0, a2 = 0.0;
  for (int k = 0; k < PG_STAT_SLOTS; ++k) { a1 += bsums[((long)n * PG_STAT_SLOTS + k) * 2]; a2 += bsums[((long)n * PG_STAT_SLOTS + k) * 2 + 1]; }
  s1 = a1;
  if constexpr (MODE == 1) s2 = (double)rstd * (a2 - (double)mean * a1);
  else s2 = (g != 0.f) ? (a2 - (double)b * a1) / (double)g : 0.0;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(T* dz, const T* y, const float* mr,
                                                             const double* bsums, const float* gamma, const float* beta, int N, long L,
                                                             float* dgamma, float* dbeta, unsigned short* dy_bf16,
                                                             const double* bsums0) {
  constexpr int V = Chunk<T>::N;
  const int n = blockIdx.y;
  const float mean = mr[2 * n], rstd = mr[2 * n + 1];
  const float g = gamma[0];
  const float bta = (MODE == 2) ? beta[0] : 0.f;
  double s1d, s2d;
  bwd_sums<MODE>(bsums, n, mean, rstd, g, bta, s1d, s2d, bsums0);
  const float m1 = (float)(s1d / (double)L);
  const float m2 = (float)(s2d / (double)L);
  const float k = g * rstd;
  T* bd = dz + (long)n * L;
  const T* by = y + (long)n * L;
  const long LC = L / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < LC; i += (long)gridDim.x * 256) {
    float d[V], v[V];
    ldc(bd, i, d); ldc(by, i, v);
#pragma unroll
    for (int e = 0; e < V; ++e) d[e] = k * (d[e] - m1 - ((v[e] - mean) * rstd) * m2);
    stc(bd, i, d);
    if constexpr (sizeof(T) == 4) {
      if (dy_bf16) {         // the bf16 operand copy the data- / weight-gradient contractions read (fp32 storage on the bf16 data path)
        uint2 pk;
        pk.x = pack_bf16_rne(d[0], d[1]); pk.y = pack_bf16_rne(d[2], d[3]);
        reinterpret_cast<uint2*>(dy_bf16 + (long)n * L)[i] = pk;
      }
    }
  }
  if (blockIdx.x == 0)
    for (long i = LC * V + threadIdx.x; i < L; i += 256) {
      const float r = k * (ld1(bd, i) - m1 - ((ld1(by, i) - mean) * rstd) * m2);
      if constexpr (sizeof(T) == 4) bd[i] = r;
      else bd[i] = (unsigned short)(pack_bf16_rne(r, 0.f) & 0xffffu);
      if (sizeof(T) == 4 && dy_bf16) dy_bf16[(long)n * L + i] = (unsigned short)(pack_bf16_rne(r, 0.f) & 0xffffu);
    }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    double sg = 0.0, sb = 0.0;
    for (int i = 0; i < N; ++i) {
      double a1, a2;
      bwd_sums<MODE>(bsums, i, mr[2 * i], mr[2 * i + 1], g, bta, a1, a2, bsums0);
      sb += a1; sg += a2;
    }
    if (dgamma) atomicAdd(dgamma, (float)sg);
    if (dbeta) atomicAdd(dbeta, (float)sb);
  }
}

// norm_bwd_reduce ends every workgroup with two double atomics on ONE 64-byte line (bsums [N][2]); same-address
// atomics serialise (measured on the bias gradient: ~90 ns per workgroup), so that kernel gets few, deep workgroups
static int norm_blocks_bwd(long L, int N, int vec = 4) {
  long b = (L / vec + 256 * 16 - 1) / (256 * 16);
  const long cap = N >= 4 ? 64 : N == 3 ? 96 : N == 2 ? 128 : 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

static int norm_blocks(long L, int vec = 4) {
  // streaming passes on this chip prefer MANY workgroups with one 16-byte chunk per lane over few deep ones (swept at batch
  // 32: 8 chunks per lane / 512 workgroups per sample 2.04 ms, 1 / 8192: 1.87 ms for the norm-backward apply passes of a step)
  static const int per = getenv("PG_NORM_PER") ? atoi(getenv("PG_NORM_PER")) : 1;
  static const int cap = getenv("PG_NORM_CAP") ? atoi(getenv("PG_NORM_CAP")) : 8192;
  long b = (L / vec + 256 * per - 1) / (256 * per);
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace pg

using namespace pg;

extern "C" int pg_norm_stats(const float* y, int32_t N, int64_t L, double* sums, void* stream) {
  PG_REQUIRE(y && sums && N > 0 && L > 0, "pg_norm_stats: bad arguments");
  PG_REQUIRE(L % 4 == 0, "pg_norm_stats: per-sample length must be a multiple of 4 (float4 path)");
  PG_KLAUNCH(norm_stats_kernel, dim3(norm_blocks(L), N), dim3(256), 0, (hipStream_t)stream, y, (long)L, sums);
  PG_LAUNCH_OK("pg_norm_stats");
  return 0;
}

extern "C" int pg_norm_finalize(const double* sums, const float* gamma, const float* beta, int32_t N, int64_t L,
                                float eps, float* mr, float* aff, void* stream) {
  PG_REQUIRE(sums && gamma && beta && mr && aff && N > 0, "pg_norm_finalize: bad arguments");
  PG_KLAUNCH(norm_finalize_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, gamma, beta, N,
                     (long)L, eps, mr, aff);
  PG_LAUNCH_OK("pg_norm_finalize");
  return 0;
}

// io_flags: 0 = dz and y fp32, 3 = both bf16 (bf16 STORAGE on the bf16 data path; sums stay double)
extern "C" int pg_norm_bwd_reduce_ex(const void* dz, const void* y, const float* mr, int32_t N, int64_t L, double* bsums,
                                     int32_t io_flags, void* stream) {
  PG_REQUIRE(dz && y && mr && bsums && N > 0 && L > 0 && ((io_flags == 0 && L % 4 == 0) || (io_flags == 3 && L % 8 == 0)),
             "pg_norm_bwd_reduce: bad arguments (dz and y share the storage type; L %% 4 == 0 fp32 / L %% 8 == 0 bf16)");
  typedef unsigned short bf;
  hipStream_t st = (hipStream_t)stream;
  if (io_flags == 0)
    PG_KLAUNCH((norm_bwd_reduce_kernel<float>), dim3(norm_blocks_bwd(L, N, 4), N), dim3(256), 0, st, (const float*)dz, (const float*)y, mr, (long)L, bsums,
               (const float*)nullptr, (const float*)nullptr);
  else
    PG_KLAUNCH((norm_bwd_reduce_kernel<bf>), dim3(norm_blocks_bwd(L, N, 8), N), dim3(256), 0, st, (const bf*)dz, (const bf*)y, mr, (long)L, bsums,
               (const float*)nullptr, (const float*)nullptr);
  PG_LAUNCH_OK("pg_norm_bwd_reduce");
  return 0;
}
// the reduce pass of a layer whose sums the producer of dz already wrote in activated-operand form (sums_mode 2): runs only when
// the layer's gamma is too small for that form (decided on the device: no host synchronisation); see norm_gamma_small
extern "C" int pg_norm_bwd_reduce_guard(const void* dz, const void* y, const float* mr, const float* gamma, const float* beta,
                                        int32_t N, int64_t L, double* bsums, int32_t io_flags, void* stream) {
  PG_REQUIRE(dz && y && mr && bsums && gamma && beta && N > 0 && L > 0 && ((io_flags == 0 && L % 4 == 0) || (io_flags == 3 && L % 8 == 0)),
             "pg_norm_bwd_reduce_guard: bad arguments");
  typedef unsigned short bf;
  hipStream_t st = (hipStream_t)stream;
  if (io_flags == 0)
    PG_KLAUNCH((norm_bwd_reduce_kernel<float>), dim3(norm_blocks_bwd(L, N, 4), N), dim3(256), 0, st, (const float*)dz, (const float*)y, mr, (long)L, bsums, gamma, beta);
  else
    PG_KLAUNCH((norm_bwd_reduce_kernel<bf>), dim3(norm_blocks_bwd(L, N, 8), N), dim3(256), 0, st, (const bf*)dz, (const bf*)y, mr, (long)L, bsums, gamma, beta);
  PG_LAUNCH_OK("pg_norm_bwd_reduce_guard");
  return 0;
}
extern "C" int pg_norm_bwd_reduce(const float* dz, const float* y, const float* mr, int32_t N, int64_t L,
                                  double* bsums, void* stream) {
  return pg_norm_bwd_reduce_ex(dz, y, mr, N, L, bsums, 0, stream);
}

// io_flags as in pg_norm_bwd_reduce_ex; with bf16 storage the in-place result IS the bf16 operand of the layer's gradient
// contractions (dy_bf16 must be NULL then).  sums_mode: where the two per-sample sums come from (include/posegan_hip.h).
extern "C" int pg_norm_bwd_apply_v3(void* dz, const void* y, const float* mr, const double* bsums, const float* gamma,
                                    const float* beta, int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16,
                                    int32_t io_flags, int32_t sums_mode, const double* bsums_guard, void* stream);
extern "C" int pg_norm_bwd_apply_v2(void* dz, const void* y, const float* mr, const double* bsums, const float* gamma,
                                    const float* beta, int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16,
                                    int32_t io_flags, int32_t sums_mode, void* stream) {
  return pg_norm_bwd_apply_v3(dz, y, mr, bsums, gamma, beta, N, L, dgamma, dbeta, dy_bf16, io_flags, sums_mode, nullptr, stream);
}
// bsums_guard (sums_mode 2 only, optional): [N][2] sums of pg_norm_bwd_reduce_guard — read instead of `bsums` when the layer's gamma
// is too small for the activated-operand form
extern "C" int pg_norm_bwd_apply_v3(void* dz, const void* y, const float* mr, const double* bsums, const float* gamma,
                                    const float* beta, int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16,
                                    int32_t io_flags, int32_t sums_mode, const double* bsums_guard, void* stream) {
  PG_REQUIRE(dz && y && mr && bsums && gamma && N > 0 && L > 0 && L % 4 == 0 && (io_flags == 0 || (io_flags == 3 && L % 8 == 0 && !dy_bf16)) &&
             sums_mode >= 0 && sums_mode <= 2 && (sums_mode != 2 || beta != nullptr), "pg_norm_bwd_apply: bad arguments");
  typedef unsigned short bf;
  hipStream_t st = (hipStream_t)stream;
#define PG_NBA(TT, VEC, MODE)                                                                                                 \
  PG_KLAUNCH((norm_bwd_apply_kernel<TT, MODE>), dim3(norm_blocks(L, VEC), N), dim3(256), 0, st, (TT*)dz, (const TT*)y, mr, bsums, \
             gamma, beta, N, (long)L, dgamma, dbeta, dy_bf16, sums_mode == 2 ? bsums_guard : (const double*)nullptr)
  if (io_flags == 0) {
    if (sums_mode == 0) PG_NBA(float, 4, 0); else if (sums_mode == 1) PG_NBA(float, 4, 1); else PG_NBA(float, 4, 2);
  } else {
    if (sums_mode == 0) PG_NBA(bf, 8, 0); else if (sums_mode == 1) PG_NBA(bf, 8, 1); else PG_NBA(bf, 8, 2);
  }
#undef PG_NBA
  PG_LAUNCH_OK("pg_norm_bwd_apply");
  return 0;
}
extern "C" int pg_norm_bwd_apply_io(void* dz, const void* y, const float* mr, const double* bsums, const float* gamma,
                                    int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, int32_t io_flags,
                                    void* stream) {
  return pg_norm_bwd_apply_v2(dz, y, mr, bsums, gamma, nullptr, N, L, dgamma, dbeta, dy_bf16, io_flags, 0, stream);
}
extern "C" int pg_norm_bwd_apply_ex(float* dz, const float* y, const float* mr, const double* bsums, const float* gamma,
                                    int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, void* stream) {
  return pg_norm_bwd_apply_io(dz, y, mr, bsums, gamma, N, L, dgamma, dbeta, dy_bf16, 0, stream);
}

extern "C" int pg_norm_bwd_apply(float* dz, const float* y, const float* mr, const double* bsums, const float* gamma,
                                 int32_t N, int64_t L, float* dgamma, float* dbeta, void* stream) {
  return pg_norm_bwd_apply_ex(dz, y, mr, bsums, gamma, N, L, dgamma, dbeta, nullptr, stream);
}
