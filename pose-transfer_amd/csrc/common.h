// Shared helpers for libposegan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <tuple>
#include <type_traits>
#include <utility>
#include "posegan_hip.h"

namespace pg {

char* err_buf();  // thread-local 512-byte message buffer (api.hip)
int& last_info();  // thread-local: tile config / loader modes / split-K of the last pg_conv / pg_conv_wgrad launch

#define PG_FAIL(code, ...)                          \
  do {                                              \
    snprintf(pg::err_buf(), 512, __VA_ARGS__);      \
    return (code);                                  \
  } while (0)

#define PG_REQUIRE(cond, ...)                       \
  do {                                              \
    if (!(cond)) PG_FAIL(1, __VA_ARGS__);           \
  } while (0)

#define PG_LAUNCH_OK(what)                                                        \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) PG_FAIL(2, "%s: %s", what, hipGetErrorString(e__));    \
  } while (0)

#define PG_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e__ = (call);                                                      \
    if (e__ != hipSuccess) PG_FAIL(3, "%s: %s", #call, hipGetErrorString(e__));   \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Ablation / debugging switches of the host-side launch code, read from the environment ONCE per process (round 3 called
// getenv ~14 times per 256-row launch).  Every switch selects another CORRECT code path; the timing experiments that produce
// wrong results exist only in builds with -DPG_TIMING_EXPERIMENTS (tools/conv_timeline.py).  Two switches stay dynamic because
// the test-suite flips them inside one process: PG_FORCE_BF16_BIG, PG_BIG_128_VARIANT (igemm_conv.hip) and PG_NO_OUT_DGRAD_MFMA
// (out_conv_dgrad.hip).
struct Env {
  bool no_vec_epilogue, splitk_debug, no_splitk_ws, no_dma, no_xcd_swizzle, conv_mask_generic, no_bf16_big, no_bf16_big64;
  bool wg_generic, bias_grad_x4, nn_loss_v1;
  unsigned debug_bits;      // PG_TIMING_EXPERIMENTS builds only: bits of ConvK::xcd_swizzle (igemm_bf16.hip)
  Env() {
    auto on = [](const char* n) { return getenv(n) != nullptr; };
    no_vec_epilogue = on("PG_NO_VEC_EPILOGUE"); splitk_debug = on("PG_SPLITK_DEBUG"); no_splitk_ws = on("PG_NO_SPLITK_WS");
    no_dma = on("PG_NO_DMA"); no_xcd_swizzle = on("PG_NO_XCD_SWIZZLE"); conv_mask_generic = on("PG_CONV_MASK_GENERIC");
    no_bf16_big = on("PG_NO_BF16_BIG"); no_bf16_big64 = on("PG_NO_BF16_BIG64");
    wg_generic = on("PG_WG_GENERIC"); bias_grad_x4 = on("PG_BIAS_GRAD_X4"); nn_loss_v1 = on("PG_NN_LOSS_V1");
    debug_bits = 0;
#ifdef PG_TIMING_EXPERIMENTS
    const char* names[] = {"PG_DEBUG_OPERAND_A", "PG_DEBUG_OPERAND_B", "PG_DEBUG_ONE_KTILE", "PG_DEBUG_CONV_TIMELINE",
                           "PG_DEBUG_EPI_NOFWD", "PG_DEBUG_EPI_NOSTORE", "PG_DEBUG_EPI_NOACC", "PG_DEBUG_NO_KBARRIER",
                           "PG_DEBUG_NO_KDMA", "PG_DEBUG_A_EVERY_4TH", "PG_DEBUG_HALF_A_FETCH", "PG_DEBUG_NO_FETCH", "PG_DEBUG_A_EVERY_2ND"};
    for (int i = 0; i < 13; ++i) if (on(names[i])) debug_bits |= 2u << i;
#endif
  }
};
inline const Env& env() { static const Env e; return e; }
// Integer tuning knob: read from the environment ONCE (the caller keeps the value in a function-local static) — unless the process was
// started with PG_DYN_ENV=1, in which case every launch re-reads it, so that tools/quad_inproc_ab.py can alternate the arms of an A/B
// inside one process (box-to-box and process-to-process drift cancel).  Every value selects a correct path.
inline bool dyn_env() { static const bool d = getenv("PG_DYN_ENV") != nullptr; return d; }
inline int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#define PG_ENV_INT(var, name, dflt)              \
  static const int var##_static = pg::env_int(name, dflt); \
  const int var = pg::dyn_env() ? pg::env_int(name, dflt) : var##_static

// PG_DETERMINISTIC (round 5; pg_set_deterministic / the environment variable at load time): every launch path whose result
// depends on an arrival order is replaced by an ordered one — split-K contractions only through the workspace + fix-up pass (or
// un-split), weight gradients un-split (no float atomics on dW), bias / first-layer reductions by ONE workgroup per output, losses by
// one workgroup, the warp backward without its float-atomic scatter fall-back... slower, bit-repeatable run to run.  What stays:
// DOUBLE atomics on the per-sample statistics (sums of fp32-accurate partials whose order moves the 53-bit result in its last
// bits: invisible after the conversion to fp32 except with probability ~1e-8 per sum).  Host-side flag.
int& deterministic_flag();
inline bool deterministic() { return deterministic_flag() != 0; }

// ---- launch tape (round 3, api.hip: pg_tape_*).  Every kernel launch / memset / stream-ordering call of the library goes through
// these macros: it is issued as usual and, while the calling thread records, also kept as a closure (kernel, geometry, stream
// and a COPY of the argument block).  pg_tape_replay re-issues the closures: one C call per training iteration instead of
// ~250 Python -> ctypes -> descriptor-check -> launch round trips (25 us each; a replayed launch costs the hipLaunchKernel
// alone).  Unlike a HIP graph the tape keeps the launches on their own streams (weight-gradient side stream, collectives).
struct Tape;
Tape* tape_recording();                                  // the tape this thread is recording into, or nullptr
void tape_push(Tape* t, void (*thunk)(void*), void* closure, void (*del)(void*), const char* where, int line);
template <typename F>
inline void tape_record(F&& f, const char* where = "", int line = 0) {
  Tape* t = tape_recording();
  if (t == nullptr) return;
  typedef typename std::decay<F>::type Fn;
  Fn* c = new Fn(std::forward<F>(f));
  tape_push(t, [](void* q) { (*static_cast<Fn*>(q))(); }, c, [](void* q) { delete static_cast<Fn*>(q); }, where, line);
}
// Every argument expression is evaluated ONCE, here: the closure holds values (a launch written as `dim3(bx, d->N)` must not
// dereference the caller's descriptor again at replay time — it is gone by then).
#define PG_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                                              \
  do {                                                                                                                    \
    const dim3 pg_g__ = (grid), pg_b__ = (block);                                                                         \
    const size_t pg_sh__ = (size_t)(shmem);                                                                               \
    hipStream_t pg_st__ = (stream);                                                                                       \
    auto pg_args__ = std::make_tuple(__VA_ARGS__);                                                                        \
    auto pg_fn__ = [=]() {                                                                                                \
      std::apply([&](const auto&... pg_a__) { hipLaunchKernelGGL(kernel, pg_g__, pg_b__, pg_sh__, pg_st__, pg_a__...); }, \
                 pg_args__);                                                                                              \
    };                                                                                                                    \
    pg_fn__();                                                                                                            \
    if (pg::tape_recording() != nullptr) pg::tape_record(pg_fn__, __FILE__, __LINE__);                                    \
  } while (0)
#define PG_MEMSET_ASYNC(ptr, val, bytes, st)                                                        \
  do {                                                                                              \
    void* pg_p__ = (void*)(ptr); const size_t pg_b__ = (bytes); hipStream_t pg_s__ = (st);          \
    const int pg_v__ = (val);                                                                       \
    PG_HIP(hipMemsetAsync(pg_p__, pg_v__, pg_b__, pg_s__));                                         \
    if (pg::tape_recording() != nullptr)                                                            \
      pg::tape_record([=]() { (void)hipMemsetAsync(pg_p__, pg_v__, pg_b__, pg_s__); }, __FILE__, __LINE__);              \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave-level (64 lanes) sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-level sum of `v` over 256 threads; result valid in thread 0. `red` = 4-float LDS scratch.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == PG_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == PG_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
  return v;
}
// branch-free forms: slope = 0 (ReLU), 0.2 (LeakyReLU), 1 (identity); exact for every finite input
// 128-bit load from GLOBAL memory at a wave-uniform base + per-lane byte offset.  The explicit address space keeps the
// compiler from emitting FLAT loads when the base is a select of two pointers (FLAT counts on lgkmcnt as well and
// would disturb the counted LDS waits); readfirstlane keeps the base in SGPRs (saddr form, no 64-bit VALU adds).
__device__ __forceinline__ const char* uniform_ptr(const char* q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float4 ldg128(const char* base, unsigned off) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x4g __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) const f32x4g gvec;
  const f32x4g v = *reinterpret_cast<gvec*>(reinterpret_cast<unsigned long long>(base) + off);
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  (void)base; (void)off;
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
__device__ __forceinline__ float ldg32(const char* base, long off) {      // per-lane base allowed
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(1))) const float gf;
  return *reinterpret_cast<gf*>(reinterpret_cast<unsigned long long>(base) + (unsigned long long)off);
#else
  (void)base; (void)off;
  return 0.f;
#endif
}
__device__ __forceinline__ float2 ldg64(const char* base, unsigned off) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x2g __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) const f32x2g gvec;
  const f32x2g v = *reinterpret_cast<gvec*>(reinterpret_cast<unsigned long long>(base) + off);
  return make_float2(v[0], v[1]);
#else
  (void)base; (void)off;
  return make_float2(0.f, 0.f);
#endif
}


__device__ __forceinline__ float act_slope(int act) { return act == PG_ACT_RELU ? 0.f : (act == PG_ACT_LEAKY ? 0.2f : 1.f); }
__device__ __forceinline__ float apply_act_s(float v, float slope) { return fmaf(slope, fminf(v, 0.f), fmaxf(v, 0.f)); }
__device__ __forceinline__ float act_grad_s(float z, float slope) { return z > 0.f ? 1.f : slope; }
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == PG_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == PG_ACT_LEAKY) return z > 0.f ? 1.f : 0.2f;
  return 1.f;
}

// The per-sample affine of a norm layer whose statistics have just been completed by the producing convolution (round 4:
// pg_norm_finalize folded into the first reader — materialise_bf16_kernel, optim.hip; out_conv_fwd_kernel, out_conv_fwd.hip).
struct NormFold {
  const double* sums;       // [N][PG_STAT_SLOTS][2] or null
  const float* gamma; const float* beta;
  long L; float eps;
  float* mr; float* aff;
};
// norm_finalize_kernel's arithmetic for sample n; `publish`: this thread writes (mean, rstd) and (a, b) for the later readers
__device__ __forceinline__ void norm_fold_affine(const NormFold& nf, int n, bool publish, float& a, float& b) {
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < PG_STAT_SLOTS; ++k) { s1 += nf.sums[((long)n * PG_STAT_SLOTS + k) * 2]; s2 += nf.sums[((long)n * PG_STAT_SLOTS + k) * 2 + 1]; }
  const double mean = s1 / (double)nf.L;
  double var = s2 / (double)nf.L - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)nf.eps);
  const double g = (double)nf.gamma[0], bt = (double)nf.beta[0];
  a = (float)(g * rstd); b = (float)(bt - g * mean * rstd);
  if (publish) {
    nf.mr[2 * n] = (float)mean; nf.mr[2 * n + 1] = (float)rstd;
    nf.aff[2 * n] = a; nf.aff[2 * n + 1] = b;
  }
}

}  // namespace pg
