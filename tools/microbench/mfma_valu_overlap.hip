// Micro-benchmark: what does one non-MFMA instruction cost a wave that is otherwise issuing back-to-back fp32 MFMAs
// (v_mfma_f32_32x32x2_f32, 64 cycles each)?  One workgroup of 4 waves per CU (blocks=256) or two (blocks=512).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o tools/microbench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { K_NONE, K_VALU, K_PKFMA, K_SALU, K_DSR32, K_DSR128, K_DSW128, K_GLD128, K_DMA128 };
template <int KIND, int PER>   // PER extra instructions of KIND after every MFMA
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, float s) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + s;
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = a + i;
  f32x2 pv[4]; for (int i = 0; i < 4; ++i) pv[i] = (f32x2){a, b};
  f32x4 q4[4]; for (int i = 0; i < 4; ++i) q4[i] = (f32x4){a, b, a, b};
  int sc = iters;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = a;
  __syncthreads();
  const unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16u;
  const f32x4* gp = reinterpret_cast<const f32x4*>(in) + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int id = (m * PER + q);
        if (KIND == K_VALU) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[id & 7]) : "v"(b));
        if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[id & 3]) : "v"(pv[(id + 1) & 3]));
        if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
        if (KIND == K_DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(v[id & 7]) : "v"(la));
        if (KIND == K_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[id & 3]) : "v"(la));
        if (KIND == K_DSW128) asm volatile("ds_write_b128 %0, %1" ::"v"(la), "v"(q4[id & 3]));
        if (KIND == K_GLD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q4[id & 3]) : "v"(gp));
        if (KIND == K_DMA128)     // global -> LDS without a register round trip (1 KB per wave instruction)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(gp), lds + (threadIdx.x >> 6) * 1024 + (id & 3) * 256, 16, 0, 0);
      }
    }
    if (KIND == K_DSR32 || KIND == K_DSR128 || KIND == K_DSW128) asm volatile("s_waitcnt lgkmcnt(0)");
    if (KIND == K_GLD128 || KIND == K_DMA128) asm volatile("s_waitcnt vmcnt(0)");
  }
  float r = sc; for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) r += acc[i][q];
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += pv[i][0] + pv[i][1] + q4[i][0] + q4[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
static double base[2];
template <int KIND, int PER> void run(float* d, const float* in, int bi, const char* tag) {
  const int blocks = bi ? 512 : 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((k<KIND, PER>), dim3(blocks), dim3(256), 0, 0, d, in, 100, 0.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, PER>), dim3(blocks), dim3(256), 0, 0, d, in, iters, 0.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_group = ms * 1e6 / iters / (bi ? 2 : 1);     // per 16 MFMAs of ONE wave
  if (KIND == K_NONE) base[bi] = ns_per_group;
  const double extra = PER ? (ns_per_group - base[bi]) / (16.0 * PER) : 0.0;
  printf("%-22s waves/SIMD=%d  %7.1f ns per 16 MFMAs  | +%.2f ns (~%.1f clk @2.4GHz) per extra instruction\n", tag,
         bi + 1, ns_per_group, extra, extra * 2.4);
}
int main() {
  float *d, *in; (void)hipMalloc(&d, 1024 * 256 * 4); (void)hipMalloc(&in, 1 << 20); (void)hipMemset(in, 0, 1 << 20);
  for (int bi = 0; bi < 2; ++bi) {
    run<K_NONE, 0>(d, in, bi, "mfma only");
    run<K_VALU, 2>(d, in, bi, "v_fma_f32 x2");
    run<K_VALU, 6>(d, in, bi, "v_fma_f32 x6");
    run<K_PKFMA, 2>(d, in, bi, "v_pk_fma_f32 x2");
    run<K_SALU, 4>(d, in, bi, "s_add_u32 x4");
    run<K_DSR32, 1>(d, in, bi, "ds_read_b32 x1");
    run<K_DSR128, 1>(d, in, bi, "ds_read_b128 x1");
    run<K_DSW128, 1>(d, in, bi, "ds_write_b128 x1");
    run<K_GLD128, 1>(d, in, bi, "global_load_dwordx4 x1");
    run<K_DMA128, 1>(d, in, bi, "global_load_lds_x4 x1");
  }
  return 0;
}
