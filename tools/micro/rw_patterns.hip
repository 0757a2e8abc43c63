// Micro-benchmark: the in-place read-modify-write streaming pattern of norm backward (bf16: read dz + y, write dz = 6 B / element) and
// the one-in / one-out pattern of the materialisation pass, U chunks of 16 bytes per lane with ALL loads issued before the first use.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rw_patterns.hip -o /tmp/rw_patterns && /tmp/rw_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint4 mix(uint4 a, uint4 b) { return make_uint4(a.x + b.y, a.y ^ b.z, a.z + b.w, a.w ^ b.x); }

// MODE 0: a[i] = f(a[i], b[i]) (in place);  1: c[i] = f(a[i]) (copy);  2: a[i] = f(a[i], b[i]) and c[i] = g (two outputs)
template <int U, int MODE, bool STRIDED>
__global__ __launch_bounds__(256) void k(uint4* a, const uint4* b, uint4* c, long n) {
  // STRIDED: the U chunks of a lane are 256 chunks apart (every instruction of the wave is one contiguous 1 KB run); else consecutive
  const long base = (long)blockIdx.x * 256 * U;
  uint4 va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = STRIDED ? base + u * 256 + threadIdx.x : base + (long)threadIdx.x * U + u;
    if (i < n) { va[u] = a[i]; if (MODE != 1) vb[u] = b[i]; }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = STRIDED ? base + u * 256 + threadIdx.x : base + (long)threadIdx.x * U + u;
    if (i < n) {
      const uint4 r = MODE == 1 ? mix(va[u], va[u]) : mix(va[u], vb[u]);
      if (MODE == 1) c[i] = r; else a[i] = r;
      if (MODE == 2) c[i] = va[u];
    }
  }
}
template <int U, int MODE, bool STRIDED>
void run(const char* name, uint4* a, uint4* b, uint4* c, long n, double bytes) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned wgs = (unsigned)((n + 256 * U - 1) / (256 * U));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<U, MODE, STRIDED>), dim3(wgs), dim3(256), 0, 0, a, b, c, n);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<U, MODE, STRIDED>), dim3(wgs), dim3(256), 0, 0, a, b, c, n);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 10 < best) best = ms / 10;
  }
  printf("%-40s U=%d %-11s wgs %7u: %7.1f us  %5.2f TB/s\n", name, U, STRIDED ? "strided" : "consecutive", wgs, best * 1e3, bytes / best / 1e9);
}
int main() {
  const long n = 32l * 256 * 256 * 64 * 2 / 16;          // one 256^2 x 64 bf16 tensor at batch 32 = 268 MB
  uint4 *a, *b, *c;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16));
  CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 2, n * 16));
  const double B = n * 16.0;
#define ALL(MODE, NAME, BYTES) \
  run<1, MODE, true>(NAME, a, b, c, n, BYTES); run<2, MODE, true>(NAME, a, b, c, n, BYTES); run<4, MODE, true>(NAME, a, b, c, n, BYTES); \
  run<8, MODE, true>(NAME, a, b, c, n, BYTES); run<2, MODE, false>(NAME, a, b, c, n, BYTES); run<4, MODE, false>(NAME, a, b, c, n, BYTES);
  ALL(0, "in place: a = f(a, b)  (3 x 268 MB)", 3 * B)
  ALL(1, "copy: c = f(a)  (2 x 268 MB)", 2 * B)
  ALL(2, "a = f(a, b), c = a  (4 x 268 MB)", 4 * B)
  return 0;
}
