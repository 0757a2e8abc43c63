#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, const unsigned short* in, int shift) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  const unsigned short* gp = in + shift + threadIdx.x * 8;      // 16 bytes per lane, misaligned by shift*2 bytes
  __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(gp), lds + (threadIdx.x >> 6) * 256, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned short* l = reinterpret_cast<const unsigned short*>(lds);
  out[threadIdx.x] = (float)l[threadIdx.x * 8] + 1000.f * (float)l[threadIdx.x * 8 + 7];
}
int main() {
  unsigned short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)i;
  unsigned short* d; float* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shift : {0, 1, 2, 3, 5}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, o, d, shift);
    float r[128]; hipMemcpy(r, o, 128 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 128; ++t) { float e = (float)(shift + t * 8) + 1000.f * (float)(shift + t * 8 + 7); if (r[t] != e) ++bad; }
    printf("shift %d (misaligned by %d bytes): %d of 128 lanes wrong; lane1 got %.0f expected %.0f\n", shift, (shift * 2) % 16, bad, r[1], (float)(shift + 8) + 1000.f * (shift + 15));
  }
  return 0;
}
