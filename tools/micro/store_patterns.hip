// Micro-benchmark behind the warp-forward analysis: what do the kernel's store / mask-read patterns cost by themselves?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_patterns.hip -o gpurun_out/store_patterns && gpurun_out/store_patterns
// Shape: level 0 at batch 32 (2 M pixels x 64 channels): out = 16 B per lane (bf16 x 8), arg-max = 8 B per lane, masks = 10 floats per pixel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* out, uint2* amax, const float* masks, long items) {
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
    unsigned v = (unsigned)it;
    if (MODE & 8) {                                    // mask pattern of warp_fwd5: lane sub of a pixel's 8 lanes reads float sub, lanes 0..1 float 8 + sub
      const long pix = it >> 3; const int sub = it & 7;
      const float m0 = masks[pix * 10 + sub];
      const float m1 = sub < 2 ? masks[pix * 10 + 8 + sub] : 0.f;
      v += (m0 != 0.f) + (m1 != 0.f);
    }
    if (MODE & 64) {                                   // the same bytes as one contiguous stream (what a plain copy would read)
      if ((it & 7) < 5) { const float2 m = reinterpret_cast<const float2*>(masks)[(it >> 3) * 5 + (it & 7)]; v += (m.x != 0.f) + (m.y != 0.f); }
    }
    const uint4 o = make_uint4(v, v + 1, v + 2, v + 3);
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    if (MODE & 1) { if (MODE & 16) { u4v q = {v, v + 1, v + 2, v + 3}; __builtin_nontemporal_store(q, reinterpret_cast<u4v*>(out) + it); } else out[it] = o; }
    if (MODE & 2) { if (MODE & 16) { u2v q = {v, v}; __builtin_nontemporal_store(q, reinterpret_cast<u2v*>(amax) + it); } else amax[it] = make_uint2(v, v); }
    if (MODE & 4) {                                    // arg-max bytes as 16-byte stores from the even lanes
      const unsigned w = __shfl_down(v, 1);
      if (!(it & 1)) reinterpret_cast<uint4*>(amax)[it >> 1] = make_uint4(v, v, w, w);
    }
    if (MODE & 32) {                                   // out as two 8-byte stores (the fp32-output width of the old kernels)
      reinterpret_cast<uint2*>(out)[2 * it] = make_uint2(v, v); reinterpret_cast<uint2*>(out)[2 * it + 1] = make_uint2(v, v);
    }
  }
}
template <int MODE>
void run(const char* name, uint4* out, uint2* amax, float* masks, long items, int wgs, double bytes) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, amax, masks, items);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, amax, masks, items);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 10 < best) best = ms / 10;
  }
  printf("%-64s wgs %6d: %7.1f us  %5.2f TB/s\n", name, wgs, best * 1e3, bytes / best / 1e9);
}
int main() {
  const long pix = 32l * 256 * 256, items = pix * 8;
  uint4* out; uint2* amax; float* masks;
  CK(hipMalloc(&out, items * 16)); CK(hipMalloc(&amax, items * 8)); CK(hipMalloc(&masks, pix * 40));
  CK(hipMemset(masks, 0, pix * 40));
  const double bo = items * 16.0, ba = items * 8.0, bm = pix * 40.0;
  for (int wgs : {8192, 65536}) {
    run<1>("out 16 B / lane", out, amax, masks, items, wgs, bo);
    run<32>("out as 2 x 8 B / lane", out, amax, masks, items, wgs, bo);
    run<2>("arg-max 8 B / lane", out, amax, masks, items, wgs, ba);
    run<3>("out 16 + arg-max 8", out, amax, masks, items, wgs, bo + ba);
    run<5>("out 16 + arg-max 16 from even lanes", out, amax, masks, items, wgs, bo + ba);
    run<19>("out 16 + arg-max 8, nontemporal", out, amax, masks, items, wgs, bo + ba);
    run<8>("mask reads (4 B per lane, 8 + 2 of 10)", out, amax, masks, items, wgs, bm);
    run<64>("mask reads (8 B per lane, contiguous)", out, amax, masks, items, wgs, bm);
    run<11>("masks (4 B) + out 16 + arg-max 8", out, amax, masks, items, wgs, bo + ba + bm);
    run<67>("masks (8 B contiguous) + out 16 + arg-max 8", out, amax, masks, items, wgs, bo + ba + bm);
    run<13>("masks (4 B) + out 16 + arg-max 16 even", out, amax, masks, items, wgs, bo + ba + bm);
  }
  return 0;
}
