// probe: semantics of ds_read_b64_tr_b16 on gfx950 (which element of LDS lands in which lane / register slot)
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* addr_in, uint16_t* out) {
  __shared__ uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned a = (unsigned)(size_t)lds + (unsigned)addr_in[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
  int* d_a; uint16_t* d_o;
  hipMalloc(&d_a, 64 * 4); hipMalloc(&d_o, 64 * 4 * 2);
  const char* names[] = {"uniform 0", "lane*8", "group-contiguous 128B", "4x(4 lanes x 32B rows, stride 64B)", "rows stride 512B: (l&3)*8 + ((l>>2)&3)*512 + (l>>4)*32"};
  for (int pat = 0; pat < 5; ++pat) {
    int a[64];
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) a[l] = 0;
      else if (pat == 1) a[l] = l * 8;
      else if (pat == 2) a[l] = (l >> 4) * 128 + (l & 15) * 8;
      else if (pat == 3) a[l] = (l & 3) * 8 + ((l >> 2) & 3) * 64 + (l >> 4) * 256;
      else a[l] = (l & 3) * 8 + ((l >> 2) & 3) * 512 + (l >> 4) * 32;
    }
    hipMemcpy(d_a, a, sizeof(a), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_a, d_o);
    uint16_t o[256];
    hipMemcpy(o, d_o, sizeof(o), hipMemcpyDeviceToHost);
    printf("pattern %d: %s  (byte address per lane; values = bf16-element index read)\n", pat, names[pat]);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d addr %4d -> %4d %4d %4d %4d%s", l, a[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3], (l & 1) ? "\n" : "   |");
    }
  }
  return 0;
}
