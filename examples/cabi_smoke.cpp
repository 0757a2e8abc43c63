// A consumer of libposegan_hip that is neither Python nor PyTorch: plain C++ + the HIP runtime, only include/posegan_hip.h.
// It is the shape of the binding a non-Python host would write: raw device pointers, sizes, a hipStream_t
// passed as void*, integer return codes, pg_last_error() for the message.
//   hipcc --offload-arch=gfx950 -Iinclude examples/cabi_smoke.cpp -Lpose-transfer_amd/lib -lposegan_hip -o cabi_smoke
// Checks, against loops on the host, three entry points of the training step:
//   pg_cords_to_map   key-point heat-maps                       reference utils/pose_utils.py:79-86
//   pg_l1_loss        L1 penalty + its gradient                 reference models/pose_gan.py:105
//   pg_adam           torch.optim.Adam single fused step        reference models/pose_gan.py:50-51
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "posegan_hip.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
  } while (0)
#define PG_OK(x)                                                                   \
  do {                                                                             \
    if ((x) != 0) { printf("%s failed: %s\n", #x, pg_last_error()); return 3; }    \
  } while (0)

template <typename T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

int main() {
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  printf("libposegan_hip version %d\n", pg_version());
  int bad = 0;

  {  // ---- heat-maps: 2 samples, 3 key-points (one missing), 12 x 10, sigma 6 (the reference's default)
    const int N = 2, P = 3, H = 12, W = 10;
    std::vector<float> cords = {3, 4, -1, -1, 11, 9, 0, 0, 6, 5, 2, 8};      // (y, x) per key-point
    float* dc = to_device(cords);
    float* dout = nullptr;
    HIP_OK(hipMalloc(&dout, sizeof(float) * N * P * H * W));
    PG_OK(pg_cords_to_map(dc, N, P, H, W, 6.0f, dout, (int64_t)P * H * W, (int64_t)H * W, W, 1, st));
    std::vector<float> out(N * P * H * W);
    HIP_OK(hipMemcpyAsync(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (int n = 0; n < N; ++n)
      for (int p = 0; p < P; ++p)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const double cy = cords[(n * P + p) * 2], cx = cords[(n * P + p) * 2 + 1];
            const float ref = (cy < 0 || cx < 0) ? 0.f : (float)std::exp(-((y - cy) * (y - cy) + (x - cx) * (x - cx)) / (2.0 * 36.0));
            const float got = out[((n * P + p) * H + y) * W + x];
            if (std::fabs(got - ref) > 1e-6f) ++bad;
          }
    printf("pg_cords_to_map: %s\n", bad ? "MISMATCH" : "ok");
    (void)hipFree(dc); (void)hipFree(dout);
  }
  {  // ---- L1 loss + gradient
    const int n = 4099;
    std::vector<float> a(n), b(n);
    for (int i = 0; i < n; ++i) { a[i] = std::sin(0.37f * i); b[i] = std::cos(0.11f * i); }
    float *da = to_device(a), *db = to_device(b), *dl = nullptr, *dg = nullptr;
    HIP_OK(hipMalloc(&dl, 4)); HIP_OK(hipMemsetAsync(dl, 0, 4, st));
    HIP_OK(hipMalloc(&dg, 4 * n));
    const float scale = 0.25f;
    PG_OK(pg_l1_loss(da, db, n, scale, dl, dg, 0, st));
    float loss = 0.f; std::vector<float> g(n);
    HIP_OK(hipMemcpyAsync(&loss, dl, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(g.data(), dg, 4 * n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    double ref = 0.0; int gb = 0;
    for (int i = 0; i < n; ++i) {
      ref += scale * std::fabs((double)a[i] - b[i]);
      const float gr = a[i] > b[i] ? scale : (a[i] < b[i] ? -scale : 0.f);
      if (std::fabs(g[i] - gr) > 1e-7f) ++gb;
    }
    const bool ok = std::fabs(loss - ref) < 1e-3 * std::fabs(ref) && gb == 0;
    printf("pg_l1_loss: %s (%.6f vs %.6f)\n", ok ? "ok" : "MISMATCH", loss, ref);
    bad += ok ? 0 : 1;
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dl); (void)hipFree(dg);
  }
  {  // ---- Adam, step t = 3
    const int n = 1024;
    std::vector<float> p(n), g(n), m(n), v(n);
    for (int i = 0; i < n; ++i) { p[i] = 0.01f * (i % 17) - 0.05f; g[i] = std::sin(0.3f * i) * 0.2f; m[i] = 0.01f * std::cos(0.2f * i); v[i] = 1e-4f * (1 + i % 5); }
    float *dp = to_device(p), *dg = to_device(g), *dm = to_device(m), *dv = to_device(v);
    const double lr = 2e-4, b1 = 0.5, b2 = 0.999, eps = 1e-8; const int t = 3;
    const double bc1 = 1.0 - std::pow(b1, t), bc2 = 1.0 - std::pow(b2, t);
    PG_OK(pg_adam(dp, dg, dm, dv, n, (float)b1, (float)b2, (float)eps, (float)(lr / bc1), (float)std::sqrt(bc2), 1.0f, st));
    std::vector<float> pn(n);
    HIP_OK(hipMemcpyAsync(pn.data(), dp, 4 * n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    int pb = 0;
    for (int i = 0; i < n; ++i) {
      const double mm = b1 * m[i] + (1 - b1) * g[i], vv = b2 * v[i] + (1 - b2) * (double)g[i] * g[i];
      const double ref = p[i] - (lr / bc1) * mm / (std::sqrt(vv) / std::sqrt(bc2) + eps);
      if (std::fabs(pn[i] - ref) > 2e-7) ++pb;
    }
    printf("pg_adam: %s\n", pb ? "MISMATCH" : "ok");
    bad += pb ? 1 : 0;
    (void)hipFree(dp); (void)hipFree(dg); (void)hipFree(dm); (void)hipFree(dv);
  }
  // ---- error reporting: a bad argument returns non-zero and leaves a message
  if (pg_l1_loss(nullptr, nullptr, 0, 1.f, nullptr, nullptr, 0, st) == 0) { printf("bad arguments were accepted\n"); ++bad; }
  else printf("error path: \"%s\"\n", pg_last_error());
  (void)hipStreamDestroy(st);
  printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad ? 1 : 0;
}
