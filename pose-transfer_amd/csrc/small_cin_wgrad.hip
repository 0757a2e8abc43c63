// Weight gradient of the first-layer convolutions (gfx950): few NCHW input channels (pose heat-maps + RGB), 64 output
// channels — reference models/networks.py:186 (encoder, k3 s1 p1) and :341 (discriminator stem, k4 s2 p0); the gradient
// is what autograd computes for `loss.backward()` in models/pose_gan.py:112,170.
//
// The generic weight-gradient kernel (igemm_wgrad.hip) launches one GEMM per filter tap and pads Cin to a 32-wide tile,
// re-reading dY once per tap.  Here the GEMM is turned around so that ALL taps share one pass over dY:
//     dW[(tap,ci)][co] = sum_pixels dY[pixel][co] * x[pixel*S + tap][ci]        M = co (64), N = (tap,ci), K = pixels
// A workgroup owns an TH x 16 tile of output pixels: the dY tile [pixels][64] and the input patch [ci][rows][cols] of
// that tile are staged in LDS once, every MFMA B operand is a gather from the patch (lane n = (tap,ci) adds its own
// constant offset, the pixel offset is an instruction immediate), and the accumulators (64 x T*Cin, all taps) stay in
// registers across the tiles a persistent workgroup walks.  At the end every workgroup stores its partial result to a
// workspace and a second kernel adds the partials into dW (float atomics straight into dW were measured at 2-3x the time
// of the MFMA loop: ~10-20 M atomics at ~150 G/s); without a workspace the kernel falls back to those atomics.
#include "common.h"

namespace pg {

struct SmallCinWgK {
  pg_src_t src[PG_MAX_SRC];
  int nsrc, Ctot;
  int cstart[PG_MAX_SRC + 1];
  int N, Hi, Wi, Ho, Wo, pad;
  const float* dY;      // NHWC [N][Ho][Wo][64]
  float* dW;            // packed [K*K][64][Ctot], accumulated
  int tiles_x, tiles_y, ntiles;
  float* part;          // != nullptr: per-workgroup partial results [blocks][64][npad] instead of atomics
  int npad;
  int CS;               // LDS floats per patch channel (odd: consecutive channels fall into different banks)
};

// NTW = 32x32 accumulator tiles per wave (the 2 x ceil(T*Ctot/32) tiles of the 64 x T*Ctot result over 4 waves)
template <int K, int S, int TH, int NTW>
__global__ __launch_bounds__(256, 2) void small_cin_wgrad_kernel(const SmallCinWgK p) {
  constexpr int TW = 16, T = K * K, PIX = TH * TW;
  constexpr int PH = (TH - 1) * S + K;
  constexpr int PWU = (TW - 1) * S + K;                     // used patch columns
  constexpr int PWS = (S == 1) ? PWU : (PWU + 1) / 2;       // S=2: one half row per column parity
  constexpr int ROWF = (S == 1) ? PWS : 2 * PWS;            // floats per patch row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dyt = smem;                                        // [PIX][64]
  float* patch = smem + PIX * 64;                           // [Ctot][CS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int mt = wave & 1;                                  // output-channel rows mt*32 .. +31
  const int ntot = T * p.Ctot;
  const int nt_base = blockIdx.y * (2 * NTW);               // column group of this workgroup (wide filters: Cin > 44 at k4)

  // lane constants: n = (tap, ci) of this lane in each of the wave's tiles -> patch offset (+ lhi: the odd pixel of a
  // k pair is one column further)
  int base[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int n = (nt_base + (wave >> 1) + 2 * i) * 32 + l31;
    const bool ok = n < ntot;
    const int tap = ok ? n / p.Ctot : 0;
    const int ci = ok ? n - tap * p.Ctot : 0;
    const int r = tap / K, s = tap % K;
    base[i] = ci * p.CS + r * ROWF + ((S == 1) ? s : (s & 1) * PWS + (s >> 1)) + lhi;
  }
  f32x16 acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const float* dyl = dyt + lhi * 64 + mt * 32 + l31;

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    int b = t;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    __syncthreads();
    // ---- stage the gradient tile and the input patch of all channels (zero outside the image).  Loads are issued in
    //      branch-free batches (clamped address + select) so that a batch costs one memory latency, not one per element.
    {
      float4 v[PIX * 16 / 256];
#pragma unroll
      for (int u = 0; u < PIX * 16 / 256; ++u) {
        const int e = tid + 256 * u;
        const int px = e >> 4, c4 = e & 15;
        const int oy = oy0 + px / TW, ox = ox0 + px % TW;
        const bool ok = (oy < p.Ho) & (ox < p.Wo);
        const long off = ok ? (((long)n * p.Ho + oy) * p.Wo + ox) * 64 + c4 * 4 : 0;
        v[u] = *reinterpret_cast<const float4*>(p.dY + off);
        if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < PIX * 16 / 256; ++u) *reinterpret_cast<float4*>(&dyt[(tid + 256 * u) * 4]) = v[u];
    }
    constexpr int U = 8;
    for (int j = 0; j < p.nsrc; ++j) {                        // per source: pointer and strides stay wave-uniform
      const char* ptr = reinterpret_cast<const char*>(p.src[j].ptr + (long)n * p.src[j].sN);
      const int sC = (int)p.src[j].sC, sH = (int)p.src[j].sH, sW = (int)p.src[j].sW;
      const int cbase = p.cstart[j];
      const int E = p.src[j].C * PH * PWU;
      for (int e0 = tid; e0 < E; e0 += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + 256 * u;
          const int col = e % PWU;
          const int rr = (e / PWU) % PH;
          const int c = e / (PWU * PH);
          const int iy = iy0 + rr, ix = ix0 + col;
          const bool ok = (e < E) & (iy >= 0) & (iy < p.Hi) & (ix >= 0) & (ix < p.Wi);
          const int off = ok ? c * sC + iy * sH + ix * sW : 0;
          v[u] = ldg32(ptr, (long)off * 4);
          if (!ok) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + 256 * u;
          const int col = e % PWU;
          const int rr = (e / PWU) % PH;
          const int c = e / (PWU * PH) + cbase;
          if (e < E) patch[c * p.CS + rr * ROWF + ((S == 1) ? col : (col & 1) * PWS + (col >> 1))] = v[u];
        }
      }
    }
    __syncthreads();
    // ---- MFMA over the tile's pixels: k pair (2ks, 2ks+1) = pixels (y, 2xh) and (y, 2xh+1) on the two lane halves
#pragma unroll
    for (int ks = 0; ks < PIX / 2; ++ks) {
      const int y = ks / (TW / 2), xh = ks % (TW / 2);
      const int pixoff = y * S * ROWF + 2 * xh;              // compile-time: folds into the ds_read offset
      const float av = dyl[ks * 128];
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const float bv = patch[base[i] + pixoff];
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
      }
    }
  }
  // ---- dW[tap][co][ci] += acc   (rows = co, lane column = (tap, ci))
  if (p.part) {                                             // plain coalesced stores; small_cin_wgrad_reduce adds them up
    float* o = p.part + (long)blockIdx.x * 64 * p.npad + (long)(mt * 32 + 4 * lhi) * p.npad + (nt_base + (wave >> 1)) * 32 + l31;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
      if ((nt_base + (wave >> 1) + 2 * i) * 32 < p.npad) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * p.npad + i * 64] = acc[i][r];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int n = (nt_base + (wave >> 1) + 2 * i) * 32 + l31;
    if (n >= ntot) continue;
    const int tap = n / p.Ctot;
    float* o = p.dW + (long)tap * 64 * p.Ctot + (n - tap * p.Ctot);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      atomicAdd(o + co * p.Ctot, acc[i][r]);
    }
  }
}

// dW[tap][co][ci] += sum over workgroups of part[b][co][n = tap*Ctot + ci]; blockIdx.y strides over the workgroups
__global__ __launch_bounds__(256) void small_cin_wgrad_reduce_kernel(const float* part, int nblocks, int npad, int ntot,
                                                                     int Ctot, float* dW) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 64 * npad) return;
  const int co = e / npad, n = e - co * npad;
  if (n >= ntot) return;
  const long stride = (long)64 * npad;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int b = blockIdx.y;
  const int step = gridDim.y;
  for (; b + 3 * step < nblocks; b += 4 * step) {
    t0 += part[b * stride + e];
    t1 += part[(b + step) * stride + e];
    t2 += part[(b + 2 * step) * stride + e];
    t3 += part[(b + 3 * step) * stride + e];
  }
  for (; b < nblocks; b += step) t0 += part[b * stride + e];
  const int tap = n / Ctot;
  atomicAdd(dW + ((long)tap * 64 + co) * Ctot + (n - tap * Ctot), (t0 + t1) + (t2 + t3));
}

template <int K, int S, int TH, int NTW>
static int launch_small_cin_wgrad(SmallCinWgK& k, float* ws, long ws_floats, hipStream_t st, int groups = 1) {
  constexpr int PH = (TH - 1) * S + K, PWU = 15 * S + K, PWS = (S == 1) ? PWU : (PWU + 1) / 2;
  constexpr int ROWF = (S == 1) ? PWS : 2 * PWS;
  k.CS = (PH * ROWF) | 1;
  k.tiles_x = (k.Wo + 15) / 16; k.tiles_y = (k.Ho + TH - 1) / TH;
  k.ntiles = k.tiles_x * k.tiles_y * k.N;
  const size_t lds = sizeof(float) * ((size_t)TH * 16 * 64 + (size_t)k.Ctot * k.CS);
  if (lds > 160 * 1024) return -1;
  auto kern = small_cin_wgrad_kernel<K, S, TH, NTW>;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    PG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  const int per_cu = (int)((160 * 1024) / lds);
  int blocks = 256 * (per_cu >= 3 ? 3 : per_cu >= 2 ? 2 : 1);
  if (const char* e = getenv("PG_SCW_BLOCKS")) blocks = atoi(e);
  if (blocks > k.ntiles) blocks = k.ntiles;
  k.npad = (K * K * k.Ctot + 31) / 32 * 32;
  const long need = (long)blocks * 64 * k.npad;
  if (ws == nullptr || ws_floats < need) {                   // no (or too small a) workspace: float atomics into dW
    PG_REQUIRE(!deterministic(), "pg_small_cin_wgrad: PG_DETERMINISTIC needs the workspace (%ld floats)", need);
    k.part = nullptr;
    if (blocks > 256) blocks = 256;                          // measured: the atomics, not the MFMA loop, set the time
  } else {
    k.part = ws;
  }
  PG_KLAUNCH(kern, dim3((unsigned)blocks, (unsigned)groups), dim3(256), lds, st, k);
  if (k.part) {
    const int ry = (blocks >= 32 && !deterministic()) ? 16 : 1;
    PG_KLAUNCH(small_cin_wgrad_reduce_kernel, dim3((unsigned)((64 * k.npad + 255) / 256), (unsigned)ry), dim3(256),
                       0, st, k.part, blocks, k.npad, K * K * k.Ctot, k.Ctot, k.dW);
  }
  return 0;
}

}  // namespace pg

extern "C" int pg_small_cin_wgrad(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K,
                                  int32_t stride, int32_t pad, const float* dY, float* dW, float* workspace,
                                  int64_t workspace_floats, void* stream) {
  PG_REQUIRE(src && nsrc >= 1 && nsrc <= PG_MAX_SRC && dY && dW, "pg_small_cin_wgrad: bad arguments");
  PG_REQUIRE((K == 3 && stride == 1) || (K == 4 && stride == 2), "pg_small_cin_wgrad: only k3s1 / k4s2 (got k%d s%d)", K, stride);
  pg::SmallCinWgK k;
  memset(&k, 0, sizeof(k));
  int c = 0;
  for (int j = 0; j < nsrc; ++j) { k.src[j] = src[j]; k.cstart[j] = c; c += src[j].C; }
  for (int j = nsrc; j <= PG_MAX_SRC; ++j) k.cstart[j] = c;
  k.nsrc = nsrc; k.Ctot = c;
  k.N = N; k.Hi = Hi; k.Wi = Wi; k.pad = pad;
  k.Ho = (Hi + 2 * pad - K) / stride + 1; k.Wo = (Wi + 2 * pad - K) / stride + 1;
  PG_REQUIRE(k.Ho > 0 && k.Wo > 0 && N > 0, "pg_small_cin_wgrad: empty output");
  k.dY = dY; k.dW = dW;
  const int tiles = 2 * ((K * K * c + 31) / 32);             // 32x32 accumulator tiles of the 64 x T*Cin result
  hipStream_t st = (hipStream_t)stream;
  int rc = -1;
  if (K == 3) {
    PG_REQUIRE(tiles <= 20, "pg_small_cin_wgrad: k3 supports Cin <= 35 (got %d)", c);
    rc = (tiles <= 12) ? pg::launch_small_cin_wgrad<3, 1, 8, 3>(k, workspace, workspace_floats, st)
                       : pg::launch_small_cin_wgrad<3, 1, 8, 5>(k, workspace, workspace_floats, st);     // P = 32: 3 + 32 channels
  } else {
    PG_REQUIRE(tiles <= 88, "pg_small_cin_wgrad: k4 supports Cin <= 88 (got %d)", c);
    if (tiles <= 24) rc = pg::launch_small_cin_wgrad<4, 2, 4, 6>(k, workspace, workspace_floats, st);
    else if (tiles <= 44) rc = pg::launch_small_cin_wgrad<4, 2, 4, 11>(k, workspace, workspace_floats, st);
    else rc = pg::launch_small_cin_wgrad<4, 2, 4, 11>(k, workspace, workspace_floats, st, 2);      // P = 32: 70 channels, two column groups
  }
  PG_REQUIRE(rc == 0, "pg_small_cin_wgrad: patch does not fit LDS");
  pg::last_info() = 5 | (1 << 4) | (1 << 16) | (1 << 30);     // tile code 5 = all-taps patch kernel, scalar X
  PG_LAUNCH_OK("pg_small_cin_wgrad");
  return 0;
}
