// Data-gradient of the generator's 256->3 output convolution (reference models/networks.py:228: ReLU -> Conv2d(k3,p1)
// -> Tanh; autograd of `loss.backward()`, models/pose_gan.py:170) from the im2col'd 3-channel gradient
//     G[pixel][t],  t = (tap, co) < 27, row pitch 32           (pg_im2col_taps, also the weight-gradient operand)
//     dX[pixel][ci] = sum_t G[pixel][t] * Wt[ci][t]             Wt = the weight viewed as [Cin][27 -> 32]
// scattered over the virtual concat with act'(fwd) / dropout mask like pg_conv's data-gradient epilogue.
// K = 27: there is no GEMM here — as a pg_conv launch this was ONE K tile followed by a 4-byte-per-lane scatter
// epilogue (285 us for 537 MB).  This kernel is the streaming form: a wave owns one pixel at a time, a lane owns 4
// consecutive input channels and keeps their 27 x 4 weights in registers for the whole launch; the pixel's 27 gradient
// values arrive with ONE coalesced 128-byte load and are broadcast lane -> SGPR (v_readlane), so the contraction is
// 108 FMAs with a scalar operand per pixel and wave, and every forward value / mask / result moves as 16 bytes per lane.
#include "common.h"

namespace pg {

struct OutDgradK {
  const float* G;          // [npix][32]
  const float* Wt;         // [Ctot][32]
  int npix, ppix, Ctot;
  pg_dst_t dst[PG_MAX_SRC];
  int ndst;
  int dstart[PG_MAX_SRC + 1];
  float* wpart;            // WG = true: per-workgroup partial weight gradients [blocks][Ctot][28] (round 3)
  int g_bf16_pitch;        // 0: G is fp32 with row pitch 32; else G is bf16 with this row pitch (64)
  const float* dpre;       // DIRECT = true: the NCHW [N][3][H][W] gradient itself — the 27 (tap, channel) values of a pixel are
  int H, W;                //   gathered from it, no im2col'd copy exists (round 3)
};

constexpr int ODG_T = 27;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wc99-designator"
__device__ __attribute__((aligned(16))) const float kOnesO[260] = {[0 ... 259] = 1.0f};
#pragma clang diagnostic pop
__device__ __attribute__((aligned(16))) const float kIdentO[4] = {1.0f, 0.0f, 1.0f, 0.0f};

// 4 consecutive elements (element index idx) of an fp32 or bf16 tensor; the branch is uniform per destination
template <bool bf>
__device__ __forceinline__ float4 ld4_dt(const float* base, long idx) {
  if constexpr (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(base + idx);
}

// WG = true (round 3, bf16 STORAGE): the same pass also accumulates the WEIGHT gradient dW[t][ci] = sum_pixels G[pixel][t] *
// x[pixel][ci].  With bf16 storage every destination's `fwd` is the ACTIVATED bf16 operand the forward contraction read
// (relu(z): its sign gives act', its value IS x), so g (27 scalars per pixel, already broadcast) and x (4 channels per
// lane, already loaded) are both in registers: 108 more FMAs per pixel and lane instead of a second kernel that reads
// the three activation tensors again (fp32 kernel before: 0.59 ms + 1.07 GB at batch 32).  Per-lane sums are merged per
// workgroup through LDS and written to a partial buffer that out_conv_wgrad_reduce_kernel adds into dW.
// DG: compute and store the data gradient; BF: every destination's gradient / forward tensor is bf16 (compile-time, so that
// the batched loads stay straight-line code).  Instantiated as <DG, !WG, fp32> (the fp32 paths), <DG, !WG, bf16> and
// <!DG, WG, bf16>: two lean passes beat one fused pass (256 VGPRs, one wave per SIMD: 3.07 ms at batch 32 against
// 0.4 + 0.4 ms).
// BS (round 4): a destination carries pg_dst_t.bsums.  The grid is (workgroups per sample, N) then — a wave stays inside ONE
// sample, keeps (sum r, sum r * f) of the values it stores per lane (a lane = 4 channels of one destination) and the workgroup
// adds them once at the end, per destination (sums_mode 1 of pg_norm_bwd_apply_v2: f is the raw forward value).
template <bool DG, bool WG, bool BF, bool DIRECT = false, bool BS = false>
__global__ __launch_bounds__(256) void out_conv_dgrad_kernel(const OutDgradK p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float bs_s = 0.f, bs_q = 0.f;
  // DIRECT: lane t < 27 = (tap, channel) = (t / 3, t % 3) reads dpre[n][c][y - (r - 1)][x - (s - 1)] (pg_im2col_taps' formula)
  const int d_c = (lane % 27) % 3, d_tap = (lane % 27) / 3;
  const int d_dy = 1 - d_tap / 3, d_dx = 1 - d_tap % 3;
  typedef float f32x2w __attribute__((ext_vector_type(2)));
  f32x2w aw01[WG ? ODG_T : 1], aw23[WG ? ODG_T : 1];          // channel pairs (packed fp32 FMAs)
  if constexpr (WG) {
#pragma unroll
    for (int t = 0; t < ODG_T; ++t) { aw01[t] = f32x2w{0.f, 0.f}; aw23[t] = f32x2w{0.f, 0.f}; }
  }
  const int cg = lane * 4;
  const bool live = cg < p.Ctot;
  const int cgc = live ? cg : 0;
  // this lane's weights
  float w[DG ? 4 : 1][DG ? ODG_T + 1 : 1];
  if constexpr (DG) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int q = 0; q < (ODG_T + 1) / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p.Wt + (long)(cgc + e) * 32 + q * 4);
        w[e][q * 4] = v.x; w[e][q * 4 + 1] = v.y; w[e][q * 4 + 2] = v.z; w[e][q * 4 + 3] = v.w;
      }
  }
  // this lane's destination (constant-index picks: no scratch copy of the kernel argument)
  float* gradp = p.dst[0].grad;
  const float *fwd0 = p.dst[0].fwd, *aff0 = p.dst[0].aff, *mask0 = p.dst[0].mask;
  int C = p.dst[0].C, dact = p.dst[0].act, dacc = p.dst[0].accumulate, cst = 0, dfl = p.dst[0].flags;
#pragma unroll
  for (int q = 1; q < PG_MAX_SRC; ++q)
    if (q < p.ndst && cgc >= p.dstart[q]) {
      gradp = p.dst[q].grad; fwd0 = p.dst[q].fwd; aff0 = p.dst[q].aff; mask0 = p.dst[q].mask;
      C = p.dst[q].C; dact = p.dst[q].act; dacc = p.dst[q].accumulate; cst = p.dstart[q]; dfl = p.dst[q].flags;
    }
  (void)dfl;
  int qsel = 0;
#pragma unroll
  for (int q = 1; q < PG_MAX_SRC; ++q)
    if (q < p.ndst && cgc >= p.dstart[q]) qsel = q;
  constexpr bool gbf = BF;
  const int c = cgc - cst;
  const float slope = act_slope(dact);
  const bool has_fwd = fwd0 != nullptr, has_aff = aff0 != nullptr && has_fwd, has_mask = mask0 != nullptr;
  const float* const fwdp = has_fwd ? fwd0 : gradp;            // dummy (valid) reads when there is no activation
  constexpr bool fbf = BF;
  const float dslope = has_fwd ? slope : 1.f;                  // slope 1: act' == 1 whatever was read

  // absent mask / affine / accumulation: loads stay unconditional (a load under a divergent branch serialises its
  // latency) but go to a per-lane-constant dummy address that stays in L1/L2
  const float* const maskp = has_mask ? mask0 : kOnesO;
  const float* const affp = has_aff ? aff0 : kIdentO;
  const int affmul = has_aff ? 2 : 0;
  const bool accum = dacc != 0;

  constexpr int U = 4;                                         // consecutive pixels per wave and step (loads in flight)
  const int stride = gridDim.x * 4 * U;
  // BS: pixels [pix_lo, pix_hi) of sample blockIdx.y; otherwise every pixel of the launch
  const int pix_lo = BS ? (int)blockIdx.y * p.ppix : 0;
  const int pix_hi = BS ? pix_lo + p.ppix : p.npix;
  for (int base0 = pix_lo + (blockIdx.x * 4 + wave) * U; base0 < pix_hi; base0 += stride) {
    const int base = __builtin_amdgcn_readfirstlane(base0);
    float gl[U];
    float4 f[U], m[U], old[U];
    float2 ab[U];
    long idx[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = base + u < pix_hi;                               // wave-uniform
      const int pix = ok[u] ? base + u : base;
      const int n = pix / p.ppix;
      idx[u] = (long)pix * C + c;
      if constexpr (DIRECT) {
        const int rem = pix - n * p.ppix;
        const int y = rem / p.W, x = rem - y * p.W;            // wave-uniform
        const int yy = y + d_dy, xx = x + d_dx;
        const bool in = (yy >= 0) & (yy < p.H) & (xx >= 0) & (xx < p.W);
        const float v = p.dpre[((long)(n * 3 + d_c) * p.H + (in ? yy : y)) * p.W + (in ? xx : x)];
        gl[u] = in ? v : 0.f;
      } else if (p.g_bf16_pitch)                               // (wave-uniform) bf16 rows of the bf16 data path
        gl[u] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p.G)[(long)pix * p.g_bf16_pitch + (lane & 31)] << 16);
      else
        gl[u] = p.G[(long)pix * 32 + (lane & 31)];             // one 128-byte row, lanes 32..63 mirror it
      f[u] = ld4_dt<fbf>(fwdp, has_fwd ? idx[u] : (long)c);
      m[u] = *reinterpret_cast<const float4*>(has_mask ? maskp + (long)n * C + c : maskp + (c & 255));
      ab[u] = *reinterpret_cast<const float2*>(affp + affmul * n);
      if constexpr (DG) old[u] = ld4_dt<gbf>(gradp, accum ? idx[u] : (long)c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (DG) {
        // packed fp32 FMAs (v_pk_fma_f32: two channels per instruction, the broadcast gradient value in both halves): the
        // loop is VALU-bound (27 readlanes + 108 FMAs per pixel and wave), 54 packed instead of 108 scalar FMAs
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < ODG_T; ++t) {
          const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl[u]), t));
          const f32x2 gg = {g, g};
          const f32x2 w01 = {w[0][t], w[1][t]}, w23 = {w[2][t], w[3][t]};
          a01 = __builtin_elementwise_fma(gg, w01, a01);
          a23 = __builtin_elementwise_fma(gg, w23, a23);
        }
        acc[0] = a01[0]; acc[1] = a01[1]; acc[2] = a23[0]; acc[3] = a23[1];
      }
      const float f4[4] = {f[u].x, f[u].y, f[u].z, f[u].w}, m4[4] = {m[u].x, m[u].y, m[u].z, m[u].w};
      const float o4[4] = {old[u].x, old[u].y, old[u].z, old[u].w};
      if constexpr (WG) {
        if (ok[u]) {                                           // wave-uniform
          const f32x2w x01 = {f4[0], f4[1]}, x23 = {f4[2], f4[3]};
#pragma unroll
          for (int t = 0; t < ODG_T; ++t) {
            const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl[u]), t));
            const f32x2w gg = {g, g};
            aw01[t] = __builtin_elementwise_fma(gg, x01, aw01[t]);
            aw23[t] = __builtin_elementwise_fma(gg, x23, aw23[t]);
          }
        }
      }
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(f4[e], ab[u].x, ab[u].y) * m4[e];
        r[e] = fmaf(acc[e] * m4[e], act_grad_s(z, dslope), accum ? o4[e] : 0.f);
      }
      if (DG && live && ok[u]) {
        if constexpr (gbf) {
          uint2 pk;
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.x) : "v"(r[0]), "v"(r[1]));
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.y) : "v"(r[2]), "v"(r[3]));
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(gradp) + idx[u]) = pk;
        } else {
          *reinterpret_cast<float4*>(gradp + idx[u]) = make_float4(r[0], r[1], r[2], r[3]);
        }
        if constexpr (BS) {
          bs_s += (r[0] + r[1]) + (r[2] + r[3]);
          bs_q += fmaf(r[0], f4[0], fmaf(r[1], f4[1], fmaf(r[2], f4[2], r[3] * f4[3])));
        }
      }
    }
  }
  if constexpr (BS) {
    __shared__ double redb[4][PG_MAX_SRC][2];
#pragma unroll
    for (int q = 0; q < PG_MAX_SRC; ++q) {
      const bool mine = live && qsel == q;
      const double ds = wave_sum_d(mine ? (double)bs_s : 0.0), dq = wave_sum_d(mine ? (double)bs_q : 0.0);
      if (lane == 0) { redb[wave][q][0] = ds; redb[wave][q][1] = dq; }
    }
    __syncthreads();
    if (threadIdx.x < PG_MAX_SRC) {
      const int q = threadIdx.x;
      double* b = p.dst[0].bsums;
#pragma unroll
      for (int k = 1; k < PG_MAX_SRC; ++k)
        if (q == k) b = p.dst[k].bsums;
      if (q < p.ndst && b != nullptr) {
        const double ds = (redb[0][q][0] + redb[1][q][0]) + (redb[2][q][0] + redb[3][q][0]);
        const double dq = (redb[0][q][1] + redb[1][q][1]) + (redb[2][q][1] + redb[3][q][1]);
        double* slot = b + ((long)blockIdx.y * PG_STAT_SLOTS + (blockIdx.x % PG_STAT_SLOTS)) * 2;
        if (ds != 0.0 || dq != 0.0) { atomicAdd(&slot[0], ds); atomicAdd(&slot[1], dq); }
      }
    }
  }
  if constexpr (WG) {
    // merge the four waves' sums (a lane's 4 x 27 values per round through LDS), then one partial block per workgroup
    __shared__ float red[64 * 4 * 28];
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < ODG_T; ++t) {
            float* q = red + (lane * 4 + e) * 28 + t;
            *q = (wv == 0 ? 0.f : *q) + (e < 2 ? aw01[t][e] : aw23[t][e - 2]);
          }
      }
      __syncthreads();
    }
    float* dstp = p.wpart + (long)blockIdx.x * p.Ctot * 28;
    for (int i = threadIdx.x; i < p.Ctot * 28; i += 256) dstp[i] = red[i];
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Data gradient AND weight gradient of the output convolution in ONE pass on the matrix cores (round 3, bf16 STORAGE, DIRECT
// gradient).  The streaming kernels above are VALU-bound (27 v_readlane + 54 packed FMAs + ~40 epilogue instructions per pixel
// and wave) and each reads the three activated forward tensors: 0.85 + 0.68 ms at batch 32 for 1.07 GB read + 1.07 GB written.
// As matrix products the layer is tiny:
//     dX^T[ci][pixel] = Wt[ci][t] * G^T[t][pixel]            (K = 27 -> 32)
//     dW[t][ci]       = sum_pixels G[pixel][t] * x[pixel][ci] (K = pixels)
// A workgroup walks 32-pixel tiles (one image-row segment); wave w owns channels 64 w .. 64 w + 63 (two 32-channel tiles):
//   * the tile's 3 x 3 x 34 patch of dpre (NCHW, zero outside the image) is staged once per workgroup in LDS (double-buffered,
//     one raw s_barrier per tile); both G operands are gathered from it with ds_read_b32 — data gradient: lane = pixel, 16 of
//     the 32 k values; weight gradient: lane = t = (tap, channel), 16 of the 32 pixels;
//   * the wave's forward values arrive as 16 bytes per lane (full 64-byte channel runs per pixel) into a wave-private LDS
//     tile [32 pixels][64 channels]; from there: act' per accumulator quad (lane = pixel, 4 consecutive channels: ds_read_b64)
//     and the weight gradient's A operand [m = ci][k = pixel] through ds_read_b64_tr_b16 (addressing as wgrad_bf16.hip);
//   * M = channel, N = pixel for the data gradient, so an accumulator register quad is four consecutive channels of one pixel;
//     the results go back through the same LDS tile and leave as 16 bytes per lane;
//   * the weight gradient accumulates in two persistent accumulator tiles per wave, written once per workgroup to the partial
//     buffer of out_conv_wgrad_reduce_kernel;
//   * the next tile's global loads are requested BEFORE this tile's stores: gfx950 counts loads and stores in one vmcnt, a
//     load waited for behind a store also waits for the store's acknowledgement.
// Host: every destination bf16 with an ACTIVATED bf16 forward operand, no affine / mask / accumulate, C % 32 == 0, W % 32 == 0.
typedef __bf16 obf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned opack_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
constexpr int OXP = 144;           // row pitch (bytes) of a wave's forward-value tile in LDS
static_assert(4 * OXP == 576 && 16 * OXP == 2304 && 20 * OXP == 2880, "ds_read_b64_tr_b16 offsets in out_conv_bwd_mfma_kernel");
constexpr int OPATCH = 336;        // 3 channels x 3 rows x pitch 36 (34 used) = 324 floats + spare
__device__ __forceinline__ float owave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256, 2) void out_conv_bwd_mfma_kernel(const OutDgradK p) {
  __shared__ __attribute__((aligned(16))) uint4 a_lds[8 * 2 * 64];          // [channel tile][k step][lane]: 16 KB
  // per wave: [32 pixels][64 channels] bf16, row pitch OXP = 144 bytes: with the natural 128-byte pitch the accumulator-layout
  // accesses (lane = pixel row, 8 bytes) put 32 lanes on 4 banks — 16-way conflicts on 16 of the tile's LDS instructions
  // (round-4 counters: SQ_LDS_BANK_CONFLICT 0.8 of SQ_LDS_IDX_ACTIVE, SQ_WAIT_INST_LDS a third of the wave cycles); 36 words: 2-way
  __shared__ __attribute__((aligned(16))) char x_lds[4 * 32 * OXP];
  __shared__ float patch[2][OPATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nct = p.Ctot / 32;
  for (int i = tid; i < 8 * 2 * 64; i += 256) {
    const int ct = i >> 7, ks = (i >> 6) & 1, ln = i & 63;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (ct < nct) {
      const int k0 = ks * 16 + (ln >> 5) * 8;
      const float* w = p.Wt + (long)(ct * 32 + (ln & 31)) * 32 + k0;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (k0 + j < ODG_T) ? w[j] : 0.f;        // the pad columns of Wt are not read
      o = make_uint4(opack_bf16(v[0], v[1]), opack_bf16(v[2], v[3]), opack_bf16(v[4], v[5]), opack_bf16(v[6], v[7]));
    }
    a_lds[i] = o;
  }
  for (int i = tid; i < 2 * OPATCH; i += 256) patch[0][i] = 0.f;
  const int HW = p.H * p.W;
  // data-gradient gather: k = ks*16 + lhi*8 + j = (tap, channel) of THIS lane's pixel: patch[c][1 + dy][1 + l31 + dx];
  // k >= 27 reads the spare cell (0)
  int goff[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = (q >> 3) * 16 + lhi * 8 + (q & 7);
    const int tap = k / 3, c = k - tap * 3, r = tap / 3, s_ = tap - r * 3;
    goff[q] = k < ODG_T ? c * 108 + (2 - r) * 36 + (2 - s_) + l31 : 330;
  }
  // weight-gradient gather: lane t = l31 = (tap, channel), pixel k of the tile: patch[c][1 + dy][1 + k + dx]; lanes t >= 27 read
  // t = 0's cells (their accumulator columns are never stored)
  const int tt = l31 < ODG_T ? l31 : 0;
  const int t_tap = tt / 3, t_c = tt - t_tap * 3, t_r = t_tap / 3, t_s = t_tap - t_r * 3;
  const int tbase = t_c * 108 + (2 - t_r) * 36 + (2 - t_s) + lhi * 8;

  const int tpr = p.W / 32;
  const int ntiles = p.npix / 32;
  const int ct0 = 2 * wave;
  const bool act0 = ct0 < nct, act1 = ct0 + 1 < nct;
  // destinations of this wave's two channel tiles (wave-uniform, fixed for the launch; constant-index picks)
  unsigned short* g16[2]; const unsigned short* f16[2]; int Cd[2], c0d[2]; float dsl[2];
  double* bsd[2];                  // (round 4) pg_dst_t.bsums of the tile's destination: fused sums of the following norm backward
#pragma unroll
  for (int cq = 0; cq < 2; ++cq) {
    const int ct = min(ct0 + cq, nct - 1);
    float* gp = p.dst[0].grad; const float* fp = p.dst[0].fwd; int C = p.dst[0].C, st = p.dstart[0], ac = p.dst[0].act;
    double* bq = p.dst[0].bsums;
#pragma unroll
    for (int q = 1; q < PG_MAX_SRC; ++q)
      if (q < p.ndst && ct * 32 >= p.dstart[q]) { gp = p.dst[q].grad; fp = p.dst[q].fwd; C = p.dst[q].C; st = p.dstart[q]; ac = p.dst[q].act; bq = p.dst[q].bsums; }
    g16[cq] = reinterpret_cast<unsigned short*>(gp); f16[cq] = reinterpret_cast<const unsigned short*>(fp);
    Cd[cq] = C; c0d[cq] = ct * 32 - st; dsl[cq] = act_slope(ac);
    bsd[cq] = ((cq == 0 ? act0 : act1) ? bq : nullptr);
  }
  const bool bs_on = bsd[0] != nullptr || bsd[1] != nullptr;       // wave-uniform
  int run_n = -1;
  float bs_s[2] = {0.f, 0.f}, bs_q[2] = {0.f, 0.f};
  const int bslot = (int)(blockIdx.x % PG_STAT_SLOTS);
  auto bs_flush = [&]() {
#pragma unroll
    for (int cq = 0; cq < 2; ++cq) {
      if (bsd[cq] == nullptr || run_n < 0) continue;
      const double ds = (double)owave_sum(bs_s[cq]), dq = (double)owave_sum(bs_q[cq]);
      if (lane == 0 && (ds != 0.0 || dq != 0.0)) {
        atomicAdd(&bsd[cq][((long)run_n * PG_STAT_SLOTS + bslot) * 2], ds);
        atomicAdd(&bsd[cq][((long)run_n * PG_STAT_SLOTS + bslot) * 2 + 1], dq);
      }
      bs_s[cq] = 0.f; bs_q[cq] = 0.f;
    }
  };
  char* const xt_ = x_lds + wave * 32 * OXP;
  const unsigned xb = (unsigned)(size_t)xt_;
  const int r4 = (lane >> 2) & 3, cql = lane & 3, mb = (lane >> 4) & 1;
  unsigned fa[2];
#pragma unroll
  for (int cq = 0; cq < 2; ++cq) fa[cq] = xb + (unsigned)((8 * lhi + r4) * OXP) + (unsigned)((((cq * 32) >> 3) + 2 * mb + (cql >> 1)) << 4) + ((cql & 1) << 3);
  char* const xq = xt_ + l31 * OXP + 8 * lhi;            // accumulator layout: this lane's pixel row, + (cq*32 + 8 g) * 2
  char* const xr = xt_ + (lane >> 2) * OXP + (lane & 3) * 16;      // row-major layout: + i * 16 * OXP + cq * 64

  f32x16 accw[2];
#pragma unroll
  for (int cq = 0; cq < 2; ++cq)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[cq][r] = 0.f;

  // the tile's forward values (row-major, [cq][i]) and this thread's patch elements (tid, tid + 256 < 306) — scalars, not arrays:
  // arrays written through a lambda stayed in scratch memory (144 bytes, every access a vmcnt round trip: 0.95 ms per launch)
  uint4 xin00, xin01, xin10, xin11;
  float pin0, pin1;             // raw loads; zeroed at PG_OCB_PUT_PATCH (a select right behind the load would wait for it)
  bool pok0 = false, pok1 = false;
  int pix0 = 0;
  const int pi0 = tid, pi1 = tid + 256;
  const int pc0 = pi0 / 102, prr0 = (pi0 - pc0 * 102) / 34, pxx0 = pi0 - pc0 * 102 - prr0 * 34;
  const int pc1 = pi1 / 102, prr1 = (pi1 - pc1 * 102) / 34, pxx1 = pi1 - pc1 * 102 - prr1 * 34;
  const bool p1_on = pi1 < 306;
  const int pslot0 = pc0 * 108 + prr0 * 36 + pxx0, pslot1 = p1_on ? pc1 * 108 + prr1 * 36 + pxx1 : 331;
  const unsigned xoff0 = (unsigned)(lane >> 2) * (unsigned)Cd[0] + (unsigned)(c0d[0] + (lane & 3) * 8);
  const unsigned xoff1 = (unsigned)(lane >> 2) * (unsigned)Cd[1] + (unsigned)(c0d[1] + (lane & 3) * 8);
#define PG_OCB_REQUEST(T)                                                                                          \
  do {                                                                                                             \
    const int row_ = (T) / tpr, xt_i = (T) - row_ * tpr;                                                           \
    const int n_ = row_ / p.H, y_ = row_ - n_ * p.H;                                                               \
    pix0 = row_ * p.W + xt_i * 32;                                                                                 \
    const unsigned e0_ = act0 ? (unsigned)pix0 * (unsigned)Cd[0] + xoff0 : 0u;                                     \
    const unsigned e1_ = act1 ? (unsigned)pix0 * (unsigned)Cd[1] + xoff1 : 0u;                                     \
    xin00 = *reinterpret_cast<const uint4*>(f16[0] + e0_);                                                         \
    xin01 = *reinterpret_cast<const uint4*>(f16[0] + e0_ + (act0 ? 16u * (unsigned)Cd[0] : 0u));                   \
    xin10 = *reinterpret_cast<const uint4*>(f16[1] + e1_);                                                         \
    xin11 = *reinterpret_cast<const uint4*>(f16[1] + e1_ + (act1 ? 16u * (unsigned)Cd[1] : 0u));                   \
    const int gy0_ = y_ - 1 + prr0, gx0_ = xt_i * 32 - 1 + pxx0;                                                   \
    const bool ok0_ = gy0_ >= 0 && gy0_ < p.H && gx0_ >= 0 && gx0_ < p.W;                                          \
    const float v0_ = p.dpre[ok0_ ? ((long)(n_ * 3 + pc0) * p.H + gy0_) * p.W + gx0_ : 0];                         \
    pin0 = v0_; pok0 = ok0_;                                                                                       \
    const int gy1_ = y_ - 1 + prr1, gx1_ = xt_i * 32 - 1 + pxx1;                                                   \
    const bool ok1_ = p1_on && gy1_ >= 0 && gy1_ < p.H && gx1_ >= 0 && gx1_ < p.W;                                 \
    const float v1_ = p.dpre[ok1_ ? ((long)(n_ * 3 + pc1) * p.H + gy1_) * p.W + gx1_ : 0];                         \
    pin1 = v1_; pok1 = ok1_;                                                                                       \
  } while (0)
#define PG_OCB_PUT_PATCH(BUF)                                        \
  do {                                                               \
    patch[BUF][pslot0] = pok0 ? pin0 : 0.f;                          \
    if (p1_on) patch[BUF][pslot1] = pok1 ? pin1 : 0.f;               \
  } while (0)

  int tile = blockIdx.x;           // host: gridDim.x <= ntiles
  PG_OCB_REQUEST(tile);
  __syncthreads();                 // weights, zeroed patches
  PG_OCB_PUT_PATCH(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int buf = 0;
  while (true) {
    const int cpix0 = pix0;
    if (bs_on) {                   // a tile = 32 pixels of one image row: one sample
      const int cn = cpix0 / HW;
      if (cn != run_n) { bs_flush(); run_n = cn; }
    }
    // ---- forward values -> the wave's LDS tile; both gradient operands from the patch
    *reinterpret_cast<uint4*>(xr) = xin00;
    *reinterpret_cast<uint4*>(xr + 16 * OXP) = xin01;
    *reinterpret_cast<uint4*>(xr + 64) = xin10;
    *reinterpret_cast<uint4*>(xr + 16 * OXP + 64) = xin11;
    // the next tile's global loads go out NOW (their registers are free again): a whole tile of work hides their latency, and
    // they precede this tile's stores
    tile += gridDim.x;
    const bool more = tile < ntiles;
    if (more) PG_OCB_REQUEST(tile);
    float gv[16], gt[16];
    const float* const pb = patch[buf];
#pragma unroll
    for (int q = 0; q < 16; ++q) gv[q] = pb[goff[q]];
#pragma unroll
    for (int q = 0; q < 16; ++q) gt[q] = pb[tbase + (q >> 3) * 16 + (q & 7)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint2 fw[8];
#pragma unroll
    for (int cq = 0; cq < 2; ++cq)
#pragma unroll
      for (int g = 0; g < 4; ++g) fw[cq * 4 + g] = *reinterpret_cast<const uint2*>(xq + (cq * 32 + 8 * g) * 2);
    const obf16x8 b0 = __builtin_bit_cast(obf16x8, make_uint4(opack_bf16(gv[0], gv[1]), opack_bf16(gv[2], gv[3]), opack_bf16(gv[4], gv[5]), opack_bf16(gv[6], gv[7])));
    const obf16x8 b1 = __builtin_bit_cast(obf16x8, make_uint4(opack_bf16(gv[8], gv[9]), opack_bf16(gv[10], gv[11]), opack_bf16(gv[12], gv[13]), opack_bf16(gv[14], gv[15])));
    const obf16x8 gb0 = __builtin_bit_cast(obf16x8, make_uint4(opack_bf16(gt[0], gt[1]), opack_bf16(gt[2], gt[3]), opack_bf16(gt[4], gt[5]), opack_bf16(gt[6], gt[7])));
    const obf16x8 gb1 = __builtin_bit_cast(obf16x8, make_uint4(opack_bf16(gt[8], gt[9]), opack_bf16(gt[10], gt[11]), opack_bf16(gt[12], gt[13]), opack_bf16(gt[14], gt[15])));
    // ---- data gradient
    uint2 res[8];
#pragma unroll
    for (int cq = 0; cq < 2; ++cq) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int ct = ct0 + cq;
      const obf16x8 a0 = __builtin_bit_cast(obf16x8, a_lds[((ct & 7) * 2 + 0) * 64 + lane]);
      const obf16x8 a1 = __builtin_bit_cast(obf16x8, a_lds[((ct & 7) * 2 + 1) * 64 + lane]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 f = fw[cq * 4 + g];
        const float f4[4] = {__uint_as_float(f.x << 16), __uint_as_float(f.x & 0xffff0000u), __uint_as_float(f.y << 16),
                             __uint_as_float(f.y & 0xffff0000u)};
        float r4v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r4v[e] = acc[4 * g + e] * act_grad_s(f4[e], dsl[cq]);
        res[cq * 4 + g] = make_uint2(opack_bf16(r4v[0], r4v[1]), opack_bf16(r4v[2], r4v[3]));
        if (bsd[cq] != nullptr) {      // (sum r, sum r * x_act): sums_mode 2 of pg_norm_bwd_apply_v2
          bs_s[cq] += (r4v[0] + r4v[1]) + (r4v[2] + r4v[3]);
          bs_q[cq] += fmaf(r4v[0], f4[0], fmaf(r4v[1], f4[1], fmaf(r4v[2], f4[2], r4v[3] * f4[3])));
        }
      }
    }
    // ---- weight gradient (k = the tile's 32 pixels)
    {
      unsigned long long x00l, x00h, x01l, x01h, x10l, x10h, x11l, x11h;      // x^T fragments [cq][k step] lo / hi
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:576\n\t"          // + 4, 16, 20 rows of OXP bytes
          "ds_read_b64_tr_b16 %2, %8 offset:2304\n\tds_read_b64_tr_b16 %3, %8 offset:2880\n\t"
          "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:576\n\t"
          "ds_read_b64_tr_b16 %6, %9 offset:2304\n\tds_read_b64_tr_b16 %7, %9 offset:2880\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(x00l), "=&v"(x00h), "=&v"(x01l), "=&v"(x01h), "=&v"(x10l), "=&v"(x10h), "=&v"(x11l), "=&v"(x11h)
          : "v"(fa[0]), "v"(fa[1])
          : "memory");
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      accw[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(obf16x8, (u64x2){x00l, x00h}), gb0, accw[0], 0, 0, 0);
      accw[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(obf16x8, (u64x2){x01l, x01h}), gb1, accw[0], 0, 0, 0);
      accw[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(obf16x8, (u64x2){x10l, x10h}), gb0, accw[1], 0, 0, 0);
      accw[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(obf16x8, (u64x2){x11l, x11h}), gb1, accw[1], 0, 0, 0);
    }
    // ---- the next tile's patch into the other buffer (its last readers passed the previous barrier) — BEFORE this tile's stores:
    // behind them (conditional on the channel tile being live) the compiler's vmcnt for these loads would also wait for stores
    if (more) { if (buf) PG_OCB_PUT_PATCH(0); else PG_OCB_PUT_PATCH(1); }
    // ---- results back through the LDS tile (every read of the forward values is done: the asm above waited), row-major out
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int cq = 0; cq < 2; ++cq)
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(xq + (cq * 32 + 8 * g) * 2) = res[cq * 4 + g];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const uint4 o00 = *reinterpret_cast<const uint4*>(xr), o01 = *reinterpret_cast<const uint4*>(xr + 16 * OXP);
      const uint4 o10 = *reinterpret_cast<const uint4*>(xr + 64), o11 = *reinterpret_cast<const uint4*>(xr + 16 * OXP + 64);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (act0) {
        const unsigned e = (unsigned)cpix0 * (unsigned)Cd[0] + xoff0;
        *reinterpret_cast<uint4*>(g16[0] + e) = o00;
        *reinterpret_cast<uint4*>(g16[0] + e + 16u * (unsigned)Cd[0]) = o01;
      }
      if (act1) {
        const unsigned e = (unsigned)cpix0 * (unsigned)Cd[1] + xoff1;
        *reinterpret_cast<uint4*>(g16[1] + e) = o10;
        *reinterpret_cast<uint4*>(g16[1] + e + 16u * (unsigned)Cd[1]) = o11;
      }
    }
    if (!more) break;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    buf ^= 1;
  }
  if (bs_on) bs_flush();
  // ---- this workgroup's weight-gradient partial: accw[cq][r] = dW[t = l31][ci = (ct0 + cq) * 32 + 8 (r >> 2) + 4 lhi + (r & 3)]
  if (l31 < ODG_T) {
    float* const dstp = p.wpart + (long)blockIdx.x * p.Ctot * 28;
#pragma unroll
    for (int cq = 0; cq < 2; ++cq)
      if (cq == 0 ? act0 : act1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dstp[((ct0 + cq) * 32 + 8 * (r >> 2) + 4 * lhi + (r & 3)) * 28 + l31] = accw[cq][r];
      }
  }
}

// dW[t][ci] += sum over workgroup partials [nb][Ctot][28]; blockIdx.y = slice of the partial list (float atomics into dW: one
// serial walk over 1024 partials per thread took 160 us at batch 4)
__global__ __launch_bounds__(256) void out_conv_wgrad_reduce_kernel(const float* part, int nb, int Ctot, float* dW) {
  const int i = blockIdx.x * 256 + threadIdx.x;       // (ci, t)
  if (i >= Ctot * 28) return;
  const int ci = i / 28, t = i - ci * 28;
  if (t >= ODG_T) return;
  const int per = (nb + (int)gridDim.y - 1) / (int)gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nb, b0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = b0;
  for (; b + 3 < b1; b += 4) {
    s0 += part[(long)b * Ctot * 28 + i]; s1 += part[(long)(b + 1) * Ctot * 28 + i];
    s2 += part[(long)(b + 2) * Ctot * 28 + i]; s3 += part[(long)(b + 3) * Ctot * 28 + i];
  }
  for (; b < b1; ++b) s0 += part[(long)b * Ctot * 28 + i];
  if (b1 > b0) atomicAdd(&dW[(long)t * Ctot + ci], (s0 + s1) + (s2 + s3));
}

}  // namespace pg

extern "C" int pg_out_conv_dgrad(const float* G, const float* Wt, int32_t N, int32_t H, int32_t W,
                                 const pg_dst_t* dst, int32_t ndst, void* stream) {
  PG_REQUIRE(G && Wt && dst && ndst >= 1 && ndst <= PG_MAX_SRC && N > 0 && H > 0 && W > 0, "pg_out_conv_dgrad: bad arguments");
  pg::OutDgradK k;
  memset(&k, 0, sizeof(k));
  k.G = G; k.Wt = Wt;
  int c = 0;
  for (int j = 0; j < ndst; ++j) {
    k.dst[j] = dst[j]; k.dstart[j] = c; c += dst[j].C;
    PG_REQUIRE(dst[j].C % 4 == 0 && ((size_t)dst[j].grad & 15) == 0 && ((size_t)dst[j].fwd & 15) == 0 &&
               ((size_t)dst[j].mask & 15) == 0, "pg_out_conv_dgrad: destinations need C %% 4 == 0 and 16-byte alignment");
  }
  for (int j = ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
  k.ndst = ndst; k.Ctot = c;
  PG_REQUIRE(c <= 256, "pg_out_conv_dgrad: at most 256 input channels (one wave per pixel), got %d", c);
  PG_REQUIRE((double)N * H * W < 2147483648.0 / 32, "pg_out_conv_dgrad: too many pixels");
  k.npix = N * H * W; k.ppix = H * W;
  long blocks = (k.npix + 15) / 16;
  if (blocks > 256 * 12) blocks = 256 * 12;     // 108 weight registers per lane are loaded once per workgroup
  int nbf = 0;
  for (int j = 0; j < ndst; ++j) {
    const int want = PG_DST_GRAD_BF16 | (dst[j].fwd ? PG_DST_FWD_BF16 : 0);
    if (dst[j].flags != 0) { PG_REQUIRE((dst[j].flags & want) == want, "pg_out_conv_dgrad: a destination mixes fp32 and bf16 tensors"); ++nbf; }
  }
  PG_REQUIRE(nbf == 0 || nbf == ndst, "pg_out_conv_dgrad: destinations must be all fp32 or all bf16");
  // (round 4) fused norm-backward sums: the fp32 form of the pass, every carrying destination with a forward tensor
  bool bs = nbf == 0;
  int nbs = 0;
  for (int j = 0; j < ndst; ++j)
    if (dst[j].bsums != nullptr) { ++nbs; bs = bs && dst[j].fwd != nullptr; }
  bs = bs && nbs > 0;
  if (!bs)
    for (int j = 0; j < PG_MAX_SRC; ++j) k.dst[j].bsums = nullptr;
  pg::last_info() = bs ? PG_INFO_BSUMS : 0;
  if (bs) {
    long bx = (k.ppix + 15) / 16;
    const long cap = (256 * 12 + N - 1) / N;
    if (bx > cap) bx = cap;
    PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, false, false, true>), dim3((unsigned)bx, (unsigned)N), dim3(256), 0, (hipStream_t)stream, k);
  } else if (nbf) PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  else PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_out_conv_dgrad");
  return 0;
}

// Data gradient AND weight gradient of the output convolution (bf16 STORAGE on the bf16 data path), two streaming passes
// over the same descriptors: every destination's `fwd` must be the ACTIVATED bf16 operand of the forward pass (no aff /
// mask on it), gradients are bf16, dW is [27][Ctot] fp32 (accumulated), `workspace` holds workspace_floats >= 64 * Ctot * 28
// floats of per-workgroup partials.
extern "C" int pg_out_conv_bwd_direct(const float* G, int32_t g_is_dpre, const float* Wt, int32_t N, int32_t H, int32_t W,
                                      const pg_dst_t* dst, int32_t ndst, float* dW, float* workspace, int64_t workspace_floats,
                                      void* wg_stream, void* stream);
extern "C" int pg_out_conv_dgrad_wgrad(const float* G, const float* Wt, int32_t N, int32_t H, int32_t W, const pg_dst_t* dst,
                                       int32_t ndst, float* dW, float* workspace, int64_t workspace_floats, void* stream) {
  return pg_out_conv_bwd_direct(G, 0, Wt, N, H, W, dst, ndst, dW, workspace, workspace_floats, nullptr, stream);
}

// g_is_dpre = 1: `G` is the NCHW (N,3,H,W) gradient wrt the pre-tanh output itself; the kernels gather each pixel's 27 (tap,
// channel) values from it (no pg_im2col_taps pass, no [pixel][32] tensor).  wg_stream: stream of the weight-gradient pass (NULL:
// `stream`); the caller orders it after the producer of G.
extern "C" int pg_out_conv_bwd_direct(const float* G, int32_t g_is_dpre, const float* Wt, int32_t N, int32_t H, int32_t W,
                                      const pg_dst_t* dst, int32_t ndst, float* dW, float* workspace, int64_t workspace_floats,
                                      void* wg_stream, void* stream) {
  PG_REQUIRE(G && Wt && dst && dW && workspace && ndst >= 1 && ndst <= PG_MAX_SRC && N > 0 && H > 0 && W > 0,
             "pg_out_conv_dgrad_wgrad: bad arguments");
  pg::OutDgradK k;
  memset(&k, 0, sizeof(k));
  k.G = G; k.Wt = Wt;
  if (g_is_dpre) { k.dpre = G; k.H = H; k.W = W; }
  int c = 0;
  for (int j = 0; j < ndst; ++j) {
    k.dst[j] = dst[j]; k.dstart[j] = c; c += dst[j].C;
    PG_REQUIRE(dst[j].C % 4 == 0 && ((size_t)dst[j].grad & 15) == 0 && ((size_t)dst[j].fwd & 15) == 0 && dst[j].fwd != nullptr &&
               dst[j].aff == nullptr && dst[j].mask == nullptr && dst[j].flags == (PG_DST_GRAD_BF16 | PG_DST_FWD_BF16),
               "pg_out_conv_dgrad_wgrad: destinations need C %% 4 == 0, 16-byte alignment, bf16 tensors and an activated forward operand");
  }
  for (int j = ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
  k.ndst = ndst; k.Ctot = c;
  PG_REQUIRE(c <= 256, "pg_out_conv_dgrad_wgrad: at most 256 input channels (one wave per pixel), got %d", c);
  PG_REQUIRE((double)N * H * W < 2147483648.0 / 32, "pg_out_conv_dgrad_wgrad: too many pixels");
  k.npix = N * H * W; k.ppix = H * W;
  long blocks = (k.npix + 15) / 16;
  if (blocks > 256 * 12) blocks = 256 * 12;
  hipStream_t st = (hipStream_t)stream, wst = wg_stream ? (hipStream_t)wg_stream : st;
  bool mfma = g_is_dpre && W % 32 == 0 && c % 32 == 0 && (double)N * H * W * 256 < 2147483648.0 && getenv("PG_NO_OUT_DGRAD_MFMA") == nullptr;
  for (int j = 0; j < ndst; ++j) mfma = mfma && dst[j].C % 32 == 0 && dst[j].accumulate == 0;
  static const bool fuse = getenv("PG_NO_OUT_BWD_FUSED") == nullptr;
  if (mfma && fuse && workspace_floats / ((long)c * 28) >= 64) {
    // ONE pass: data gradient + weight gradient on the matrix cores (out_conv_bwd_mfma_kernel), on `stream`
    long cap = workspace_floats / ((long)c * 28);
    long fb = (long)k.npix / 32;
    static const long fcap = getenv("PG_OUT_BWD_WGS") ? atol(getenv("PG_OUT_BWD_WGS")) : 512;      // persistent: two workgroups per CU
    if (fb > fcap) fb = fcap;
    if (fb > cap) fb = cap;
    k.wpart = workspace;
    PG_KLAUNCH(pg::out_conv_bwd_mfma_kernel, dim3((unsigned)fb), dim3(256), 0, st, k);
    PG_LAUNCH_OK("pg_out_conv_bwd_direct (fused MFMA pass)");
    PG_KLAUNCH(pg::out_conv_wgrad_reduce_kernel, dim3((c * 28 + 255) / 256, pg::deterministic() ? 1 : 16), dim3(256), 0, st, workspace, (int)fb, c, dW);
    PG_LAUNCH_OK("pg_out_conv_bwd_direct (reduce)");
    {
      bool bs = false;
      for (int j = 0; j < ndst; ++j) bs = bs || dst[j].bsums != nullptr;
      pg::last_info() = bs ? PG_INFO_BSUMS : 0;       // the fused pass filled pg_dst_t.bsums (sums_mode 2)
    }
    // the fused pass wrote dW on `stream`; callers (and the data-parallel reducer: runtime/dp.py orders a gradient range
    // against the weight-gradient stream only) treat dW as a product of `wg_stream` — make that true
    if (wg_stream != nullptr && wst != st) return pg_stream_wait(wg_stream, stream);
    return 0;
  }
  pg::last_info() = 0;             // streaming passes: pg_dst_t.bsums is not filled
  if (g_is_dpre) PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, k);
  else PG_KLAUNCH((pg::out_conv_dgrad_kernel<true, false, true, false>), dim3((unsigned)blocks), dim3(256), 0, st, k);
  PG_LAUNCH_OK("pg_out_conv_dgrad_wgrad (data gradient)");
  blocks = (k.npix + 15) / 16;
  long cap = workspace_floats / ((long)c * 28);
  if (cap > 1024) cap = 1024;
  PG_REQUIRE(cap >= 64, "pg_out_conv_dgrad_wgrad: workspace too small");
  if (blocks > cap) blocks = cap;
  k.wpart = workspace;
  if (g_is_dpre) PG_KLAUNCH((pg::out_conv_dgrad_kernel<false, true, true, true>), dim3((unsigned)blocks), dim3(256), 0, wst, k);
  else PG_KLAUNCH((pg::out_conv_dgrad_kernel<false, true, true, false>), dim3((unsigned)blocks), dim3(256), 0, wst, k);
  PG_LAUNCH_OK("pg_out_conv_dgrad_wgrad (weight gradient)");
  PG_KLAUNCH(pg::out_conv_wgrad_reduce_kernel, dim3((c * 28 + 255) / 256, pg::deterministic() ? 1 : 16), dim3(256), 0, wst, workspace, (int)blocks, c, dW);
  PG_LAUNCH_OK("pg_out_conv_dgrad_wgrad (reduce)");
  return 0;
}

// Weight gradient of the output convolution alone (bf16 data path, round 3): dW[27][Ctot] += sum_pixels G[pixel][t] * x[pixel][ci]
// with G the bf16 im2col'd gradient (row pitch g_pitch, pg_im2col_taps_bf16) and x the ACTIVATED bf16 operands of the forward
// pass, passed as the `fwd` tensors of `dst` (grad / aff / mask unused).  The data gradient of the same layer runs as a
// bf16 contraction (pg_conv with the padded weight).
extern "C" int pg_out_conv_wgrad_bf16(const void* G_bf16, int32_t g_pitch, int32_t N, int32_t H, int32_t W, const pg_dst_t* dst,
                                      int32_t ndst, float* dW, float* workspace, int64_t workspace_floats, void* stream) {
  PG_REQUIRE(G_bf16 && dst && dW && workspace && ndst >= 1 && ndst <= PG_MAX_SRC && N > 0 && H > 0 && W > 0 && g_pitch >= 32,
             "pg_out_conv_wgrad_bf16: bad arguments");
  pg::OutDgradK k;
  memset(&k, 0, sizeof(k));
  k.G = reinterpret_cast<const float*>(G_bf16); k.Wt = nullptr; k.g_bf16_pitch = g_pitch;
  int c = 0;
  for (int j = 0; j < ndst; ++j) {
    k.dst[j] = dst[j]; k.dstart[j] = c; c += dst[j].C;
    k.dst[j].grad = const_cast<float*>(dst[j].fwd);          // never written (DG = false); keeps the dummy addresses valid
    k.dst[j].accumulate = 0;
    PG_REQUIRE(dst[j].C % 4 == 0 && ((size_t)dst[j].fwd & 15) == 0 && dst[j].fwd != nullptr && dst[j].aff == nullptr &&
               dst[j].mask == nullptr && (dst[j].flags & PG_DST_FWD_BF16),
               "pg_out_conv_wgrad_bf16: sources need C %% 4 == 0, 16-byte alignment and an activated bf16 operand");
  }
  for (int j = ndst; j <= PG_MAX_SRC; ++j) k.dstart[j] = c;
  k.ndst = ndst; k.Ctot = c;
  PG_REQUIRE(c <= 256, "pg_out_conv_wgrad_bf16: at most 256 input channels (one wave per pixel), got %d", c);
  PG_REQUIRE((double)N * H * W < 2147483648.0 / 64, "pg_out_conv_wgrad_bf16: too many pixels");
  k.npix = N * H * W; k.ppix = H * W;
  long blocks = (k.npix + 15) / 16;
  long cap = workspace_floats / ((long)c * 28);
  if (cap > 1024) cap = 1024;
  PG_REQUIRE(cap >= 64, "pg_out_conv_wgrad_bf16: workspace too small");
  if (blocks > cap) blocks = cap;
  k.wpart = workspace;
  PG_KLAUNCH((pg::out_conv_dgrad_kernel<false, true, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, k);
  PG_LAUNCH_OK("pg_out_conv_wgrad_bf16");
  PG_KLAUNCH(pg::out_conv_wgrad_reduce_kernel, dim3((c * 28 + 255) / 256, pg::deterministic() ? 1 : 16), dim3(256), 0, (hipStream_t)stream, workspace,
                     (int)blocks, c, dW);
  PG_LAUNCH_OK("pg_out_conv_wgrad_bf16 (reduce)");
  return 0;
}
