/*
 * posegan_hip.h — C ABI of libposegan_hip.so: the MI355X (gfx950) kernels behind the
 * Deformable-GAN training step.
 *
 * The reference (saurabhsharma1993/pose-transfer) has no FFI: its seam is the ATen op set its
 * Python modules dispatch.  Each entry point below names the reference call site(s) it replaces
 * (paths relative to /root/reference/src_deformable).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error (message: pg_last_error(), thread-local);
 *   - all pointers are DEVICE pointers unless stated; buffers are borrowed (caller-owned, must outlive
 *     the stream work); the library allocates nothing persistent;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is stream-ordered, no
 *     host synchronisation happens inside any call;
 *   - activations are fp32 NHWC ("pixel-major": [N][H][W][C]) unless a stride quadruple is given;
 *   - conv / conv-transpose weights are fp32 packed [KH][KW][Cout][Cin] (host packs once from the
 *     reference's OIHW / IOHW state_dict layouts).
 */
#ifndef POSEGAN_HIP_H
#define POSEGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_MAX_SRC 4
#define PG_ACT_NONE 0
#define PG_ACT_RELU 1   /* nn.ReLU         models/networks.py:152 */
#define PG_ACT_LEAKY 2  /* nn.LeakyReLU(0.2) models/networks.py:150 */
#define PG_OUT_NONE 0
#define PG_OUT_TANH 1   /* nn.Tanh         models/networks.py:232 */

/* One K-operand source of a (virtually concatenated) convolution input: replaces torch.cat
 * (models/networks.py:241,245,271,284,286; models/pose_gan.py:86,133,135) — never materialised.
 * value(n,c,y,x) = act( (a_n * raw + b_n) * mask[n][c] ), zero outside the image.            */
typedef struct {
  const float* ptr;   /* raw tensor                                                          */
  int32_t C;          /* channels of this source                                             */
  int32_t _pad;
  const float* aff;   /* [N][2] per-sample (a,b): the deferred per-sample norm; NULL = (1,0) */
  const float* mask;  /* [N][C] channel-dropout multipliers (Dropout2d); NULL = none         */
  int64_t sN, sC, sH, sW; /* element strides; used only when the conv is run in scalar mode  */
} pg_src_t;

/* One destination of a data-gradient epilogue (the gradient of the concat is split back):
 * grad[n,y,x,c] (+)= g * mask[n][c] * act'( (a_n*fwd + b_n) * mask[n][c] )                   */
typedef struct {
  float* grad;        /* NHWC [N][Ho][Wo][C]                                                 */
  const float* fwd;   /* raw forward tensor the activation derivative is evaluated on, or NULL */
  const float* aff;   /* [N][2] or NULL                                                      */
  const float* mask;  /* [N][C] or NULL                                                      */
  int32_t C;
  int32_t act;        /* PG_ACT_* of the consumer that read `fwd`                            */
  int32_t accumulate; /* 1: grad += ...; 0: grad = ...                                       */
  int32_t flags;      /* PG_DST_*: bf16 STORAGE of `grad` / `fwd` (bf16 data path, round 3); 0 = fp32 */
  double* bsums;      /* optional (round 4), [N][PG_STAT_SLOTS][2], zeroed by the caller: the epilogue that writes the FINAL value r of
                       * this gradient also adds the two per-sample sums of the following norm backward — (sum r, sum r * fwd) —
                       * so that pg_norm_bwd_reduce's pass over the tensor does not run (pg_norm_bwd_apply_v2 converts them).
                       * Implemented by pg_conv's data-gradient epilogues (the bf16 256-row kernels; the generic kernels' row-major
                       * scatter when a column tile lies in one destination and a sample has >= 64 rows; the split-K fix-up pass),
                       * pg_out_conv_bwd_direct's fused pass (sums_mode 2) and pg_out_conv_dgrad (fp32).  Whether a launch did is
                       * reported in pg_last_launch_info() bit 14 (PG_INFO_BSUMS) — otherwise the field is ignored           */
} pg_dst_t;
#define PG_INFO_BSUMS (1 << 14)
#define PG_DST_GRAD_BF16 1
#define PG_DST_FWD_BF16 2

/* Implicit-GEMM convolution, fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   mode 0 "down": out[n,oy,ox,:] = sum_{r,s} W[r][s] . in[n, oy*stride + r - pad, ox*stride + s - pad, :]
 *                  = nn.Conv2d forward (models/networks.py:154,186,228,341) and the data-gradient of
 *                    nn.ConvTranspose2d+Cropping2D(1) (models/networks.py:156-157).
 *   mode 1 "up":   out[n,Y,X,:]   = sum_{r,s: (Y+pad-r)%stride==0} W[r][s] . in[n,(Y+pad-r)/stride,(X+pad-s)/stride,:]
 *                  = nn.ConvTranspose2d(k4,s2)+crop forward (pad=1) and the data-gradient of nn.Conv2d.
 *   w_transposed 0: GEMM-N runs over the packed weight's Cout, K over its Cin (forward passes);
 *                1: GEMM-N over Cin, K over Cout (data-gradient passes); n_off/n_cnt select a sub-range of N.
 *   epilogue 0: out (+bias)(tanh) stored through (oN,oC,oH,oW) strides;
 *   epilogue 1: data-gradient split over `dst[]` with fused activation derivative / dropout mask.  */
typedef struct {
  pg_src_t src[PG_MAX_SRC];
  int32_t nsrc, N, Hi, Wi;
  int32_t act;              /* PG_ACT_* applied to the operand after affine+mask                   */
  int32_t scalar_in;        /* 1: sources are small-C strided tensors (e.g. NCHW inputs)            */
  int32_t mode, KH, KW, stride, pad, Ho, Wo;
  int32_t w_transposed;
  const float* W;           /* packed [KH][KW][wCout][wCin]                                        */
  int32_t wCout, wCin;
  int32_t n_off, n_cnt;     /* GEMM-N sub-range (0, full) in the common case                       */
  int32_t epilogue;
  int32_t out_act;          /* PG_OUT_*                                                            */
  float* out;
  const float* bias;        /* [n_cnt] or NULL                                                     */
  int64_t oN, oC, oH, oW;   /* output element strides (NHWC: Ho*Wo*C, 1, Wo*C, C)                  */
  pg_dst_t dst[PG_MAX_SRC];
  int32_t ndst;
  int32_t ksplit;           /* 0 = auto; >1 splits K across workgroups (atomic accumulate)         */
  int32_t precision;        /* PG_PREC_*: MFMA operand format (accumulation stays fp32)              */
  int32_t out_bf16;         /* 1: `out` (epilogue 0, dense NHWC, no out_act) is a bf16 tensor — bf16 STORAGE of the raw
                               convolution output on the bf16 data path; the fused statistics still come from the fp32
                               accumulators.  Needs the workspace for split-K launches.                            */
  double* stats;            /* optional [N][PG_STAT_SLOTS][2], caller-zeroed: per-sample (sum, sum of squares) of the stored output =
                               pg_norm_stats of `out` (the following per-sample norm, models/networks.py:159), fused
                               into the epilogue when the launch is not split-K; epilogue 0, dense NHWC only        */
  void* workspace;          /* optional scratch for split-K launches (16-byte aligned): when it holds ksplit x the output,
                               every split stores its partial tile there and one fix-up kernel reduces them and applies
                               the epilogue (bias / statistics, or the data-gradient scatter); otherwise split-K adds
                               into the zero-filled destination with float atomics (slower, order-dependent rounding) */
  int64_t workspace_bytes;
} pg_conv_t;

/* MFMA operand precision of pg_conv.  F32 is the reference-parity path and the default everywhere; BF16X3 splits
   each fp32 operand into two bf16 (hi+lo, 3 MFMAs per product, ~2^-16 relative product error); BF16 rounds operands
   to bf16 (mixed precision, NOT a parity mode).  Layers the low-precision kernels do not cover run in fp32. */
#define PG_PREC_F32 0
#define PG_PREC_BF16 1
#define PG_PREC_BF16X3 2
#define PG_PREC_BF16_DATA 3   /* src[].ptr and W point to bf16 TENSORS (pg_materialise_bf16 / pg_weights_to_bf16): no
                               prologue, w_transposed must be 0, every source C % 64 == 0; fp32 accumulate + outputs */

int pg_conv(const pg_conv_t* desc, void* stream);

/* bf16 data path helpers.  materialise: out[n,p,c] = bf16(act((aff[n].a * x + aff[n].b) * mask[n,c])) over an NHWC fp32
 * tensor (aff / mask may be NULL) — the pre-activation input of the reference's Block (networks.py:142-172) written
 * once, so that the contraction can DMA its tiles.  weights_to_bf16: packed fp32 [taps][Cout][Cin] -> bf16 in the same
 * layout (`nt`, forward operand) and/or per-tap transposed [taps][Cin][Cout] (`t`, data-gradient operand). */
int pg_materialise_bf16(const float* x, const float* aff, const float* mask, int32_t act, int32_t N, int64_t HW,
                        int32_t C, void* out_bf16, void* stream);
int pg_weights_to_bf16(const float* W, int32_t taps, int32_t Cout, int32_t Cin, void* nt_bf16, void* t_bf16,
                       void* stream);
/* Batched NT GEMM on the bf16 data path: out[t][m][n] = sum_k A[m][a_off[t]+k] * B[n][b_off[t]+k] (A [M][K], B [N][K]
 * bf16, row pitch K, K % 64 == 0; offsets in elements, any parity / sign as long as the reads stay inside the buffers).
 * With A = channel-major gradient, B = channel-major activated input (pg_channel_major_bf16) and one offset per filter
 * tap this is the weight gradient of a Block convolution (autograd of networks.py:154-157 wrt weight). */
/* Channel-major, zero-bordered bf16 image of an NHWC fp32 tensor (sub = 1), or of one stride-2 phase plane (sub = 2,
 * parity (py, px)): out[c][(n*(Hq+2) + yy)*Wp + xx] = bf16(act((a*x+b)*mask)) at source pixel (sub*(yy-1)+py,
 * sub*(xx-1)+px), zero on the border / padding / tail; channel rows K elements apart (K % 64 == 0, Wp % 8 == 0). */
int pg_channel_major_bf16(const float* x, const float* aff, const float* mask, int32_t act, int32_t N, int32_t H,
                          int32_t W, int32_t C, int32_t sub, int32_t py, int32_t px, int32_t Hq, int32_t Wq,
                          int32_t Wp, int64_t K, void* out_bf16, void* stream);
int pg_gemm_taps_bf16(const void* A, const void* B, int32_t M, int32_t N, int32_t K, int32_t gtaps,
                      const int64_t* a_off, const int64_t* b_off, float* out, void* stream);

/* Weight gradient of the same relation (torch autograd of conv2d / conv_transpose2d wrt weight):
 *   dW[r][s][co][ci] += sum_{n,qy,qx} dY_small/large[...,co] * X_large/small[...,ci]
 * with small pixel (qy,qx) <-> large pixel (qy*stride + r - pad, qx*stride + s - pad).
 *   x_is_large 1: X is the large tensor (nn.Conv2d); 0: X is the small one (nn.ConvTranspose2d+crop).
 * X is a virtual concat of `src[]` with the same fused prologue as the forward; dW must be
 * zero-initialised by the caller (it is accumulated, split-K uses float atomics).              */
typedef struct {
  pg_src_t src[PG_MAX_SRC];
  int32_t nsrc, N;
  int32_t act, scalar_x;
  const float* dY;          /* NHWC [N][Hy][Wy][Cout] (or strided when scalar_y)                   */
  int64_t yN, yC, yH, yW;   /* dY strides (used when scalar_y)                                     */
  int32_t scalar_y;
  int32_t x_is_large;
  int32_t Hs, Ws, Hl, Wl;   /* small / large spatial sizes                                         */
  int32_t KH, KW, stride, pad;
  float* dW;                /* packed [KH][KW][Cout][Cin]                                          */
  int32_t Cout, Cin;
  int32_t ksplit;           /* 0 = auto                                                            */
  int32_t cout_store;       /* 0 = Cout; else only rows co < cout_store of dW (row stride Cin) are written    */
} pg_wgrad_t;

int pg_conv_wgrad(const pg_wgrad_t* desc, void* stream);

/* ---- small-N edge layers (the generator's 256->3 output convolution, models/networks.py:228), re-associated so
 * that no MFMA tile is 29/32 empty (csrc/edge.hip):
 * tap_gather:  out[n,co,y,x] = act(bias[co] + sum_{r,s} Y[n,y+r-pad,x+s-pad,(r*KW+s)*Co+co]) after a 1x1 pg_conv;
 * im2col_taps: G[n,y,x,(r*KW+s)*C+c] = dY[n,c,y-(r-pad),x-(s-pad)], zero padded to Cpad channels (wgrad operand);
 * small_cout_dgrad: dX[p][ci] = sum_{tap,co} dY[p-off(tap)][co]*W[tap][co][ci], split over dst[] like pg_conv's
 *              data-gradient epilogue (K = taps*Co is tiny: a streaming kernel); p-off(tap) = (p + pad - tap)/stride
 *              where that is an integer inside dY.  Used for the discriminator's 512->1 output convolution
 *              (networks.py:346, k4 s2 p1); the generator's 256->3 output convolution uses pg_out_conv_dgrad.     */
int pg_tap_gather(const float* Y, int32_t N, int32_t H, int32_t W, int32_t KH, int32_t KW, int32_t pad, int32_t Co,
                  const float* bias, int32_t out_act, float* out, int64_t oN, int64_t oC, int64_t oH, int64_t oW,
                  void* stream);
int pg_im2col_taps(const float* dY, int64_t yN, int64_t yC, int64_t yH, int64_t yW, int32_t N, int32_t H, int32_t W,
                   int32_t KH, int32_t KW, int32_t pad, int32_t C, int32_t Cpad, float* G, void* stream);
int pg_small_cout_dgrad(const float* dY, int64_t yN, int64_t yC, int64_t yH, int64_t yW, int32_t N, int32_t H,
                        int32_t W, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t Co, const float* Wt,
                        const pg_dst_t* dst, int32_t ndst, void* stream);

/* data-gradient of the 256->3 output convolution from the im2col'd gradient G [N*H*W][32] (pg_im2col_taps, Cpad 32)
 * and the weight viewed as Wt [Cin][32] ((tap, co) per input channel, 27 used): dX[p][ci] = sum_t G[p][t] * Wt[ci][t],
 * scattered over dst[] with act'(fwd) / mask like pg_conv's data-gradient epilogue.  Streaming kernel, one wave per
 * pixel, Cin = sum dst[].C <= 256, every dst[].C % 4 == 0 (csrc/out_conv_dgrad.hip). */
int pg_out_conv_dgrad(const float* G, const float* Wt, int32_t N, int32_t H, int32_t W, const pg_dst_t* dst,
                      int32_t ndst, void* stream);

/* first-layer convolutions with few NCHW input channels and 64 outputs (models/networks.py:186 k3 s1 p1; :341 k4 s2 p0):
 * the input patch of an 8x16 output tile is staged in LDS and feeds the MFMA directly (csrc/edge.hip).
 * `Wt` = pg_repack_small_cin(W packed [KH][KW][64][Cin]) -> [Cin][KH*KW][64]; sources use their (sN,sC,sH,sW) strides. */
int pg_repack_small_cin(const float* W, int32_t KH, int32_t KW, int32_t Cout, int32_t Cin, float* Wt, void* stream);
int pg_small_cin_conv(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                      int32_t pad, const float* Wt, const float* bias, float* out, void* stream);
/* their weight gradient (autograd of the same two layers): dW packed [KH][KW][64][Cin] += sum over pixels of
 * dY[n,oy,ox,co] * x[n,ci,oy*stride+r-pad,ox*stride+s-pad]; dY NHWC [N][Ho][Wo][64].  All taps share one pass over dY,
 * the input patch of a pixel tile is gathered from LDS (csrc/small_cin_wgrad.hip).  k3s1: Cin <= 35; k4s2: Cin <= 88.
 * `workspace` (optional, PG_SMALL_CIN_WGRAD_WS floats cover every supported shape): per-workgroup partial results that
 * a second kernel reduces; without it the workgroups add into dW with float atomics (slower). */
#define PG_SMALL_CIN_WGRAD_WS (768L * 64 * 704)
int pg_small_cin_wgrad(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                       int32_t pad, const float* dY, float* dW, float* workspace, int64_t workspace_floats,
                       void* stream);

/* The same two first layers on the bf16 data path (csrc/stem_bf16.hip): operands rounded to bf16 (RNE), fp32
 * accumulation, v_mfma_f32_32x32x16_bf16.  The input patch of a pixel tile is staged once, channel-last, as bf16; the
 * forward reads its A operand straight out of the patch, the weight gradient reads both operands with the transposing
 * LDS read.  Reference: models/networks.py:186 (k3 s1 p1) and :341 (k4 s2 p0) and their autograd.
 *   pg_stem_pack_elems(K, Cin)  : number of bf16 elements of the packed filter
 *   pg_stem_pack_bf16           : W packed fp32 [K][K][64][Cin] -> Wp (groups of 24 / 40 channels, zero padded)
 *   pg_stem_conv_bf16           : out NHWC [N][Ho][Wo][64] fp32 = conv(x) + bias;  Cin <= 80
 *   pg_stem_wgrad_bf16          : dW packed [K][K][64][Cin] += ...; k3: Cin <= 36, k4: Cin <= 72; `workspace` is REQUIRED
 *                                 (PG_SMALL_CIN_WGRAD_WS floats cover every shape; fewer -> fewer persistent workgroups). */
int64_t pg_stem_pack_elems(int32_t K, int32_t Cin);
int pg_stem_pack_bf16(const float* W, int32_t K, int32_t Cin, uint16_t* Wp, void* stream);
int pg_stem_conv_bf16(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                      int32_t pad, const uint16_t* Wp, const float* bias, float* out, void* stream);
/* the same; `out_bf16` (NULL = none): also write bf16(act(out)) NHWC — the activated operand the next layer's contraction
 * reads on the bf16 data path (saves its pg_materialise_bf16 pass over `out`); act = PG_ACT_* */
int pg_stem_conv_bf16_ex(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                         int32_t pad, const uint16_t* Wp, const float* bias, float* out, uint16_t* out_bf16, int32_t act,
                         void* stream);
int pg_stem_wgrad_bf16(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                       int32_t pad, const float* dY, float* dW, float* workspace, int64_t workspace_floats, void* stream);

/* data-gradient of a first-layer convolution (64 output channels, dY NHWC) towards `nc` <= 4 of its input channels
 * [c_off, c_off + nc), written (not accumulated) through (oN,oC,oH,oW) element strides — d loss / d generated image through
 * the discriminator's stem in gen_update (reference models/pose_gan.py:95-98; autograd of networks.py:341), and the
 * stage-to-stage chain of the stacked generator (networks.py:186).  W packed [K][K][64][Cin]. */
int pg_small_cin_dgrad(const float* dY, const float* W, int32_t N, int32_t Ho, int32_t Wo, int32_t K, int32_t stride,
                       int32_t pad, int32_t Hi, int32_t Wi, int32_t Cin, int32_t c_off, int32_t nc, float* out,
                       int64_t oN, int64_t oC, int64_t oH, int64_t oW, void* stream);
/* (round 6) the same with dY in bf16 STORAGE (io_flags bit 0): the stage-to-stage chain of the stacked generator on the bf16 data path
 * (reference models/networks.py:306-327: stage i's image input is stage i-1's output; autograd of the first Conv2d, networks.py:186). */
int pg_small_cin_dgrad_io(const void* dY, const float* W, int32_t N, int32_t Ho, int32_t Wo, int32_t K, int32_t stride,
                          int32_t pad, int32_t Hi, int32_t Wi, int32_t Cin, int32_t c_off, int32_t nc, float* out, int64_t oN,
                          int64_t oC, int64_t oH, int64_t oW, int32_t io_flags, void* stream);

/* db[c] += sum over rows of a strided [rows][C] view (bias gradient; torch autograd of conv bias). */
int pg_bias_grad(const float* dY, int64_t rows_outer, int64_t rows_inner, int32_t C, int64_t s_outer,
                 int64_t s_inner, int64_t sC, float* db, void* stream);

/* ---- per-sample normalisation: nn.InstanceNorm3d(1, eps=1e-3, affine=True) on x.unsqueeze(1)
 *      (models/networks.py:159,166-169) = LayerNorm over (C,H,W) with scalar gamma/beta.
 * stats: per-sample sum / sum of squares (double); the caller zero-initialises `sums` [N][PG_STAT_SLOTS][2] — partial
 *        sums are spread over PG_STAT_SLOTS slots per sample (no single hot atomic address), finalize adds them up.
 * finalize: mr[N][2] = (mean, rstd);  aff[N][2] = (gamma*rstd, beta - gamma*mean*rstd).            */
#define PG_STAT_SLOTS 16
int pg_norm_stats(const float* y, int32_t N, int64_t L, double* sums, void* stream);
int pg_norm_finalize(const double* sums, const float* gamma, const float* beta, int32_t N, int64_t L,
                     float eps, float* mr, float* aff, void* stream);
/* backward: bsums[N][2] += (sum dz, sum dz*zhat) (caller zero-initialises);
 * apply: dz <- dy = gamma*rstd*(dz - mean(dz) - zhat*mean(dz*zhat)) in place; dgamma/dbeta accumulated. */
int pg_norm_bwd_reduce(const float* dz, const float* y, const float* mr, int32_t N, int64_t L,
                       double* bsums, void* stream);
int pg_norm_bwd_apply(float* dz, const float* y, const float* mr, const double* bsums, const float* gamma,
                      int32_t N, int64_t L, float* dgamma, float* dbeta, void* stream);
/* the same; `dy_bf16` (NULL = none): also write dy as a bf16 tensor of the same shape — the operand copy the data- and
 * weight-gradient contractions of the bf16 data path read (saves their pg_materialise_bf16 pass over dy). */
int pg_norm_bwd_apply_ex(float* dz, const float* y, const float* mr, const double* bsums, const float* gamma,
                         int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, void* stream);

/* ---- key-point heat-maps (utils/pose_utils.py:79-86 cords_to_map; SURVEY.md §8f row 1: the step before the path).
 * cords [N][P][2] = (y, x) as float, -1 = missing (zero map); out[n*oN + c*oC + y*oH + x*oW] =
 * float32(exp(-((y-cy)^2 + (x-cx)^2) / (2 sigma^2))) evaluated in float64 like numpy does.  With the strides of a
 * channel slice of the NCHW network input the maps are written in place (no HWC array, no transpose, no H2D copy). */
int pg_cords_to_map(const float* cords, int32_t N, int32_t P, int32_t H, int32_t W, float sigma, float* out,
                    int64_t oN, int64_t oC, int64_t oH, int64_t oW, void* stream);

/* ---- per-sample pose geometry, the CPU work right before the step (utils/pose_transform.py:94-289,
 * executed by the reference inside Dataset.__getitem__ on the main thread: datasets/PoseTransfer_Dataset.py:89-108).
 * kp_* [N][P][2] = (y, x) as float, -1 = missing; P = 16 (LABELS) or 18 (LABELS_PAF), pose_utils.py:25-35.
 * pg_affine_transforms: affine_transforms(kp_from, kp_to) (pose_transform.py:213-289) -> out [N][10][8] float32
 *   = [a0 a1 a2 b0 b1 b2 0 0] (inverse map target -> source in pixels; "no point" rows [1 0 1000 0 1 1000 0 0]).
 *   Fit = skimage AffineTransform.estimate restated (Hartley-normalised total least squares), fp64; PARITY UNPINNED
 *   against scikit-image itself (absent offline), pinned against oracle/pose_geometry.py.
 * pg_pose_masks: pose_masks(kp_to, (H,W)) (pose_transform.py:143-184) -> out [N][10][H][W] float32 in {0,1}
 *   (polygon fill = the pnpoly crossing test of skimage.measure.grid_points_in_poly restated; same caveat).
 * A sample without the four torso joints makes the reference raise KeyError (compute_st_distance, :119-122); here it
 * yields "no point" transforms / a body-only mask set, and the host wrapper raises.                                   */
int pg_affine_transforms(const float* kp_from, const float* kp_to, int32_t N, int32_t P, float* out, void* stream);
int pg_pose_masks(const float* kp_to, int32_t N, int32_t P, int32_t H, int32_t W, float* out, void* stream);
/* estimate_uniform_transform (pose_transform.py:293-326; warp_skip='full'): one torso(+knees) fit -> out [N][1][8]. */
int pg_uniform_transform(const float* kp_from, const float* kp_to, int32_t N, int32_t P, float* out, void* stream);
/* _preprocess_image (utils/pose_utils.py:216-217) of a uint8 HWC batch [N][H][W][3] (DEVICE pointer, e.g. the async copy
 * of a pinned decode buffer): out[n*oN + c*oC + y*oH + x*oW] = float32((v / 255 - 0.5) * 2) evaluated in float64 —
 * with the strides of input[:, :3] the image lands in the network input without transpose / cat (Dataset.py:135-144,183). */
int pg_preprocess_image(const uint8_t* img, int32_t N, int32_t H, int32_t W, float* out, int64_t oN, int64_t oC,
                        int64_t oH, int64_t oW, void* stream);

/* ---- deformable skip connection (utils/pose_transform.py:16-92)
 * mask pyramid: cv2.resize(mask_HWT,(w,h)) INTER_LINEAR (pose_transform.py:84-87) on device.
 *   masks [N][T][H0][W0] (float32, or float64 when is_f64) -> out [N][h][w][T] float32.            */
int pg_mask_pyramid(const void* masks, int32_t is_f64, int32_t N, int32_t T, int32_t H0, int32_t W0,
                    int32_t h, int32_t w, float* out, void* stream);
/* fused affine_grid + grid_sample(bilinear, zeros) + mask multiply + max over T
 * (pose_transform.py:20-46,69-92).  feat raw NHWC [N][h][w][C] with optional deferred-norm `aff`;
 * warps [N][T][8]; lvl_masks [N][h][w][T]; out NHWC; argmax [N][h][w][C] uint8 (255 = a masked-out 0 won). */
int pg_warp_mask_max_fwd(const float* feat, const float* aff, const float* warps, const float* lvl_masks,
                         int32_t N, int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0,
                         int32_t align_corners, float* out, uint8_t* argmax, void* stream);
/* gradient wrt the (post-affine) features: dfeat = sum over the output pixels whose selected transform t* samples it of
 * gout * mask[t*] * bilinear weight.  dfeat is OVERWRITTEN (gather form: deterministic, no atomics; transforms that shrink
 * by more than ~0.6 fall back to a float-atomic scatter on top). */
int pg_warp_mask_max_bwd(const float* gout, const uint8_t* argmax, const float* warps, const float* lvl_masks,
                         int32_t N, int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0,
                         int32_t align_corners, float* dfeat, void* stream);

/* ---- losses (models/pose_gan.py)
 * GAN log loss on D logits [rows][K] (sigmoid fused; pose_gan.py:90-98,140-160):
 *   mode 0: -log(sigmoid(x)+1e-7) (real / generator), mode 1: -log(1-sigmoid(x)+1e-7) (fake);
 *   *loss += scale * sum; dlogits = scale * d/dx (NULL to skip); sig = sigmoid(x) (NULL to skip).   */
int pg_gan_logloss(const float* logits, int64_t count, int32_t mode, float scale, float* loss,
                   float* dlogits, float* sig, void* stream);
/* nn.L1Loss (pose_gan.py:66,105): *loss += scale*sum|p-t|; gout (+)= scale*sign(p-t).              */
int pg_l1_loss(const float* pred, const float* target, int64_t count, float scale, float* loss,
               float* gout, int32_t accumulate, void* stream);
/* d(pre-tanh) = g * (1 - out^2), in place on g (autograd of nn.Tanh, models/networks.py:232).      */
int pg_tanh_bwd(float* g, const float* out, int64_t count, void* stream);
/* Feature_Extractor(vgg19,'block1_conv2') = ReLU(conv1_1(prep(x))) with the view-not-permute
 * pre-process (utils/pose_utils.py:312-338).  x NCHW [N][3][H][W]; w [64][3][3][3]; feat NHWC [N][H][W][64]. */
int pg_vgg_conv1_relu_fwd(const float* x, const float* w, const float* b, int32_t N, int32_t H, int32_t W,
                          float* feat, void* stream);
/* gout[N][3][H][W] += d/dx of the above given dfeat (already masked by feat>0). */
int pg_vgg_conv1_dgrad(const float* dfeat, const float* w, int32_t N, int32_t H, int32_t W, float* gout,
                       void* stream);
/* nearest-neighbour L1 loss (pose_gan.py:173-199) on NHWC features [N][H][W][C], C%4==0, C<=256:
 *   *loss += scale * sum_pixels min_{|dy|,|dx|<=area/2} sum_c |G[p+d]-P[p]|;
 *   dP = scale * sign(P-G*) * (P>0 if relu_mask) at the arg-min offset.                             */
int pg_nn_loss(const float* P, const float* G, int32_t N, int32_t H, int32_t W, int32_t C, int32_t area,
               float scale, int32_t relu_mask, float* loss, float* dP, void* stream);

/* ---- optimiser / misc
 * torch.optim.Adam(betas=(b1,b2), eps) single fused step over a flat arena (pose_gan.py:50-51,111,167);
 * step_size = lr/(1-b1^t), bc2_sqrt = sqrt(1-b2^t) computed by the host in double.                  */
int pg_adam(float* p, const float* g, float* m, float* v, int64_t n, float b1, float b2, float eps,
            float step_size, float bc2_sqrt, float grad_scale, void* stream);
/* the same step with (a) bf16 gradients `g_bf16` (NULL = read fp32 `g`): the all-reduced bf16 bucket of the data-parallel
 * path, and (b) an optional bf16 copy `p_bf16` (NULL = none) of the UPDATED parameters written in the same pass — the
 * K-contiguous weight operand of the bf16 data path (the arena's packed [tap][Cout][Cin] layout).  n % 4 == 0.          */
int pg_adam_ex(float* p, const float* g, const void* g_bf16, float* m, float* v, int64_t n, float b1, float b2, float eps,
               float step_size, float bc2_sqrt, float grad_scale, void* p_bf16, void* stream);

/* ---- replay-safe scalars (HIP-graph capture of the iteration, runtime/graph.py): a captured graph freezes kernel
 * arguments, so the two per-iteration scalars of the step — the dropout key and Adam's step number — come from a device
 * counter that the graph itself increments once per replay.
 *   pg_counter_add       : *ctr += inc (one thread)
 *   pg_dropout_mask_ctr  : pg_dropout_mask with key' = mix64(key + *ctr * golden-ratio constant)
 *   pg_adam_ctr          : pg_adam_ex with step = step0 + *ctr; bias corrections 1 - b^step evaluated in double on the device
 *                          (torch.optim.Adam semantics, reference models/pose_gan.py:50-51) */
/* out[i] = a[i] + b[i], n elements (out may alias an input): loss totals (pose_gan.py:109,160), dW += product buffer */
int pg_add2(float* out, const float* a, const float* b, int64_t n, void* stream);
int pg_counter_add(uint64_t* ctr, uint64_t inc, void* stream);
int pg_dropout_mask_ctr(float* out, int64_t n, uint64_t key, float p, const uint64_t* ctr, void* stream);
int pg_adam_ctr(float* p, const float* g, const void* g_bf16, float* m, float* v, int64_t n, double b1, double b2, float eps,
                float lr, int64_t step0, const uint64_t* ctr, float grad_scale, void* p_bf16, void* stream);

/* Weight gradient of a k4/s2/p1 Block convolution on the bf16 data path (autograd of nn.Conv2d / nn.ConvTranspose2d+crop
 * weights, models/networks.py:154-157), straight from pixel-major bf16 operands with transposing LDS reads
 * (csrc/wgrad_bf16.hip): dW[16 taps][Cout][ldw] fp32, columns [col_off, col_off + Cx) (+)= sum over pixels dY * X for ONE
 * K-source x (already normalised / activated / masked: pg_materialise_bf16) of the virtual concat.
 * x_is_large = 1: Conv2d (x on the 2Hs x 2Ws grid, dy on Hs x Ws); 0: ConvTranspose2d + crop (x small, dy large).
 * Cx % 128 == 0, Cout % 128 == 0; ksplit <= 0: chosen by the library (float atomics when > 1).                          */
int pg_wgrad_bf16(const void* x_bf16, int32_t Cx, const void* dy_bf16, int32_t Cout, int32_t x_is_large, int32_t N,
                  int32_t Hs, int32_t Ws, float* dW, int32_t ldw, int32_t col_off, int32_t ksplit, void* stream);
/* the same for a large grid that is not exactly twice the small one: Conv2d k4 s2 p1 on an odd map (the discriminator's
 * 127 -> 63 -> 31 ... blocks, reference models/networks.py:341-347): Hs = (Hl - 2) / 2 + 1 */
int pg_wgrad_bf16_ex(const void* x_bf16, int32_t Cx, const void* dy_bf16, int32_t Cout, int32_t x_is_large, int32_t N,
                     int32_t Hs, int32_t Ws, int32_t Hl, int32_t Wl, float* dW, int32_t ldw, int32_t col_off,
                     int32_t ksplit, void* stream);

/* ---- data-parallel gradient exchange (NEW: the reference is single-process; main.py:44-159 has no
 * counterpart).  One process per GPU; RCCL (xGMI) is resolved at run time (dlopen librccl.so.1), the communication stream
 * and every ordering event belong to the caller.
 *   pg_comm_unique_id : rank 0 draws the 128-byte rendezvous token (ncclGetUniqueId); the host broadcasts it out of band
 *   pg_comm_init      : every rank joins (ncclCommInitRank) -> opaque handle
 *   pg_comm_allreduce_bucket : in-place SUM over ranks of `count` elements, dtype 0 = fp32 / 1 = bf16, enqueued on `stream`
 *   pg_pack_bf16      : fp32 -> bf16 (RNE) staging of a finished gradient range before a bf16 bucket is reduced
 *   pg_comm_ranks     : what RCCL itself reports for the communicator (ncclCommUserRank / ncclCommCount) — bench.py prints
 *                       it as `rccl_ranks` so that a scaling line proves how many ranks the collective really spanned      */
int pg_comm_unique_id(void* out128);
int pg_comm_init(const void* unique_id128, int32_t rank, int32_t world, void** comm);
int pg_comm_allreduce_bucket(void* comm, void* buf, int64_t count, int32_t dtype, void* stream);
int pg_comm_destroy(void* comm);
int pg_comm_ranks(void* comm, int32_t* rank, int32_t* world);
int pg_pack_bf16(const float* src, void* dst, int64_t n, void* stream);

/* channel-dropout multipliers in {0, 1/(1-p)} from a stateless counter hash (nn.Dropout2d, networks.py:161). */
int pg_dropout_mask(float* out, int64_t n, uint64_t key, float p, void* stream);
int pg_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int pg_nhwc_to_nchw(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
/* y = act((a_n*x+b_n)*mask): materialise a deferred-norm tensor (module-level API / tests only).   */
int pg_apply_affine_act(const float* x, const float* aff, const float* mask, int32_t act, int32_t N,
                        int64_t HW, int32_t C, float* y, void* stream);

const char* pg_last_error(void);
int pg_version(void);
/* (round 5) PG_DETERMINISTIC: run-to-run bit-repeatable results at the price of speed.  Every launch path whose result depends on
 * an arrival order is replaced by an ordered one: split-K contractions only through the workspace + fix-up pass (summed in split
 * order) or un-split; weight gradients un-split (no float atomics on dW); bias / first-layer / output-convolution partial
 * reductions by one serial walk per element; loss sums (pg_l1_loss, pg_gan_logloss) by one workgroup; the warp backward's
 * per-pixel candidate lists sorted before they are summed.  What keeps an arrival order: DOUBLE atomics on the per-sample
 * statistics (fp32-invisible: the order moves a 53-bit sum of fp32-accurate partials in its last bits), the value (not the
 * gradient) of pg_nn_loss, and the float-atomic scatter of warp transforms that shrink by more than ~0.6 (none in the synthetic
 * or data-set transforms).  The environment variable PG_DETERMINISTIC=1 sets the initial value; pg_set_deterministic flips it at
 * run time (host-side flag, read per launch).  The reference has no counterpart (torch.use_deterministic_algorithms is never
 * called in src_deformable). */
int pg_set_deterministic(int32_t on);
int pg_get_deterministic(void);
/* diagnostics: tile config | loader modes << 4/8 | split-K << 16 of this thread's last pg_conv / pg_conv_wgrad */
int pg_last_launch_info(void);
/* timing helper for bench.py: HIP events on the caller's stream (torch.cuda.Event sees only torch's). */
int pg_event_create(void** ev);
int pg_event_record(void* ev, void* stream);
int pg_event_elapsed_ms(void* start, void* stop, float* ms);  /* synchronises on `stop` */
int pg_event_destroy(void* ev);

/* Test aid (no reference counterpart): one lane busy-waits ~`microseconds` (0..50000) on `stream`, delaying whatever is
 * enqueued behind it — used by the stream-ordering stress test of the data-parallel reducer (runtime/dp.py). */
int pg_debug_spin(int32_t microseconds, void* stream);
/* Measurement aid (no reference counterpart): with PG_DEBUG_CONV_TIMELINE set, every workgroup of the 256-row bf16 convolution
 * kernel stamps its phases (0 start, 1 row table built, 2 first K tile landed, 3 K loop done, 4 epilogue done: shader clock;
 * 5 / 6 the 100 MHz wall clock at start / end; 7 the XCD); this copies the stamps of the LAST launch: n_wgs x 8 uint64. */
int pg_debug_conv_timeline(unsigned long long* host_out, int32_t n_wgs);
/* Test aid: number of 32-pixel tiles the LAST gather-form warp backward handed to its full-capacity second launch (a pixel's
 * 48-entry candidate list overflowed); synchronises the device. */
int pg_debug_warp_gather_overflows(int32_t* count);

/* ---------------------------------------------------------------------------------------------------------------------
 * Round 3 — bf16 STORAGE on the bf16 data path (PG_PREC_BF16_DATA).  Raw convolution outputs (the tensors the reference's
 * Block keeps between conv and norm, models/networks.py:154-169) and the gradients flowing back through them are kept as
 * bf16 NHWC tensors instead of fp32: every HBM-bound pass around the contractions (norm backward, warp, materialisation,
 * first / last layers, the epilogues) moves half the bytes.  Statistics, accumulation, master weights, Adam, losses stay
 * fp32 / double.  pg_conv_t.out_bf16 and pg_dst_t.flags select it per launch for the contractions; the entry points
 * below are the `_io` forms of the streaming kernels (io_flags bits documented per function; 0 = the fp32 form).
 * Reference lines replaced are those of the fp32 forms they extend. */
/* pg_materialise_bf16 with a bf16 raw input (x_is_bf16) and an optional SECOND activated copy (out2_bf16, act2) written in
 * the same pass: an encoder skip is read through LeakyReLU by the next level (networks.py:150) and through ReLU by the
 * decoder (networks.py:152). */
int pg_materialise_bf16_ex(const void* x, int32_t x_is_bf16, const float* aff, const float* mask, int32_t act, int32_t N,
                           int64_t HW, int32_t C, void* out_bf16, void* out2_bf16, int32_t act2, void* stream);
/* (round 4) pg_materialise_bf16_ex with pg_norm_finalize folded in (InstanceNorm3d(1) statistics -> per-sample affine,
 * networks.py:159,166-169): `sums` [N][PG_STAT_SLOTS][2] are the complete statistics the producing pg_conv accumulated; the kernel
 * evaluates the affine per workgroup and publishes (mean, rstd) to `mr` and (a, b) to `aff` for the readers in later launches. */
int pg_materialise_bf16_norm(const void* x, int32_t x_is_bf16, const double* sums, const float* gamma, const float* beta,
                             int64_t L, float eps, float* mr, float* aff, const float* mask, int32_t act, int32_t N, int64_t HW,
                             int32_t C, void* out_bf16, void* out2_bf16, int32_t act2, void* stream);
/* (round 5) The generator's output convolution, forward, in ONE streaming pass on the bf16 data path (csrc/out_conv_fwd.hip;
 * replaces networks.py:228 `ReLU -> Conv2d(cin, 3, k3, p1) -> Tanh` over the concat of the last decoder block's normalised output
 * and the level-0 skips, networks.py:236-250): x0 = the block's RAW bf16 output [N][H][W][C0], normalised here with the published
 * per-sample affine `aff` or — pg_norm_finalize folded in as in pg_materialise_bf16_norm — from the complete statistics `sums`
 * (then gamma / beta / L / eps / mr / aff_out are required and (mean, rstd), (a, b) are published); op0 receives
 * bf16(relu(a x0 + b)), the activated operand the backward pass reads; x1 / x2 = further sources as bf16 ACTIVATED operands
 * (C = 0: absent); W27 = the fp32 weight viewed as [27 = tap * 3 + co][C0 + C1 + C2]; out = act(bias + conv) as NCHW fp32.
 * (C0, C1, C2) in {(128, 64, 64), (128, 64, 0), (64, 64, 0)}. */
int pg_out_conv_fwd_fused(const void* x0_bf16, int32_t C0, const float* aff, const double* sums, const float* gamma,
                          const float* beta, int64_t L, float eps, float* mr, float* aff_out, void* op0_bf16, const void* x1_bf16,
                          int32_t C1, const void* x2_bf16, int32_t C2, const float* W27, const float* bias, int32_t N, int32_t H,
                          int32_t W, int32_t out_act, float* out, void* stream);
/* io_flags: bit 0 = dz is bf16, bit 1 = y is bf16 */
int pg_norm_bwd_reduce_ex(const void* dz, const void* y, const float* mr, int32_t N, int64_t L, double* bsums,
                          int32_t io_flags, void* stream);
int pg_norm_bwd_apply_io(void* dz, const void* y, const float* mr, const double* bsums, const float* gamma, int32_t N,
                         int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, int32_t io_flags, void* stream);
/* (round 4) the same with the per-sample sums taken from a producer's epilogue (pg_dst_t.bsums, [N][PG_STAT_SLOTS][2]):
 * sums_mode 1: (sum r, sum r * y_raw)  -> sum r * xhat = rstd * (S2 - mean * S1)        (data-gradient scatter: `fwd` = raw tensor)
 * sums_mode 2: (sum r, sum r * x_act)  -> sum r * xhat = (S2 - beta * S1) / gamma       (pg_out_conv_bwd_direct: `fwd` = relu(norm(y)),
 *              r = 0 wherever the ReLU is off; gamma == 0 makes dz = 0 and the gamma gradient 0 — the normalised value is not
 *              recoverable from a constant)
 * sums_mode 0 = pg_norm_bwd_apply_io (bsums [N][2] from pg_norm_bwd_reduce). */
int pg_norm_bwd_apply_v2(void* dz, const void* y, const float* mr, const double* sums, const float* gamma, const float* beta,
                         int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, int32_t io_flags,
                         int32_t sums_mode, void* stream);
/* (round 5) sums_mode 2 divides by gamma: ill-conditioned when |gamma| < 1e-3 + 0.02 |beta| (undefined at gamma == 0, where the
 * gamma gradient could never leave zero).  pg_norm_bwd_reduce_guard is pg_norm_bwd_reduce_ex that runs ONLY then — the predicate
 * is evaluated on the device from (gamma, beta), no host synchronisation; every workgroup of a well-conditioned layer exits at
 * once — into `bsums_guard` [N][2] (caller-zeroed); pg_norm_bwd_apply_v3 is pg_norm_bwd_apply_v2 that reads `bsums_guard` under
 * the same predicate (sums_mode 2 only; NULL = the v2 behaviour).  Reference: autograd of nn.InstanceNorm3d, networks.py:159. */
int pg_norm_bwd_reduce_guard(const void* dz, const void* y, const float* mr, const float* gamma, const float* beta, int32_t N,
                             int64_t L, double* bsums_guard, int32_t io_flags, void* stream);
int pg_norm_bwd_apply_v3(void* dz, const void* y, const float* mr, const double* sums, const float* gamma, const float* beta,
                         int32_t N, int64_t L, float* dgamma, float* dbeta, uint16_t* dy_bf16, int32_t io_flags,
                         int32_t sums_mode, const double* bsums_guard, void* stream);
/* io_flags: bit 0 = feat is bf16, bit 1 = out is bf16, bit 2 = store relu(out) (the decoder reads the warped skip only
 * through its ReLU; relu(x) > 0 <=> x > 0 keeps the backward's activation derivative) */
int pg_warp_mask_max_fwd_io(const void* feat, const float* aff, const float* warps, const float* lvl_masks, int32_t N,
                            int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0, int32_t align_corners,
                            void* out, uint8_t* argmax, int32_t io_flags, void* stream);
/* io_flags: bit 0 = gout is bf16, bit 1 = dfeat is bf16 */
int pg_warp_mask_max_bwd_io(const void* gout, const uint8_t* argmax, const float* warps, const float* lvl_masks, int32_t N,
                            int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0, int32_t align_corners,
                            void* dfeat, int32_t io_flags, void* stream);
/* bounding boxes (ymin, ymax, xmin, xmax; empty: ymax < ymin) of the non-zero pixels of the (N, T, H0, W0) limb masks, and the
 * warp backward that uses them to skip (pixel, transform) pairs whose candidates all carry a zero mask
 * (utils/pose_transform.py:84-89: the masks are zero outside the limb polygons) */
int pg_mask_bbox(const void* masks, int32_t is_f64, int32_t N, int32_t T, int32_t H0, int32_t W0, int32_t* bbox, void* stream);
int pg_warp_mask_max_bwd_bbox(const void* gout, const uint8_t* argmax, const float* warps, const float* lvl_masks,
                              const int32_t* bbox, int32_t N, int32_t T, int32_t C, int32_t h, int32_t w, int32_t H0, int32_t W0,
                              int32_t align_corners, void* dfeat, int32_t io_flags, void* stream);
/* first layers: `out` (fp32) may be NULL; up to three bf16 outputs bf16(act_k(conv + bias)), PG_ACT_NONE = the raw tensor */
int pg_stem_conv_bf16_v3(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                         int32_t pad, const uint16_t* Wp, const float* bias, float* out, uint16_t* out_bf16, int32_t act,
                         uint16_t* out2_bf16, int32_t act2, uint16_t* out3_bf16, int32_t act3, void* stream);
int pg_stem_wgrad_bf16_ex(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                          int32_t pad, const void* dY, int32_t dy_is_bf16, float* dW, float* workspace,
                          int64_t workspace_floats, void* stream);
/* round 4: as _ex, and optionally the layer's BIAS gradient from the same pass — db[co] += sum over pixels of dY, through a constant-one
 * input channel in a spare slot of the kernel's channel padding (reference: autograd of the conv bias, models/networks.py:186,341).
 * Done when pg_last_launch_info() has PG_INFO_STEM_BIAS set (Cin below the padded channel count; k3 p1 or p0); otherwise the
 * caller runs pg_bias_grad / pg_bias_grad_bf16 as before. */
#define PG_INFO_STEM_BIAS (1 << 15)
int pg_stem_wgrad_bf16_v2(const pg_src_t* src, int32_t nsrc, int32_t N, int32_t Hi, int32_t Wi, int32_t K, int32_t stride,
                          int32_t pad, const void* dY, int32_t dy_is_bf16, float* dW, float* dbias, float* workspace,
                          int64_t workspace_floats, void* stream);
/* db[c] += sum over pixels of a dense NHWC bf16 gradient (bias gradient of the first layers, networks.py:186,341) */
int pg_bias_grad_bf16(const void* dY_bf16, int64_t npix, int32_t C, float* db, void* stream);
/* pg_tap_gather (k3 p1, 3 outputs) over a tap tensor whose pixel rows are `pitch` >= 27 floats apart */
int pg_tap_gather_pitch(const float* Y, int32_t pitch, int32_t N, int32_t H, int32_t W, const float* bias, int32_t out_act,
                        float* out, int64_t oN, int64_t oC, int64_t oH, int64_t oW, void* stream);
/* pg_out_conv_dgrad that also accumulates the weight gradient dW[27][Ctot] of the output convolution (networks.py:228) in
 * the same pass: every destination's `fwd` is the ACTIVATED operand of the forward pass (no aff / mask); Ctot = 256;
 * workspace >= 64 * Ctot * 28 floats. */
int pg_out_conv_dgrad_wgrad(const float* G, const float* Wt, int32_t N, int32_t H, int32_t W, const pg_dst_t* dst,
                            int32_t ndst, float* dW, float* workspace, int64_t workspace_floats, void* stream);
/* the same; g_is_dpre = 1: G is the NCHW (N,3,H,W) gradient wrt the pre-tanh output itself (the kernels gather the 27 (tap,
 * channel) values of a pixel: no pg_im2col_taps pass); wg_stream: stream of the weight-gradient pass (NULL = stream) */
int pg_out_conv_bwd_direct(const float* G, int32_t g_is_dpre, const float* Wt, int32_t N, int32_t H, int32_t W,
                           const pg_dst_t* dst, int32_t ndst, float* dW, float* workspace, int64_t workspace_floats,
                           void* wg_stream, void* stream);
/* bf16 im2col of the output convolution's gradient (pg_im2col_taps with a bf16 G, Cpad = 64) and its weight gradient alone:
 * dW[27][Ctot] += sum_pixels G[pixel][t] * x[pixel][ci], x = the activated bf16 forward operands given as dst[].fwd.  The data
 * gradient of that layer is then a plain bf16 pg_conv (K = 64) with the zero-padded weight. */
int pg_im2col_taps_bf16(const float* dY, int64_t yN, int64_t yC, int64_t yH, int64_t yW, int32_t N, int32_t H, int32_t W,
                        int32_t KH, int32_t KW, int32_t pad, int32_t C, int32_t Cpad, void* G_bf16, void* stream);
int pg_out_conv_wgrad_bf16(const void* G_bf16, int32_t g_pitch, int32_t N, int32_t H, int32_t W, const pg_dst_t* dst,
                           int32_t ndst, float* dW, float* workspace, int64_t workspace_floats, void* stream);

/* every per-tap TRANSPOSED bf16 weight copy of a parameter arena (the data-gradient operands of the bf16 data path) in one
 * launch: table = device array of {int64 arena offset (floats); int32 taps, Cout, Cin, first tile} records, 32 x 32 tiles */
int pg_weights_to_bf16_batch(const float* arena, const void* table, int32_t ntab, int32_t total_tiles, void* out_t_bf16,
                             void* stream);

/* ---- launch tape (round 3; no reference counterpart: the reference's loop is eager PyTorch, main.py:77-108).  Between
 * pg_tape_begin and pg_tape_end the calling thread's enqueues (kernel launches, memsets, pg_stream_wait, pg_zero, pg_copy) are
 * issued as usual AND recorded with a copy of their arguments; pg_tape_replay re-issues them on the same streams — one call per
 * training iteration.  Requirements as for a HIP graph: persistent buffers, per-iteration scalars in device memory
 * (pg_dropout_mask_ctr, pg_adam_ctr, pg_counter_add), inputs copied into static tensors before the replay. */
int pg_tape_begin(void);
int pg_tape_end(void** tape, int64_t* n_ops);
int pg_tape_replay(void* tape);
int pg_tape_destroy(void* tape);
/* stream `waiter` waits for the work enqueued on `waited` so far (event record + wait; recorded on a tape) */
int pg_stream_wait(void* waiter, void* waited);
int pg_zero(void* ptr, int64_t bytes, void* stream);
int pg_copy(void* dst, const void* src, int64_t bytes, void* stream);
/* dst[c][r] = src[r][c], R x C row-major fp32, destination rows `ld` floats apart */
int pg_transpose_f32(const float* src, int32_t R, int32_t C, float* dst, int32_t ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POSEGAN_HIP_H */
