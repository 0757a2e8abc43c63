// Key-point geometry on the device: the per-sample CPU work the reference does on the main thread right before the
// training step — reference utils/pose_transform.py:94-289.
//
//  * pg_affine_transforms : affine_transforms() (pose_transform.py:213-289): ten inverse (target -> source) affine maps
//                           per sample, each a least-squares fit between corresponding key-point sets / limb polygons.
//  * pg_pose_masks        : pose_masks() (pose_transform.py:143-184): ten {0,1} masks per sample (whole image, head box,
//                           eight limb quadrilaterals rasterised with an even-odd point-in-polygon test).
//  * pg_preprocess_image  : _preprocess_image() (pose_utils.py:216-217) on a uint8 HWC image batch, written NCHW.
//
// The two third-party primitives the reference calls are restated from their published algorithms (scikit-image is not in
// this image: PARITY UNPINNED for them, oracle/pose_geometry.py is the NumPy restatement the tests compare against):
//  * skimage.transform.estimate_transform('affine', src, dst)  = AffineTransform.estimate: Hartley normalisation of both
//    point sets (centroid to 0, RMS distance to sqrt 2), total least squares on the 2n x 7 system [xs ys 1 0 0 0 xd; 0 0 0
//    xs ys 1 yd] (right singular vector of the smallest singular value, params = -v[0..5] / v[6]), de-normalisation
//    H = inv(T_dst) * Hn * T_src.  Here: smallest eigenvector of the 7x7 Gram matrix by cyclic Jacobi in fp64.
//  * skimage.measure.grid_points_in_poly(shape, verts)  = W. R. Franklin's pnpoly crossing test evaluated at every
//    integer (row, col), x = row, y = col (the release contemporary with the reference, <= 0.15).
// Everything is tiny: one thread per (sample, transform) for the fits, one thread per pixel for the masks.
#include "common.h"

namespace pg {

// name slots used by the geometry (index into the key-point array, or -1 when the skeleton lacks the joint):
// LABELS_PAF / LABELS of reference utils/pose_utils.py:25-35
enum { J_RHIP, J_LHIP, J_RSHO, J_LSHO, J_RKNE, J_LKNE, J_RANK, J_LANK, J_RELB, J_LELB, J_RWRI, J_LWRI,
       J_NOSE, J_LEYE, J_REYE, J_LEAR, J_REAR, J_COUNT };

struct JointTab { int idx[J_COUNT]; };

static JointTab joint_table(int pose_dim) {
  JointTab t;
  for (int i = 0; i < J_COUNT; ++i) t.idx[i] = -1;
  if (pose_dim == 16) {
    // 'Rank','Rknee','Rhip','Lhip','Lknee','Lank','pelv','spine','neck','head','Rwri','Relb','Rsho','Lsho','Lelb','Lwri'
    // (the geometry asks for 'Rkne' / 'Lkne' / 'nose' / eyes / ears, which this skeleton does not have)
    t.idx[J_RANK] = 0; t.idx[J_RHIP] = 2; t.idx[J_LHIP] = 3; t.idx[J_LANK] = 5; t.idx[J_RWRI] = 10; t.idx[J_RELB] = 11;
    t.idx[J_RSHO] = 12; t.idx[J_LSHO] = 13; t.idx[J_LELB] = 14; t.idx[J_LWRI] = 15;
  } else {
    // 'nose','neck','Rsho','Relb','Rwri','Lsho','Lelb','Lwri','Rhip','Rkne','Rank','Lhip','Lkne','Lank','Leye','Reye','Lear','Rear'
    t.idx[J_NOSE] = 0; t.idx[J_RSHO] = 2; t.idx[J_RELB] = 3; t.idx[J_RWRI] = 4; t.idx[J_LSHO] = 5; t.idx[J_LELB] = 6;
    t.idx[J_LWRI] = 7; t.idx[J_RHIP] = 8; t.idx[J_RKNE] = 9; t.idx[J_RANK] = 10; t.idx[J_LHIP] = 11; t.idx[J_LKNE] = 12;
    t.idx[J_LANK] = 13; t.idx[J_LEYE] = 14; t.idx[J_REYE] = 15; t.idx[J_LEAR] = 16; t.idx[J_REAR] = 17;
  }
  return t;
}

struct P2 { double x, y; };

// give_name_to_keypoints (pose_transform.py:94-104): a joint is present when neither coordinate is MISSING_VALUE (-1);
// the value is array[i][::-1] = (x, y).
__device__ __forceinline__ bool joint(const float* kp, int P, const JointTab& jt, int j, P2& out) {
  const int i = jt.idx[j];
  if (i < 0 || i >= P) return false;
  const float y = kp[2 * i], x = kp[2 * i + 1];
  if (y == -1.f || x == -1.f) return false;
  out.x = (double)x; out.y = (double)y;
  return true;
}

// compute_st_distance (pose_transform.py:119-122)
__device__ __forceinline__ bool st_distance(const float* kp, int P, const JointTab& jt, double& st) {
  P2 rh, lh, rs, ls;
  if (!(joint(kp, P, jt, J_RHIP, rh) & joint(kp, P, jt, J_LHIP, lh) & joint(kp, P, jt, J_RSHO, rs) & joint(kp, P, jt, J_LSHO, ls)))
    return false;
  const double d1 = (rh.x - rs.x) * (rh.x - rs.x) + (rh.y - rs.y) * (rh.y - rs.y);
  const double d2 = (lh.x - ls.x) * (lh.x - ls.x) + (lh.y - ls.y) * (lh.y - ls.y);
  st = sqrt((d1 + d2) / 2.0);
  return true;
}

// estimate_polygon (pose_transform.py:187-209); note `to` is extended from the ALREADY extended `fr`.
__device__ __forceinline__ void limb_polygon(P2 fr, P2 to, double st, double inc_to, P2 (&v)[4]) {
  const double inc_from = 0.1, p_to = 0.2, p_from = 0.2;
  fr.x = fr.x + (fr.x - to.x) * inc_from; fr.y = fr.y + (fr.y - to.y) * inc_from;
  to.x = to.x + (to.x - fr.x) * inc_to;   to.y = to.y + (to.y - fr.y) * inc_to;
  double nx = -(fr.y - to.y), ny = fr.x - to.x;
  const double norm = sqrt(nx * nx + ny * ny);
  if (norm == 0.0) {
    v[0] = {fr.x + 1, fr.y + 1}; v[1] = {fr.x - 1, fr.y - 1}; v[2] = {to.x - 1, to.y - 1}; v[3] = {to.x + 1, to.y + 1};
    return;
  }
  nx /= norm; ny /= norm;
  v[0] = {fr.x + st * p_from * nx, fr.y + st * p_from * ny};
  v[1] = {fr.x - st * p_from * nx, fr.y - st * p_from * ny};
  v[2] = {to.x - st * p_to * nx, to.y - st * p_to * ny};
  v[3] = {to.x + st * p_to * nx, to.y + st * p_to * ny};
}

// _center_and_normalize_points of skimage.transform._geometric
__device__ __forceinline__ bool hartley(const P2* p, int n, double& nf, double& cx, double& cy) {
  cx = 0; cy = 0;
  for (int i = 0; i < n; ++i) { cx += p[i].x; cy += p[i].y; }
  cx /= n; cy /= n;
  double ss = 0;
  for (int i = 0; i < n; ++i) ss += (p[i].x - cx) * (p[i].x - cx) + (p[i].y - cy) * (p[i].y - cy);
  const double rms = sqrt(ss / n);
  if (!(rms > 0.0)) return false;
  nf = sqrt(2.0) / rms;
  return true;
}

constexpr int MAXPTS = 8;

// AffineTransform.estimate(src, dst) -> H (row-major 3x3, last row 0 0 1).  false = degenerate input (no finite fit).
__device__ bool estimate_affine(const P2* src, const P2* dst, int n, double (&H)[9]) {
  double ns, sx, sy, nd, dx, dy;
  if (n < 1 || !hartley(src, n, ns, sx, sy) || !hartley(dst, n, nd, dx, dy)) return false;
  // Gram matrix G = A^T A of the 2n x 7 system, columns [xs ys 1 | xs ys 1 | rhs]
  double G[7][7];
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) G[i][j] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double xs = ns * (src[i].x - sx), ys = ns * (src[i].y - sy);
    const double xd = nd * (dst[i].x - dx), yd = nd * (dst[i].y - dy);
    const double r1[7] = {xs, ys, 1, 0, 0, 0, xd}, r2[7] = {0, 0, 0, xs, ys, 1, yd};
    for (int a = 0; a < 7; ++a) for (int b = 0; b < 7; ++b) G[a][b] += r1[a] * r1[b] + r2[a] * r2[b];
  }
  // cyclic Jacobi: G = V diag(e) V^T
  double V[7][7];
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 7; ++p) for (int q = p + 1; q < 7; ++q) off += G[p][q] * G[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 7; ++q) {
        const double apq = G[p][q];
        if (fabs(apq) < 1e-320) continue;
        const double theta = (G[q][q] - G[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 7; ++k) {
          const double gkp = G[k][p], gkq = G[k][q];
          G[k][p] = c * gkp - s * gkq; G[k][q] = s * gkp + c * gkq;
        }
        for (int k = 0; k < 7; ++k) {
          const double gpk = G[p][k], gqk = G[q][k];
          G[p][k] = c * gpk - s * gqk; G[q][k] = s * gpk + c * gqk;
        }
        for (int k = 0; k < 7; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int m = 0;
  for (int i = 1; i < 7; ++i) if (G[i][i] < G[m][m]) m = i;
  const double w = V[6][m];
  if (w == 0.0) return false;
  double h[6];
  for (int i = 0; i < 6; ++i) h[i] = -V[i][m] / w;
  // H = inv(T_dst) * Hn * T_src with T = [[nf,0,-nf*c.x],[0,nf,-nf*c.y],[0,0,1]]
  const double a = h[0] * ns, b = h[1] * ns, c0 = h[2] - h[0] * ns * sx - h[1] * ns * sy;
  const double d = h[3] * ns, e = h[4] * ns, f0 = h[5] - h[3] * ns * sx - h[4] * ns * sy;
  H[0] = a / nd; H[1] = b / nd; H[2] = c0 / nd + dx;
  H[3] = d / nd; H[4] = e / nd; H[5] = f0 / nd + dy;
  H[6] = 0; H[7] = 0; H[8] = 1;
  for (int i = 0; i < 6; ++i) if (!isfinite(H[i])) return false;
  return true;
}

__device__ __forceinline__ void no_point(float* o) {     // pose_transform.py:221
  o[0] = 1; o[1] = 0; o[2] = 1000; o[3] = 0; o[4] = 1; o[5] = 1000; o[6] = 0; o[7] = 0;
}

// to_transforms (pose_transform.py:224-230): a matrix numpy cannot invert becomes the "no point" transform
__device__ __forceinline__ void emit(const double (&H)[9], bool ok, float* o) {
  const double det = H[0] * H[4] - H[1] * H[3];
  if (!ok || det == 0.0 || !isfinite(det)) { no_point(o); return; }
  for (int i = 0; i < 8; ++i) o[i] = (float)H[i];
}

struct LimbDef { int fr, to, fr_m, to_m; double inc_tr, inc_mask; };
__device__ const LimbDef kLimbs[8] = {
    {J_RHIP, J_RKNE, J_LHIP, J_LKNE, 0.1, 0.1}, {J_LHIP, J_LKNE, J_RHIP, J_RKNE, 0.1, 0.1},
    {J_RKNE, J_RANK, J_LKNE, J_LANK, 0.3, 0.5}, {J_LKNE, J_LANK, J_RKNE, J_RANK, 0.3, 0.5},
    {J_RSHO, J_RELB, J_LSHO, J_LELB, 0.1, 0.1}, {J_LSHO, J_LELB, J_RSHO, J_RELB, 0.1, 0.1},
    {J_RELB, J_RWRI, J_LELB, J_LWRI, 0.3, 0.5}, {J_LELB, J_LWRI, J_RELB, J_RWRI, 0.3, 0.5}};

// one thread per (sample, transform): out [N][10][8] = [a0 a1 a2 b0 b1 b2 0 0]
__global__ __launch_bounds__(64) void affine_transforms_kernel(const float* kp1, const float* kp2, int N, int P, JointTab jt,
                                                               float* out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= N * 10) return;
  const int n = i / 10, t = i - n * 10;
  const float* k1 = kp1 + (long)n * P * 2;
  const float* k2 = kp2 + (long)n * P * 2;
  float* o = out + (long)i * 8;
  double st1 = 0, st2 = 0;
  const bool ok1 = st_distance(k1, P, jt, st1), ok2 = st_distance(k2, P, jt, st2);
  if (!(ok1 && ok2)) { no_point(o); return; }    // the reference raises KeyError here; see include/posegan_hip.h
  P2 a[MAXPTS], b[MAXPTS];
  double H[9];
  if (t == 0) {           // body: ['Rhip','Lhip','Lsho','Rsho'], src = pose 2, dst = pose 1 (pose_transform.py:232-235)
    const int names[4] = {J_RHIP, J_LHIP, J_LSHO, J_RSHO};
    for (int q = 0; q < 4; ++q) { joint(k1, P, jt, names[q], a[q]); joint(k2, P, jt, names[q], b[q]); }
    emit(H, estimate_affine(b, a, 4, H), o);
    return;
  }
  if (t == 1) {           // head (pose_transform.py:240-254)
    const int cand[5] = {J_LEYE, J_REYE, J_LEAR, J_REAR, J_NOSE};
    int cnt = 0;
    for (int q = 0; q < 5; ++q) {
      P2 u, v;
      if (joint(k1, P, jt, cand[q], u) & joint(k2, P, jt, cand[q], v)) { a[cnt] = u; b[cnt] = v; ++cnt; }
    }
    if (cnt == 0) { no_point(o); return; }
    joint(k1, P, jt, J_LSHO, a[cnt]); joint(k2, P, jt, J_LSHO, b[cnt]); ++cnt;
    joint(k1, P, jt, J_RSHO, a[cnt]); joint(k2, P, jt, J_RSHO, b[cnt]); ++cnt;
    emit(H, estimate_affine(b, a, cnt, H), o);
    return;
  }
  // limbs: estimate_join (pose_transform.py:256-275)
  const LimbDef L = kLimbs[t - 2];
  P2 f2, t2, f1, t1;
  if (!(joint(k2, P, jt, L.fr, f2) & joint(k2, P, jt, L.to, t2))) { no_point(o); return; }
  if (!(joint(k1, P, jt, L.fr, f1) & joint(k1, P, jt, L.to, t1)))
    if (!(joint(k1, P, jt, L.fr_m, f1) & joint(k1, P, jt, L.to_m, t1))) { no_point(o); return; }   // the mirrored limb
  P2 p2[4], p1[4];
  limb_polygon(f2, t2, st2, L.inc_tr, p2);
  limb_polygon(f1, t1, st1, L.inc_tr, p1);
  emit(H, estimate_affine(p2, p1, 4, H), o);
}

// estimate_uniform_transform (pose_transform.py:293-326, warp_skip='full'): ONE fit over the torso joints plus the knees
// present in both poses.  out [N][1][8] (the reference hands over 9 numbers when the fit is invertible, 8 otherwise; only
// the first six are ever read, pose_transform.py:28).
__global__ __launch_bounds__(64) void uniform_transform_kernel(const float* kp1, const float* kp2, int N, int P, JointTab jt,
                                                               float* out) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const float* k1 = kp1 + (long)n * P * 2;
  const float* k2 = kp2 + (long)n * P * 2;
  float* o = out + (long)n * 8;
  P2 a[MAXPTS], b[MAXPTS];
  const int names[6] = {J_RHIP, J_LHIP, J_LSHO, J_RSHO, J_RKNE, J_LKNE};
  int cnt = 0;
  bool torso = true;
  for (int q = 0; q < 6; ++q) {
    P2 u, v;
    const bool ok = joint(k1, P, jt, names[q], u) & joint(k2, P, jt, names[q], v);
    if (ok) { a[cnt] = u; b[cnt] = v; ++cnt; }
    else if (q < 4) torso = false;
  }
  double H[9];
  if (!torso) { no_point(o); return; }
  emit(H, estimate_affine(b, a, cnt, H), o);
}

struct MaskShape { int kind; int x0, x1, y0, y1; double vr[4], vc[4]; };   // kind: 0 empty, 1 full, 2 box, 3 polygon

// pnpoly at (row, col): x = row against vr, y = col against vc
__device__ __forceinline__ bool in_poly(const MaskShape& s, double x, double y) {
  bool c = false;
  int j = 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if ((((s.vc[i] <= y) && (y < s.vc[j])) || ((s.vc[j] <= y) && (y < s.vc[i]))) &&
        (x < (s.vr[j] - s.vr[i]) * (y - s.vc[i]) / (s.vc[j] - s.vc[i]) + s.vr[i]))
      c = !c;
    j = i;
  }
  return c;
}

// grid (pixel blocks, 10, N): out [N][10][H][W] float32 in {0, 1}
__global__ __launch_bounds__(256) void pose_masks_kernel(const float* kp2, int P, int H, int W, JointTab jt, float* out) {
  __shared__ MaskShape sh;
  const int n = blockIdx.z, t = blockIdx.y;
  if (threadIdx.x == 0) {
    const float* k2 = kp2 + (long)n * P * 2;
    MaskShape s;
    s.kind = 0;
    double st2 = 0;
    const bool ok = st_distance(k2, P, jt, st2);
    if (t == 0) s.kind = 1;                                  // body mask = ones (pose_transform.py:149)
    else if (ok && t == 1) {                                 // head box around the integer centre of mass (:153-165)
      const int cand[5] = {J_LEYE, J_REYE, J_LEAR, J_REAR, J_NOSE};
      int cnt = 0; double mx = 0, my = 0; P2 u;
      for (int q = 0; q < 5; ++q) if (joint(k2, P, jt, cand[q], u)) { mx += u.x; my += u.y; ++cnt; }
      if (cnt) {
        const int cx = (int)(mx / cnt), cy = (int)(my / cnt);     // astype(int): truncation
        const int b = (int)(0.40 * st2);                          // mask_from_kp_array: int(border_inc)
        s.kind = 2;
        s.x0 = max(cx - b, 0); s.y0 = max(cy - b, 0);
        s.x1 = min(cx + b, W); s.y1 = min(cy + b, H);
      }
    } else if (ok && t >= 2) {                               // limb quadrilateral (:167-182)
      const LimbDef L = kLimbs[t - 2];
      P2 f, to;
      if (joint(k2, P, jt, L.fr, f) & joint(k2, P, jt, L.to, to)) {
        P2 v[4];
        limb_polygon(f, to, st2, L.inc_mask, v);
        s.kind = 3;
        for (int q = 0; q < 4; ++q) { s.vr[q] = v[q].y; s.vc[q] = v[q].x; }     // [:, ::-1] -> (row, col)
      }
    }
    sh = s;
  }
  __syncthreads();
  const MaskShape s = sh;
  float* o = out + ((long)n * 10 + t) * H * W;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int r = i / W, c = i - r * W;
    float v = 0.f;
    if (s.kind == 1) v = 1.f;
    else if (s.kind == 2) v = (r >= s.y0 && r < s.y1 && c >= s.x0 && c < s.x1) ? 1.f : 0.f;
    else if (s.kind == 3) v = in_poly(s, (double)r, (double)c) ? 1.f : 0.f;
    o[i] = v;
  }
}

// (image / 255 - 0.5) * 2 in float64 then float32 (pose_utils.py:216-217 + Dataset.py:183,  .float()), HWC uint8 -> strided out
__global__ __launch_bounds__(256) void preprocess_image_kernel(const uint8_t* img, int H, int W, float* out, long oN, long oC,
                                                               long oH, long oW) {
  const int n = blockIdx.y;
  const uint8_t* b = img + (long)n * H * W * 3;
  float* o = out + (long)n * oN;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      o[(long)c * oC + (long)y * oH + (long)x * oW] = (float)(((double)b[(long)i * 3 + c] / 255.0 - 0.5) * 2.0);
  }
}

}  // namespace pg

using namespace pg;

extern "C" int pg_affine_transforms(const float* kp_from, const float* kp_to, int32_t N, int32_t P, float* out, void* stream) {
  PG_REQUIRE(kp_from && kp_to && out && N > 0 && (P == 16 || P == 18), "pg_affine_transforms: need pose_dim 16 or 18 (got %d)", P);
  PG_KLAUNCH(affine_transforms_kernel, dim3((N * 10 + 63) / 64), dim3(64), 0, (hipStream_t)stream, kp_from, kp_to, N, P,
                     joint_table(P), out);
  PG_LAUNCH_OK("pg_affine_transforms");
  return 0;
}

extern "C" int pg_uniform_transform(const float* kp_from, const float* kp_to, int32_t N, int32_t P, float* out, void* stream) {
  PG_REQUIRE(kp_from && kp_to && out && N > 0 && (P == 16 || P == 18), "pg_uniform_transform: need pose_dim 16 or 18 (got %d)", P);
  PG_KLAUNCH(uniform_transform_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, kp_from, kp_to, N, P,
                     joint_table(P), out);
  PG_LAUNCH_OK("pg_uniform_transform");
  return 0;
}

extern "C" int pg_pose_masks(const float* kp_to, int32_t N, int32_t P, int32_t H, int32_t W, float* out, void* stream) {
  PG_REQUIRE(kp_to && out && N > 0 && H > 0 && W > 0 && (P == 16 || P == 18), "pg_pose_masks: need pose_dim 16 or 18 (got %d)", P);
  int bx = (H * W + 255) / 256;
  if (bx > 64) bx = 64;
  PG_KLAUNCH(pose_masks_kernel, dim3(bx, 10, N), dim3(256), 0, (hipStream_t)stream, kp_to, P, H, W, joint_table(P), out);
  PG_LAUNCH_OK("pg_pose_masks");
  return 0;
}

extern "C" int pg_preprocess_image(const uint8_t* img, int32_t N, int32_t H, int32_t W, float* out, int64_t oN, int64_t oC,
                                   int64_t oH, int64_t oW, void* stream) {
  PG_REQUIRE(img && out && N > 0 && H > 0 && W > 0, "pg_preprocess_image: bad arguments");
  int bx = (H * W + 255) / 256;
  if (bx > 256) bx = 256;
  PG_KLAUNCH(preprocess_image_kernel, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, img, H, W, out, (long)oN,
                     (long)oC, (long)oH, (long)oW);
  PG_LAUNCH_OK("pg_preprocess_image");
  return 0;
}
