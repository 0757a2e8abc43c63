"""ORACLE (test infrastructure, never imported by the product) — NumPy restatement of the key-point geometry that
precedes the training step: reference src_deformable/utils/pose_transform.py:94-326 and the two scikit-image
primitives it calls.

PARITY UNPINNED for the two third-party primitives: scikit-image is not installed in this image, so
``estimate_affine`` and ``grid_points_in_poly`` restate the published algorithms of scikit-image 0.14.x
(``skimage/transform/_geometric.py``: ``AffineTransform`` inherits ``ProjectiveTransform.estimate``, which normalises both
point sets with ``_center_and_normalize_points``; ``skimage/measure/pnpoly.pyx``).  The reference pins no version (its
bytecode is CPython 3.6, 2018); a release whose estimate does NOT normalise the points gives different transforms for every
inconsistent over-determined fit — quantified on the fixture inputs by oracle/tls_normalisation_study.py
(profiles/round3_tls_normalisation_study.txt):

* ``skimage.transform.estimate_transform('affine', src, dst)`` -> ``AffineTransform.estimate``: both point sets are
  Hartley-normalised (centroid to the origin, RMS distance sqrt(2)), the 2n x 7 system
  ``[xs ys 1 0 0 0 xd; 0 0 0 xs ys 1 yd]`` is solved in the total-least-squares sense (right singular vector of the
  smallest singular value, ``params = -V[-1,:-1]/V[-1,-1]``), then ``H = inv(T_dst) @ Hn @ T_src``.
* ``skimage.measure.grid_points_in_poly(shape, verts)``: W. R. Franklin's pnpoly crossing-number test at every
  integer ``(row, col)`` with ``x = row`` against ``verts[:,0]`` and ``y = col`` against ``verts[:,1]``.

Everything above those two primitives (which joints feed which fit, polygon construction, mirrored limbs, the "no
point" transform, the head box, output layout) is pinned against the REAL reference functions, which
oracle/make_golden_r2.py imports with these two primitives installed as the ``skimage`` shim
(tests/golden/pose_geom.npz).  Known-answer tests of the primitives: tests/test_oracle_golden.py.
"""
import numpy as np

MISSING_VALUE = -1
# reference utils/pose_utils.py:25-35
LABELS = ['Rank', 'Rknee', 'Rhip', 'Lhip', 'Lknee', 'Lank', 'pelv', 'spine', 'neck', 'head', 'Rwri', 'Relb', 'Rsho',
          'Lsho', 'Lelb', 'Lwri']
LABELS_PAF = ['nose', 'neck', 'Rsho', 'Relb', 'Rwri', 'Lsho', 'Lelb', 'Lwri', 'Rhip', 'Rkne', 'Rank', 'Lhip', 'Lkne',
              'Lank', 'Leye', 'Reye', 'Lear', 'Rear']
NO_POINT = np.array([[1, 0, 1000], [0, 1, 1000], [0, 0, 1]], dtype=np.float64)       # pose_transform.py:221


# ------------------------------------------------------------------------------------------ third-party primitives
def _center_and_normalize(points):
    c = points.mean(axis=0)
    rms = np.sqrt(((points - c) ** 2).sum() / points.shape[0])
    nf = np.sqrt(2.0) / rms
    T = np.array([[nf, 0, -nf * c[0]], [0, nf, -nf * c[1]], [0, 0, 1]])
    return T, (points - c) * nf


def estimate_affine(src, dst):
    """AffineTransform.estimate(src, dst).params (3x3, float64)."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    Ts, s = _center_and_normalize(src)
    Td, d = _center_and_normalize(dst)
    n = src.shape[0]
    A = np.zeros((2 * n, 7))
    A[:n, 0], A[:n, 1], A[:n, 2], A[:n, 6] = s[:, 0], s[:, 1], 1, d[:, 0]
    A[n:, 3], A[n:, 4], A[n:, 5], A[n:, 6] = s[:, 0], s[:, 1], 1, d[:, 1]
    _, _, V = np.linalg.svd(A)
    H = np.zeros((3, 3))
    H.flat[[0, 1, 2, 3, 4, 5]] = -V[-1, :-1] / V[-1, -1]
    H[2, 2] = 1
    return np.linalg.inv(Td) @ H @ Ts


class _Tr:
    def __init__(self, params):
        self.params = params


def estimate_transform(ttype, src, dst):
    """drop-in for skimage.transform.estimate_transform (affine only)."""
    assert ttype == "affine"
    return _Tr(estimate_affine(src, dst))


def grid_points_in_poly(shape, verts):
    """drop-in for skimage.measure.grid_points_in_poly: (M, N) bool."""
    verts = np.asarray(verts, dtype=np.float64)
    xp, yp = verts[:, 0], verts[:, 1]
    x = np.arange(int(shape[0]), dtype=np.float64)[:, None]
    y = np.arange(int(shape[1]), dtype=np.float64)[None, :]
    c = np.zeros((int(shape[0]), int(shape[1])), dtype=bool)
    j = len(xp) - 1
    for i in range(len(xp)):
        cond = ((yp[i] <= y) & (y < yp[j])) | ((yp[j] <= y) & (y < yp[i]))
        with np.errstate(divide="ignore", invalid="ignore"):
            xi = (xp[j] - xp[i]) * (y - yp[i]) / (yp[j] - yp[i]) + xp[i]
        c ^= cond & (x < xi)
        j = i
    return c


# ------------------------------------------------------------------------------------------ the reference's own geometry
def give_name_to_keypoints(array, pose_dim):
    """pose_transform.py:94-104: {name: (x, y)} of the present joints."""
    names = LABELS if pose_dim == 16 else LABELS_PAF
    return {n: np.asarray(array[i][::-1]) for i, n in enumerate(names)
            if array[i][0] != MISSING_VALUE and array[i][1] != MISSING_VALUE}


def compute_st_distance(kp):
    """pose_transform.py:119-122."""
    d1 = np.sum((kp['Rhip'] - kp['Rsho']) ** 2)
    d2 = np.sum((kp['Lhip'] - kp['Lsho']) ** 2)
    return np.sqrt((d1 + d2) / 2.0)


def estimate_polygon(fr, to, st, inc_to, inc_from=0.1, p_to=0.2, p_from=0.2):
    """pose_transform.py:187-209."""
    fr = fr + (fr - to) * inc_from
    to = to + (to - fr) * inc_to
    nv = fr - to
    nv = np.array([-nv[1], nv[0]])
    norm = np.linalg.norm(nv)
    if norm == 0:
        return np.array([fr + 1, fr - 1, to - 1, to + 1])
    nv = nv / norm
    return np.array([fr + st * p_from * nv, fr - st * p_from * nv, to - st * p_to * nv, to + st * p_to * nv])


def _checked(tr):
    """to_transforms (pose_transform.py:224-230)."""
    try:
        np.linalg.inv(tr)
        return tr if np.all(np.isfinite(tr)) else NO_POINT
    except np.linalg.LinAlgError:
        return NO_POINT


_LIMBS = [('Rhip', 'Rkne', 0.1, 0.1), ('Lhip', 'Lkne', 0.1, 0.1), ('Rkne', 'Rank', 0.3, 0.5), ('Lkne', 'Lank', 0.3, 0.5),
          ('Rsho', 'Relb', 0.1, 0.1), ('Lsho', 'Lelb', 0.1, 0.1), ('Relb', 'Rwri', 0.3, 0.5), ('Lelb', 'Lwri', 0.3, 0.5)]
_HEAD = ('Leye', 'Reye', 'Lear', 'Rear', 'nose')


def _mirror(name):
    return ('L' if name[0] == 'R' else 'R') + name[1:]


def affine_transforms(array1, array2, pose_dim):
    """pose_transform.py:213-289 -> (10, 8) float64."""
    kp1, kp2 = give_name_to_keypoints(array1, pose_dim), give_name_to_keypoints(array2, pose_dim)
    st1, st2 = compute_st_distance(kp1), compute_st_distance(kp2)
    pts = lambda kp, names: np.array([kp[n] for n in names])
    out = []
    body = ['Rhip', 'Lhip', 'Lsho', 'Rsho']
    out.append(_checked(estimate_affine(pts(kp2, body), pts(kp1, body))))
    head = [n for n in _HEAD if n in kp1 and n in kp2]
    if head:
        head += ['Lsho', 'Rsho']
        out.append(_checked(estimate_affine(pts(kp2, head), pts(kp1, head))))
    else:
        out.append(NO_POINT)
    for fr, to, inc, _ in _LIMBS:
        if not (fr in kp2 and to in kp2):
            out.append(NO_POINT)
            continue
        f1, t1 = fr, to
        if not (f1 in kp1 and t1 in kp1):
            f1, t1 = _mirror(fr), _mirror(to)
            if not (f1 in kp1 and t1 in kp1):
                out.append(NO_POINT)
                continue
        p2 = estimate_polygon(kp2[fr], kp2[to], st2, inc)
        p1 = estimate_polygon(kp1[f1], kp1[t1], st1, inc)
        out.append(_checked(estimate_affine(p2, p1)))
    return np.array(out).reshape(-1, 9)[..., :-1]


def mask_from_kp_array(kp_array, border_inc, img_size):
    """pose_transform.py:125-137 (kp_array: integer (x, y) rows)."""
    lo = np.min(kp_array, axis=0) - int(border_inc)
    hi = np.max(kp_array, axis=0) + int(border_inc)
    lo = np.maximum(lo, 0)
    hi = np.minimum(hi, img_size[::-1])
    m = np.zeros(img_size)
    m[lo[1]:hi[1], lo[0]:hi[0]] = 1
    return m


def pose_masks(array2, img_size, pose_dim):
    """pose_transform.py:143-184 -> (10, H, W) float64."""
    kp2 = give_name_to_keypoints(array2, pose_dim)
    st2 = compute_st_distance(kp2)
    img_size = tuple(img_size)
    masks = [np.ones(img_size)]
    head = [n for n in _HEAD if n in kp2]
    if head:
        com = np.mean(np.array([kp2[n] for n in head]), axis=0, keepdims=True).astype(int)
        masks.append(mask_from_kp_array(com, 0.40 * st2, img_size))
    else:
        masks.append(np.zeros(img_size))
    for fr, to, _, inc in _LIMBS:
        if fr in kp2 and to in kp2:
            masks.append(grid_points_in_poly(img_size, estimate_polygon(kp2[fr], kp2[to], st2, inc)[:, ::-1]).astype(np.float64))
        else:
            masks.append(np.zeros(img_size))
    return np.array(masks)


def estimate_uniform_transform(array1, array2, pose_dim):
    """pose_transform.py:293-326 (warp_skip='full'): ONE torso(+knees) fit.  The reference returns 9 values when the
    fit is invertible and 8 otherwise (:322,:325); only the first six are ever read (:28)."""
    kp1, kp2 = give_name_to_keypoints(array1, pose_dim), give_name_to_keypoints(array2, pose_dim)
    names = ['Rhip', 'Lhip', 'Lsho', 'Rsho'] + [n for n in ('Rkne', 'Lkne') if n in kp1 and n in kp2]
    tr = _checked(estimate_affine(np.array([kp2[n] for n in names]), np.array([kp1[n] for n in names])))
    return tr.reshape(-1, 9)[..., :-1]
