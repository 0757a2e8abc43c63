"""ORACLE tooling (test infrastructure): how much does the Hartley point normalisation inside the restated
`AffineTransform.estimate` matter on the fixture inputs?  ("missing 2".)

The restatement in oracle/pose_geometry.py follows scikit-image 0.14.x (`skimage/transform/_geometric.py`):
`AffineTransform` inherits `ProjectiveTransform.estimate`, which calls `_center_and_normalize_points` on both point sets,
builds the 2n x 9 DLT system, keeps the columns of `self._coeffs` (range(6) for the affine class) plus the last one and takes
the right singular vector of the smallest singular value — a TOTAL least-squares fit, which is not scale invariant.  A release
that fits the un-normalised points therefore gives a different transform whenever the fit is not exact (every 4-corner limb
polygon).  scikit-image is not installed here, so which variant the reference's environment had cannot be checked; this
script quantifies the difference on the key-point sets of tests/golden/pose_geom.npz:
    python oracle/tls_normalisation_study.py        -> profiles/round3_tls_normalisation_study.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pose_geometry as G  # noqa: E402


def estimate_affine_unnormalised(src, dst):
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    n = src.shape[0]
    A = np.zeros((2 * n, 7))
    A[:n, 0], A[:n, 1], A[:n, 2], A[:n, 6] = src[:, 0], src[:, 1], 1, dst[:, 0]
    A[n:, 3], A[n:, 4], A[n:, 5], A[n:, 6] = src[:, 0], src[:, 1], 1, dst[:, 1]
    _, _, V = np.linalg.svd(A)
    H = np.zeros((3, 3))
    H.flat[[0, 1, 2, 3, 4, 5]] = -V[-1, :-1] / V[-1, -1]
    H[2, 2] = 1
    return H


def study():
    fix = np.load(os.path.join(ROOT, "tests", "golden", "pose_geom.npz"))
    tags = sorted({k[:-4] for k in fix.files if k.endswith("_kp1")})
    lines, worst = [], 0.0
    for tag in tags:
        P = int(tag.split("_")[0][1:])
        H, W = [int(v) for v in tag.split("_")[1].split("x")]
        k1, k2 = fix[tag + "_kp1"], fix[tag + "_kp2"]
        yy, xx = np.mgrid[0:H, 0:W]
        pts = np.stack([xx.ravel(), yy.ravel(), np.ones(H * W)], 0).astype(np.float64)
        disp = []
        for i in range(len(k1)):
            ref = G.affine_transforms(k1[i], k2[i], P)
            orig = G.estimate_affine
            G.estimate_affine = estimate_affine_unnormalised
            try:
                alt = G.affine_transforms(k1[i], k2[i], P)
            finally:
                G.estimate_affine = orig
            for r, a in zip(ref, alt):
                if r[2] == 1000 or a[2] == 1000:            # the "no point" row
                    continue
                Mr = np.array([[r[0], r[1], r[2]], [r[3], r[4], r[5]]])
                Ma = np.array([[a[0], a[1], a[2]], [a[3], a[4], a[5]]])
                disp.append(float(np.abs(Mr @ pts - Ma @ pts).max()))
        d = np.array(disp)
        worst = max(worst, float(np.median(d)))
        lines.append("%-14s %3d fits: largest displacement of a warped in-image pixel between the two variants — median %.3f px, "
                     "90th percentile %.3g px, max %.3g px; %d fits differ by more than 1 px, %d by less than 0.01 px"
                     % (tag, len(d), np.median(d), np.percentile(d, 90), d.max(), int((d > 1).sum()), int((d < 0.01).sum())))
    return lines, worst


if __name__ == "__main__":
    lines, worst = study()
    hdr = ["# oracle/tls_normalisation_study.py: Hartley-normalised (restated, scikit-image 0.14.x ProjectiveTransform.estimate) vs",
           "# un-normalised total-least-squares affine fits on the key-point sets of tests/golden/pose_geom.npz (reference",
           "# utils/pose_transform.py:213-289: 10 limb / body fits per sample)."]
    out = "\n".join(hdr + lines + [
        "# Consistent point sets (the limb parallelograms estimate_polygon builds from two joints: ~70 %% of the fits) agree to rounding;",
        "# inconsistent over-determined sets (body: 4 torso joints, head: face joints + shoulders, of random fixture key-points) differ by",
        "# whole pixels to hundreds of pixels, and the un-normalised system is occasionally near-singular",
        "# (last component of the singular vector ~ 0).  So the choice of variant matters and cannot be settled without the reference's",
        "# scikit-image: f1 stays 'parity unpinned' for this primitive (largest per-case median: %.3f px)." % worst]) + "\n"
    print(out)
    with open(os.path.join(ROOT, "profiles", "round3_tls_normalisation_study.txt"), "w") as f:
        f.write(out)
