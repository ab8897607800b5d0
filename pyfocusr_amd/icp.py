"""ICP pre-alignment without VTK.

The reference registers one mesh to the other before building the graphs
(`/root/reference/pyfocusr/focusr.py:110-131` -> `vtk_functions.py:12-37`):
`vtkIterativeClosestPointTransform` with a rigid-body or similarity
`vtkLandmarkTransform`, `StartByMatchingCentroidsOn`, 100 iterations, no
mean-distance test, and (because `SetMaximumNumberOfLandmarks(1000)` comes after
the first `Update()` and re-triggers it) 1000 landmarks.

Here the iteration is VTK's published one; its only heavy step — the closest
surface point of every landmark, a `vtkCellLocator` query in VTK — runs on the
MI355X (`pf_surface_create / pf_surface_closest`, exact search).  The 3x3
cross-covariance and Horn's 4x4 eigenproblem per iteration are host work on
<= 1000 points.  VTK itself is absent from the build image, so parity with VTK is
unpinned; `tests/test_icp.py` checks this module against a brute-force CPU
restatement of the same algorithm.
"""
import numpy as np

from . import _hip


class _Matrix4x4(object):
    """The slice of vtkMatrix4x4 callers of `icp.GetMatrix()` use."""

    def __init__(self, m):
        self._m = m

    def GetElement(self, i, j):
        return float(self._m[i, j])

    def __array__(self, dtype=None, copy=None):
        return np.array(self._m, dtype=dtype)


class IcpTransform(object):
    """Result of `icp_transform`: `matrix` (4,4) maps source coordinates onto the target."""

    def __init__(self, matrix, n_iterations, mean_distance, n_landmarks, mode):
        self.matrix = matrix
        self.n_iterations = n_iterations
        self.mean_distance = mean_distance
        self.n_landmarks = n_landmarks
        self.mode = mode

    # vtkAbstractTransform-style accessors
    def GetMatrix(self):
        return _Matrix4x4(self.matrix)

    def GetNumberOfIterations(self):
        return self.n_iterations

    def GetMeanDistance(self):
        return self.mean_distance

    def TransformPoint(self, p):
        q = self.matrix[:3, :3] @ np.asarray(p, dtype=np.float64) + self.matrix[:3, 3]
        return (float(q[0]), float(q[1]), float(q[2]))

    def transform_points(self, points):
        return np.asarray(points, dtype=np.float64) @ self.matrix[:3, :3].T + self.matrix[:3, 3]


def landmark_transform(src, dst, mode="rigid"):
    """vtkLandmarkTransform (rigid body / similarity): Horn's closed-form quaternion solution."""
    n = len(src)
    sc, tc = src.sum(axis=0) / n, dst.sum(axis=0) / n
    m = np.eye(4)
    if n == 1:
        m[:3, 3] = tc - sc
        return m
    a, b = src - sc, dst - tc
    M = a.T @ b
    N = np.array([
        [M[0, 0] + M[1, 1] + M[2, 2], M[1, 2] - M[2, 1], M[2, 0] - M[0, 2], M[0, 1] - M[1, 0]],
        [M[1, 2] - M[2, 1], M[0, 0] - M[1, 1] - M[2, 2], M[0, 1] + M[1, 0], M[2, 0] + M[0, 2]],
        [M[2, 0] - M[0, 2], M[0, 1] + M[1, 0], -M[0, 0] + M[1, 1] - M[2, 2], M[1, 2] + M[2, 1]],
        [M[0, 1] - M[1, 0], M[2, 0] + M[0, 2], M[1, 2] + M[2, 1], -M[0, 0] - M[1, 1] + M[2, 2]],
    ])
    vals, vecs = np.linalg.eigh(N)
    w, x, y, z = vecs[:, np.argmax(vals)]
    R = np.array([
        [w * w + x * x - y * y - z * z, 2.0 * (-w * z + x * y), 2.0 * (w * y + x * z)],
        [2.0 * (w * z + x * y), w * w - x * x + y * y - z * z, 2.0 * (-w * x + y * z)],
        [2.0 * (-w * y + x * z), 2.0 * (w * x + y * z), w * w - x * x - y * y + z * z],
    ])
    if mode == "similarity":
        R = R * np.sqrt(float((b * b).sum()) / float((a * a).sum()))
    elif mode != "rigid":
        raise ValueError("Error invalid transform mode")
    m[:3, :3] = R
    m[:3, 3] = tc - R @ sc
    return m


def icp_transform(target_points, target_faces, source_points, numberOfIterations=100, number_landmarks=1000,
                  transform_mode="rigid", ctx=None, float_landmarks=True):
    """ICP of `source_points` onto the surface (target_points, target_faces).

    `float_landmarks` reproduces VTK's float32 `vtkPoints` for the landmark sets."""
    if transform_mode not in ("rigid", "similarity"):
        raise ValueError("Error invalid transform mode")
    tp = np.ascontiguousarray(target_points, dtype=np.float64).reshape(-1, 3)
    sp = np.ascontiguousarray(source_points, dtype=np.float64).reshape(-1, 3)
    if len(sp) == 0 or len(tp) == 0:
        raise ValueError("Can't execute with NULL or empty input")  # VTK's error text
    f32 = (lambda x: x.astype(np.float32).astype(np.float64)) if float_landmarks else (lambda x: x)
    surface = _hip.DeviceSurface(tp, target_faces, ctx=ctx)
    try:
        n = len(sp)
        step = n // number_landmarks if n > number_landmarks else 1
        nb = n // step
        acc = np.eye(4)
        acc[:3, 3] = tp.sum(axis=0) / len(tp) - sp.sum(axis=0) / n  # StartByMatchingCentroidsOn
        a = f32(sp[: nb * step : step] + acc[:3, 3])
        it = 0
        d2 = None
        while True:
            cp, _, d2 = surface.closest(a)
            if not np.all(np.isfinite(d2)):
                raise FloatingPointError("icp_transform: non-finite landmark or surface coordinates")
            m = landmark_transform(a, f32(cp), transform_mode)
            acc = m @ acc
            it += 1
            if it >= numberOfIterations:
                break
            a = f32(a @ m[:3, :3].T + m[:3, 3])
    finally:
        surface.close()
    return IcpTransform(acc, it, float(np.mean(np.sqrt(d2))), nb, transform_mode)
