"""ORACLE — test infrastructure, NOT product code.  **Parity unpinned.**

CPU restatement of the ICP pre-alignment the reference delegates to VTK
(`/root/reference/pyfocusr/vtk_functions.py:12-37`, called from `focusr.py:110-131`):
`vtk.vtkIterativeClosestPointTransform` with a `vtkLandmarkTransform` in rigid-body or
similarity mode, `StartByMatchingCentroidsOn`, 100 iterations, no mean-distance check.

The algorithm lives in the third-party `vtk` wheel (un-pinned: `requirements.txt:1-8`), which
is absent from the build image and from `/root/reference`, and the reference's tests hold no
vector for it — so this file restates VTK's *published* algorithm (VTK 9.x
`vtkIterativeClosestPointTransform::InternalUpdate`, `vtkLandmarkTransform::InternalUpdate`:
Horn's closed-form quaternion solution) and the HIP path is tested against THIS restatement,
not against VTK.  Known places where rounding (not the mathematics) may differ from VTK:
closest point on a triangle (VTK: `vtkTriangle::EvaluatePosition`; here Ericson's region
walk), the 4x4 symmetric eigenproblem (VTK: Jacobi; here LAPACK `eigh`), and VTK's float32
`vtkPoints` for the landmark sets (emulated by `float_landmarks=True`).

Effective number of landmarks: the reference calls `Update()` with VTK's default of 200 and
only then `SetMaximumNumberOfLandmarks(1000)` (`vtk_functions.py:26-28`); the setter bumps the
transform's MTime, so the first consumer (`vtkTransformPolyDataFilter`, `:32-37`) re-runs the
whole ICP with 1000 — which is what is restated here.
"""
import numpy as np


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def closest_point_on_triangles(p, a, b, c):
    """Closest point to `p` (3,) on each triangle (a,b,c) (T,3) — Ericson, Real-Time Collision
    Detection §5.1.5, evaluated for all triangles at once with the operation order of the HIP
    kernel (`pf_surface.hip: closest_on_triangle`).  Returns (T,3) points and (T,) squared distances."""
    with np.errstate(divide="ignore", invalid="ignore"):
        ab, ac, ap = b - a, c - a, p - a
        d1, d2 = _dot(ab, ap), _dot(ac, ap)
        bp = p - b
        d3, d4 = _dot(ab, bp), _dot(ac, bp)
        cp = p - c
        d5, d6 = _dot(ab, cp), _dot(ac, cp)
        vc = d1 * d4 - d3 * d2
        vb = d5 * d2 - d1 * d6
        va = d3 * d6 - d5 * d4
        out = np.empty_like(a)
        done = np.zeros(len(a), dtype=bool)

        def put(mask, value):
            m = mask & ~done
            out[m] = value[m]
            done[m] = True

        put((d1 <= 0) & (d2 <= 0), a)
        put((d3 >= 0) & (d4 <= d3), b)
        v = d1 / (d1 - d3)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + v[:, None] * ab)
        put((d6 >= 0) & (d5 <= d6), c)
        w = d2 / (d2 - d6)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + w[:, None] * ac)
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + w[:, None] * (c - b))
        denom = 1.0 / (va + vb + vc)
        v, w = vb * denom, vc * denom
        put(np.ones(len(a), dtype=bool), (a + ab * v[:, None]) + ac * w[:, None])
        diff = p - out
        dist2 = diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1] + diff[:, 2] * diff[:, 2]
    return out, dist2


def fan_triangles(faces):
    """(F,v) polygons -> (F*(v-2), 3) triangles (0,j+1,j+2), face-major: triangle t belongs to face t // (v-2)."""
    faces = np.asarray(faces)
    v = faces.shape[1]
    tris = np.stack([np.stack([faces[:, 0], faces[:, j + 1], faces[:, j + 2]], axis=1) for j in range(v - 2)], axis=1)
    return tris.reshape(-1, 3)


def closest_points_on_surface(points, faces, queries):
    """Brute force: for every query the closest point of the triangulated surface; lowest
    triangle index on exact distance ties; NaN distances never win.  Returns (pts (q,3), face (q,) i32, dist2 (q,))."""
    tris = fan_triangles(faces)
    per_face = np.asarray(faces).shape[1] - 2
    a, b, c = points[tris[:, 0]], points[tris[:, 1]], points[tris[:, 2]]
    out = np.empty((len(queries), 3))
    face = np.empty(len(queries), dtype=np.int32)
    dist2 = np.empty(len(queries))
    for i, p in enumerate(np.asarray(queries, dtype=np.float64)):
        cp, d2 = closest_point_on_triangles(p, a, b, c)
        t = int(np.argmin(np.where(np.isnan(d2), np.inf, d2)))
        out[i], face[i], dist2[i] = cp[t], t // per_face, d2[t]
    return out, face, dist2


def landmark_transform(src, dst, mode="rigid"):
    """4x4 matrix of vtkLandmarkTransform (rigid body / similarity): Horn's quaternion method."""
    n = len(src)
    sc, tc = src.sum(axis=0) / n, dst.sum(axis=0) / n
    if n == 1:
        m = np.eye(4)
        m[:3, 3] = tc - sc
        return m
    a, b = src - sc, dst - tc
    M = a.T @ b
    sa, sb = float((a * a).sum()), float((b * b).sum())
    N = np.empty((4, 4))
    N[0, 0] = M[0, 0] + M[1, 1] + M[2, 2]
    N[1, 1] = M[0, 0] - M[1, 1] - M[2, 2]
    N[2, 2] = -M[0, 0] + M[1, 1] - M[2, 2]
    N[3, 3] = -M[0, 0] - M[1, 1] + M[2, 2]
    N[0, 1] = N[1, 0] = M[1, 2] - M[2, 1]
    N[0, 2] = N[2, 0] = M[2, 0] - M[0, 2]
    N[0, 3] = N[3, 0] = M[0, 1] - M[1, 0]
    N[1, 2] = N[2, 1] = M[0, 1] + M[1, 0]
    N[1, 3] = N[3, 1] = M[2, 0] + M[0, 2]
    N[2, 3] = N[3, 2] = M[1, 2] + M[2, 1]
    vals, vecs = np.linalg.eigh(N)
    w, x, y, z = vecs[:, np.argmax(vals)]
    ww, wx, wy, wz = w * w, w * x, w * y, w * z
    xx, yy, zz = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    R = np.array([[ww + xx - yy - zz, 2.0 * (-wz + xy), 2.0 * (wy + xz)],
                  [2.0 * (wz + xy), ww - xx + yy - zz, 2.0 * (-wx + yz)],
                  [2.0 * (-wy + xz), 2.0 * (wx + yz), ww - xx - yy + zz]])
    if mode == "similarity":
        R = R * np.sqrt(sb / sa)
    elif mode != "rigid":
        raise ValueError("Error invalid transform mode")
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = tc - R @ sc
    return m


def _apply(m, pts):
    return pts @ m[:3, :3].T + m[:3, 3]


def icp(target_points, target_faces, source_points, n_iterations=100, n_landmarks=1000, mode="rigid",
        float_landmarks=True, closest=None):
    """vtkIterativeClosestPointTransform::InternalUpdate as the reference configures it.
    Returns the accumulated 4x4 matrix (source -> target).  `closest(queries)` may replace the
    brute-force closest-point search (the tests pass the HIP one to check the loop itself)."""
    if closest is None:
        def closest(q):
            return closest_points_on_surface(target_points, target_faces, q)[0]
    f32 = (lambda x: x.astype(np.float32).astype(np.float64)) if float_landmarks else (lambda x: x)
    n = len(source_points)
    step = n // n_landmarks if n > n_landmarks else 1
    nb = n // step
    acc = np.eye(4)
    acc[:3, 3] = target_points.sum(axis=0) / len(target_points) - source_points.sum(axis=0) / n  # StartByMatchingCentroids
    a = f32(_apply(acc, source_points[: nb * step : step]))
    it = 0
    while True:
        cp = f32(closest(a))
        m = landmark_transform(a, cp, mode)
        acc = m @ acc  # PostMultiply: the new step acts after the accumulated ones
        it += 1
        if it >= n_iterations:
            break
        a = f32(_apply(m, a))
    return acc
