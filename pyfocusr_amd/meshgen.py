"""Synthetic closed-manifold "blob" meshes for benchmarks and size-scaling tests
(SURVEY.md §8d, configs C3-C5).  Not part of the reference: the reference ships
only the four bone meshes, so BASELINE.json's 250k / 1M-vertex configs need a
generator.  Recipe: N Fibonacci-sphere directions, triangulated by their convex
hull (F = 2N-4, closed genus-0, degree 5-7), faces oriented outward, star-shaped
radial deformation with 6 random bumps and anisotropic axes, then a random vertex
permutation so that index order carries no spatial locality (as in scanned
meshes).  All randomness from `np.random.default_rng(seed)`; source seed 0, target
seed 1 by convention."""
import numpy as np

from .vtk_functions import PolyMesh


def fibonacci_sphere(n):
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0**0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def blob_mesh(n_points, seed=0, permute=True):
    """Closed triangle mesh with `n_points` vertices and 2*n_points-4 faces."""
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(seed)
    u = fibonacci_sphere(n_points)
    faces = ConvexHull(u).simplices.astype(np.int64)
    a, b, c = u[faces[:, 0]], u[faces[:, 1]], u[faces[:, 2]]
    inward = np.einsum("ij,ij->i", np.cross(b - a, c - a), a + b + c) < 0
    faces[inward] = faces[inward][:, [0, 2, 1]]

    axes = np.array([1.0, 0.62, 0.41]) * (1.0 + rng.uniform(-0.05, 0.05, 3))
    amp = rng.uniform(0.05, 0.15, 6)
    sharp = rng.uniform(2.0, 6.0, 6)
    centers = rng.normal(size=(6, 3))
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    r = 1.0 + np.sum(amp[None, :] * np.exp(sharp[None, :] * (u @ centers.T - 1.0)), axis=1)
    pts = 40.0 * u * axes[None, :] * r[:, None]

    if permute:
        perm = rng.permutation(n_points)  # new index of old vertex i is perm[i]
        new_pts = np.empty_like(pts)
        new_pts[perm] = pts
        pts = new_pts
        faces = perm[faces]
    return PolyMesh(pts, faces.astype(np.int32))


def messy_blob_mesh(n_points, seed=0, n_single=7, n_star=3, n_flip=2, n_fin=2):
    """`blob_mesh` with the defect classes of the reference's own scanned meshes (`source_mesh_15k` has
    duplicated directed edges, edges in three faces, one-way edges, unreferenced points) at the density of a cleaned-up
    scan: `n_single` faces deleted (a triangular hole each: three one-way edges), the faces around `n_star` vertices
    deleted (a hexagonal hole and a stranded vertex each), `n_flip` faces with reversed orientation (three directed
    edges listed twice, their reverses missing), `n_fin` extra faces on an existing edge (an edge in three faces).  With
    the defaults ~50 one-way entries: W is asymmetric (graph.py:178), L non-normal, `n_star` isolated vertices add to
    the null eigenvalues the reference's widen-and-retry loop (graph.py:374-379) has to step over."""
    m = blob_mesh(n_points, seed=seed)
    pts, faces = m.points, m.faces.copy()
    rng = np.random.default_rng(1000 + seed)
    n_faces = len(faces)
    pick = rng.choice(n_faces, size=n_single + n_flip + n_fin, replace=False)
    single, flip, fin = pick[:n_single], pick[n_single:n_single + n_flip], pick[n_single + n_flip:]
    faces[flip] = faces[flip][:, [0, 2, 1]]
    extra = [[faces[f][0], faces[f][1], faces[(f + n_faces // 2) % n_faces][0]] for f in fin]
    stars = rng.choice(n_points, size=n_star, replace=False)
    keep = np.ones(n_faces, dtype=bool)
    keep[single] = False
    keep &= ~np.isin(faces, stars).any(axis=1)
    faces = np.concatenate([faces[keep], np.asarray(extra, dtype=faces.dtype).reshape(-1, 3)])
    return PolyMesh(pts, faces.astype(np.int32))
