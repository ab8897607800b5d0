"""ICP pre-alignment.  VTK is absent here, so parity with VTK itself is unpinned
(see `oracle/icp_port.py`); these tests pin (i) the oracle's own mathematics on the CPU and (ii) the
HIP closest-point search + ICP loop against that oracle through the C-ABI."""
import numpy as np
import pytest

from oracle import icp_port


def _rotation(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


# ------------------------------------------------------------------------------ CPU: the oracle itself
def test_oracle_closest_point_regions():
    """Every Voronoi region of one triangle (3 vertices, 3 edges, interior) against hand-computed answers."""
    a, b, c = np.array([[0.0, 0, 0]]), np.array([[2.0, 0, 0]]), np.array([[0.0, 2, 0]])
    cases = [((-1, -1, 1), (0, 0, 0)), ((3, -1, 0), (2, 0, 0)), ((-1, 3, 0), (0, 2, 0)), ((1, -2, 5), (1, 0, 0)),
             ((-3, 1, 1), (0, 1, 0)), ((2, 2, 0), (1, 1, 0)), ((0.5, 0.5, 7), (0.5, 0.5, 0))]
    for p, want in cases:
        cp, d2 = icp_port.closest_point_on_triangles(np.array(p, dtype=np.float64), a, b, c)
        np.testing.assert_allclose(cp[0], want, atol=1e-15)
        np.testing.assert_allclose(d2[0], np.sum((np.array(p) - want) ** 2), rtol=1e-15)


def test_oracle_closest_point_is_a_minimum():
    """No sampled point of any triangle is closer than the reported closest point."""
    rng = np.random.default_rng(0)
    a, b, c = rng.normal(size=(3, 40, 3))
    u = rng.random((2000, 2))
    u[u.sum(1) > 1] = 1 - u[u.sum(1) > 1]
    for p in rng.normal(size=(20, 3)) * 2:
        cp, d2 = icp_port.closest_point_on_triangles(p, a, b, c)
        for t in range(len(a)):
            samples = a[t] + u[:, :1] * (b[t] - a[t]) + u[:, 1:] * (c[t] - a[t])
            assert d2[t] <= np.min(np.sum((samples - p) ** 2, axis=1)) * (1 + 1e-12)
            # and the closest point lies in the triangle's plane, inside it (barycentric coordinates in [0,1])
            sol = np.linalg.lstsq(np.stack([b[t] - a[t], c[t] - a[t]], axis=1), cp[t] - a[t], rcond=None)[0]
            assert sol.min() > -1e-9 and sol.sum() < 1 + 1e-9


@pytest.mark.parametrize("mode", ["rigid", "similarity"])
def test_oracle_landmark_transform_recovers_motion(mode):
    rng = np.random.default_rng(1)
    src = rng.normal(size=(50, 3))
    R, t, s = _rotation([1, 2, 3], 0.7), np.array([0.3, -2.0, 5.0]), (1.7 if mode == "similarity" else 1.0)
    m = icp_port.landmark_transform(src, s * src @ R.T + t, mode)
    np.testing.assert_allclose(m[:3, :3], s * R, atol=1e-12)
    np.testing.assert_allclose(m[:3, 3], t, atol=1e-12)
    one = icp_port.landmark_transform(src[:1], src[:1] + 2.0, mode)
    np.testing.assert_allclose(one[:3, 3], [2.0, 2.0, 2.0])


def test_oracle_icp_recovers_small_rigid_motion(golden):
    g = golden("target_mesh")
    pts, faces = g["points"], g["faces"]
    R, t = _rotation([0.2, 1.0, -0.4], 0.05), np.array([0.4, -0.3, 0.2])
    moved = pts @ R.T + t
    m = icp_port.icp(pts, faces, moved[::25], n_iterations=30, n_landmarks=1000, float_landmarks=False)
    back = moved @ m[:3, :3].T + m[:3, 3]
    assert np.max(np.linalg.norm(back - pts, axis=1)) < 1e-3 * np.ptp(pts)


def test_product_icp_does_not_import_oracle():
    import pyfocusr_amd.icp as mod

    src = open(mod.__file__).read()
    assert "import oracle" not in src and "from oracle" not in src


# ------------------------------------------------------------------------------ GPU: HIP vs oracle
@pytest.fixture(scope="module")
def ctx():
    from pyfocusr_amd import _hip

    _hip.load_library()
    return _hip.default_context()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["target_mesh", "source_mesh_15k"])
def test_closest_points_bit_exact(golden, ctx, name):
    from pyfocusr_amd import _hip

    g = golden(name)
    pts, faces = g["points"], g["faces"]
    rng = np.random.default_rng(3)
    box_lo, box_hi = pts.min(0), pts.max(0)
    q = np.concatenate([
        pts[::37] + rng.normal(size=pts[::37].shape) * 0.5,       # near the surface
        rng.uniform(box_lo - 20, box_hi + 20, size=(150, 3)),     # far away / inside
        pts[:40],                                                  # exactly on vertices (distance 0, many ties)
        (pts[faces[:40, 0]] + pts[faces[:40, 1]]) / 2,             # on edges: two triangles tie
    ])
    surf = _hip.DeviceSurface(pts, faces, ctx=ctx)
    cp, face, d2 = surf.closest(q)
    surf.close()
    want_cp, want_face, want_d2 = icp_port.closest_points_on_surface(pts, faces, q)
    assert np.array_equal(d2, want_d2)
    assert np.array_equal(face, want_face)
    assert np.array_equal(cp, want_cp)


@pytest.mark.gpu
def test_closest_points_quads_degenerate_and_nan(ctx):
    from pyfocusr_amd import _hip

    rng = np.random.default_rng(5)
    pts = rng.normal(size=(300, 3))
    faces = rng.integers(0, 300, size=(500, 4)).astype(np.int32)   # quads, many degenerate (repeated vertices)
    faces[:20, 1] = faces[:20, 0]
    faces[20:30] = faces[20:30, :1]                                 # all four corners equal: a point
    q = rng.normal(size=(400, 3)) * 2
    surf = _hip.DeviceSurface(pts, faces, ctx=ctx)
    cp, face, d2 = surf.closest(q)
    want_cp, want_face, want_d2 = icp_port.closest_points_on_surface(pts, faces, q)
    assert np.array_equal(d2, want_d2) and np.array_equal(face, want_face) and np.array_equal(cp, want_cp)
    bad = q[:3].copy()
    bad[1, 2] = np.nan
    cp, face, d2 = surf.closest(bad)
    assert face[1] == -1 and np.isinf(d2[1]) and np.all(np.isnan(cp[1])) and face[0] >= 0 and face[2] >= 0
    assert surf.closest(np.zeros((0, 3)))[0].shape == (0, 3)
    surf.close()
    with pytest.raises(_hip.PfError):
        _hip.DeviceSurface(pts, np.array([[0, 1, 300]], dtype=np.int32), ctx=ctx)


@pytest.mark.gpu
def test_closest_points_tiny_surfaces(ctx):
    """One triangle; fewer triangles than a chunk; exactly one chunk; one more than a chunk / a super-chunk."""
    from pyfocusr_amd import _hip

    rng = np.random.default_rng(9)
    for n_tri in (1, 5, 64, 65, 64 * 64, 64 * 64 + 1):
        pts = rng.normal(size=(max(3, n_tri // 2 + 3), 3))
        faces = rng.integers(0, len(pts), size=(n_tri, 3)).astype(np.int32)
        q = rng.normal(size=(23, 3)) * 1.5
        surf = _hip.DeviceSurface(pts, faces, ctx=ctx)
        cp, face, d2 = surf.closest(q)
        surf.close()
        want_cp, want_face, want_d2 = icp_port.closest_points_on_surface(pts, faces, q)
        assert np.array_equal(d2, want_d2) and np.array_equal(face, want_face) and np.array_equal(cp, want_cp), n_tri


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rigid", "similarity"])
def test_icp_equals_oracle_loop(golden, ctx, mode):
    """The reference's configuration (1000 landmarks, centroid start, no mean-distance test) on the 5k pair; 30
    instead of 100 iterations keep the brute-force CPU side of the comparison under a minute."""
    from pyfocusr_amd import icp

    gt, gs = golden("target_mesh"), golden("source_mesh")
    got = icp.icp_transform(gt["points"], gt["faces"], gs["points"], numberOfIterations=30, transform_mode=mode, ctx=ctx)
    want = icp_port.icp(gt["points"], gt["faces"], gs["points"], n_iterations=30, mode=mode)
    assert got.n_iterations == 30 and got.n_landmarks == 1000
    np.testing.assert_allclose(got.matrix, want, rtol=0, atol=1e-12)
    R = got.matrix[:3, :3]
    scale = np.cbrt(np.linalg.det(R))
    np.testing.assert_allclose(R @ R.T, scale ** 2 * np.eye(3), atol=1e-12)
    if mode == "rigid":
        assert abs(scale - 1) < 1e-12


@pytest.mark.gpu
def test_icp_recovers_known_motion_and_focusr_defaults(golden, ctx):
    """ICP undoes a small rigid motion; `Focusr` with its DEFAULT `icp_register_first=True` runs without VTK
    and — the spectrum being invariant to rigid motion — reproduces the reference eigenvalues."""
    from pyfocusr_amd import Focusr, PolyMesh, vtk_functions

    gt = golden("target_mesh")
    pts, faces = gt["points"], gt["faces"]
    R, t = _rotation([0.2, 1.0, -0.4], 0.06), np.array([0.5, -0.4, 0.3])
    moved = PolyMesh(pts @ R.T + t, faces)
    tr = vtk_functions.icp_transform(target=PolyMesh(pts, faces), source=moved, ctx=ctx)
    back = vtk_functions.apply_transform(moved, tr)
    assert isinstance(back, PolyMesh)
    assert np.max(np.linalg.norm(back.points - pts, axis=1)) < 1e-4 * np.ptp(pts)
    assert tr.GetMatrix().GetElement(3, 3) == 1.0 and len(tr.TransformPoint(moved.points[0])) == 3

    gs = golden("source_mesh")
    reg = Focusr(PolyMesh(pts, faces), PolyMesh(gs["points"], gs["faces"]), n_spectral_features=3, n_extra_spectral=3,
                 ctx=ctx, registration=lambda src, tgt, kind: tgt)   # list_features_to_calc left at its default
    assert reg.icp_transform is not None and reg.icp_transform.n_iterations == 100
    np.testing.assert_allclose(reg.graph_target.eig_vals, gt["k6_eig_vals"], rtol=1e-8)
    np.testing.assert_allclose(reg.graph_source.eig_vals, gs["k6_eig_vals"], rtol=1e-7)  # points moved by a rigid map
    reg.align_maps()
    assert reg.corresponding_target_idx_for_each_source_pt.shape == (5000,)
