"""CPD registration of spectral coordinates.  cycpd is absent here, so parity with it is
unpinned (see `oracle/cpd_port.py`); these tests pin the oracle's mathematics on the CPU and the HIP E-step /
affinity products / full registrations against that oracle through the C-ABI."""
import numpy as np
import pytest

from oracle import cpd_port


def _clouds(seed, N=400, M=350, D=3):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(N, D))
    A = np.eye(D) + 0.15 * rng.normal(size=(D, D))
    Y = (X[rng.choice(N, M, replace=False)] - 0.2) @ np.linalg.inv(A) + 0.01 * rng.normal(size=(M, D))
    return X, Y


# ------------------------------------------------------------------------------ CPU: the oracle itself
def test_oracle_expectation_is_a_posterior():
    X, Y = _clouds(0)
    P1, Pt1, PX, Np = cpd_port.expectation(X, Y, 0.3, w=0.0)
    np.testing.assert_allclose(Pt1, 1.0, rtol=1e-12)          # w = 0: every x_n is fully explained
    np.testing.assert_allclose(Np, len(X), rtol=1e-12)
    P1w, Pt1w, _, _ = cpd_port.expectation(X, Y, 0.3, w=0.3)
    assert np.all(Pt1w < 1.0) and np.all(Pt1w > 0.0) and P1w.sum() < P1.sum()


def test_oracle_affine_recovers_affine_map():
    rng = np.random.default_rng(1)
    X = rng.normal(size=(300, 3))
    B, t = np.eye(3) + 0.1 * rng.normal(size=(3, 3)), np.array([0.2, -0.1, 0.3])
    Y = (X - t) @ np.linalg.inv(B)
    reg = cpd_port.AffineRegistration(X, Y, max_iterations=200, tolerance=1e-12)
    TY, (Bf, tf) = reg.register()
    np.testing.assert_allclose(TY, X, atol=1e-5)
    np.testing.assert_allclose(Bf, B, atol=1e-5)


def test_oracle_low_rank_equals_full_rank_when_rank_is_full():
    X, Y = _clouds(2, N=120, M=90)
    kw = dict(alpha=0.7, beta=0.8, max_iterations=15, tolerance=0.0)
    full = cpd_port.DeformableRegistration(X, Y, low_rank=False, **kw).register()[0]
    low = cpd_port.DeformableRegistration(X, Y, low_rank=True, num_eig=90, eig_floor=0.0, **kw).register()[0]
    np.testing.assert_allclose(low, full, atol=1e-7)


def test_product_cpd_does_not_import_oracle():
    import pyfocusr_amd.cpd as mod

    src = open(mod.__file__).read()
    assert "oracle" not in src


def test_sigma2_initialisation_formula():
    from pyfocusr_amd import cpd

    X, Y = _clouds(3, D=5)
    np.testing.assert_allclose(cpd.initialize_sigma2(X, Y + 7.0), cpd_port.initialize_sigma2(X, Y + 7.0), rtol=1e-12)


# ------------------------------------------------------------------------------ GPU: HIP vs oracle
@pytest.fixture(scope="module")
def ctx():
    from pyfocusr_amd import _hip

    _hip.load_library()
    return _hip.default_context()


@pytest.mark.gpu
@pytest.mark.parametrize("D", [1, 3, 6, 9, 16])
@pytest.mark.parametrize("w", [0.0, 0.2])
def test_estep_matches_oracle(ctx, D, w):
    from pyfocusr_amd import _hip

    X, Y = _clouds(10 + D, N=1300, M=1111, D=D)   # not multiples of the tile / chunk sizes
    dev = _hip.DeviceCpd(X, Y, ctx=ctx)
    for sigma2 in (0.5, 0.02):
        P1, Pt1, PX = dev.estep(Y, sigma2, w)
        wP1, wPt1, wPX, _ = cpd_port.expectation(X, Y, sigma2, w)
        np.testing.assert_allclose(Pt1, wPt1, rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(P1, wP1, rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(PX, wPX, rtol=1e-10, atol=1e-13)
    # far-apart clouds: every column sum underflows to 0 -> eps rule, all outputs 0 and finite
    P1, Pt1, PX = dev.estep(Y + 1e3, 1e-3, 0.0)
    assert np.all(Pt1 == 0) and np.all(P1 == 0) and np.all(PX == 0)
    with pytest.raises(_hip.PfError):
        dev.estep(Y, -1.0, 0.0)
    dev.close()


@pytest.mark.gpu
def test_gram_product_and_low_rank(ctx):
    from pyfocusr_amd import _hip, cpd

    rng = np.random.default_rng(4)
    A, B = rng.normal(size=(700, 4)), rng.normal(size=(333, 4))
    for cols in (1, 8, 13):
        V = rng.normal(size=(333, cols))
        np.testing.assert_allclose(_hip.gaussian_gram_product(A, B, 0.9, V, ctx=ctx), cpd_port.gaussian_kernel(A, 0.9, B) @ V,
                                   rtol=1e-12, atol=1e-12)
    Y = rng.normal(size=(600, 3))
    Q, S = cpd.low_rank_affinity(Y, 1.5, 40, ctx=ctx)
    wQ, wS = cpd_port.low_rank_eigen(cpd_port.gaussian_kernel(Y, 1.5), 40)
    np.testing.assert_allclose(S, wS, rtol=1e-9, atol=1e-13 * wS[0])
    np.testing.assert_allclose(Q.T @ Q, np.eye(40), atol=1e-10)
    big = wS > 1e-6 * wS[0]   # eigenvectors are comparable where the eigenvalues are separated from the noise
    np.testing.assert_allclose(np.abs(np.sum(Q[:, big] * wQ[:, big], axis=0)), 1.0, atol=1e-6)


@pytest.mark.gpu
def test_affine_registration_equals_oracle(ctx):
    from pyfocusr_amd import cpd

    X, Y = _clouds(5, N=900, M=800, D=6)
    got = cpd.affine_registration(X=X, Y=Y, max_iterations=40, tolerance=0.0, ctx=ctx)
    TY, (B, t) = got.register()
    want = cpd_port.AffineRegistration(X, Y, max_iterations=40, tolerance=0.0)
    wTY, (wB, wt) = want.register()
    assert got.iteration == 40
    np.testing.assert_allclose(B, wB, atol=1e-9)
    np.testing.assert_allclose(t, wt, atol=1e-9)
    np.testing.assert_allclose(TY, wTY, atol=1e-9)
    np.testing.assert_allclose(got.sigma2, want.sigma2, rtol=1e-8)
    extra = np.random.default_rng(0).normal(size=(50, 6))
    np.testing.assert_allclose(got.transform_point_cloud(extra), extra @ wB + wt, atol=1e-9)
    # the reference's stopping rule (tolerance on the objective) stops both at the same iteration
    a = cpd.affine_registration(X=X, Y=Y, max_iterations=100, tolerance=1e-8, ctx=ctx)
    a.register()
    b = cpd_port.AffineRegistration(X, Y, max_iterations=100, tolerance=1e-8)
    b.register()
    assert abs(a.iteration - b.iteration) <= 1 and a.iteration < 100


@pytest.mark.gpu
def test_deformable_registration_equals_oracle(ctx):
    from pyfocusr_amd import cpd

    rng = np.random.default_rng(6)
    X = rng.normal(size=(700, 3))
    Y = X[:600] + 0.15 * np.sin(2.0 * X[:600][:, [1, 2, 0]]) + 0.01 * rng.normal(size=(600, 3))   # smooth warp
    kw = dict(alpha=0.5, beta=1.0, num_eig=60, max_iterations=30, tolerance=0.0)
    got = cpd.deformable_registration(X=X, Y=Y, ctx=ctx, **kw)
    TY, (Q, S, W) = got.register()
    want = cpd_port.DeformableRegistration(X, Y, low_rank=True, **kw)
    wTY, _ = want.register()
    assert got.iteration == 30
    np.testing.assert_allclose(got.sigma2, want.sigma2, rtol=1e-6)
    np.testing.assert_allclose(TY, wTY, atol=1e-7)
    assert got.sigma2 < 0.5 * cpd.initialize_sigma2(X, Y)   # the mixture is tightening around the fixed set
    extra = rng.normal(size=(1234, 3))
    np.testing.assert_allclose(got.transform_point_cloud(extra), want.transform_point_cloud(extra), atol=1e-7)


@pytest.mark.gpu
def test_tiny_and_ragged_sizes(ctx):
    """One-point sets, sets smaller than one tile / one basis block, K clipped to M."""
    from pyfocusr_amd import _hip, cpd

    rng = np.random.default_rng(8)
    for N, M, D in ((1, 1, 2), (3, 1, 1), (1, 7, 3), (5, 9, 4), (130, 3, 2)):
        X, Y = rng.normal(size=(N, D)), rng.normal(size=(M, D))
        dev = _hip.DeviceCpd(X, Y, ctx=ctx)
        P1, Pt1, PX = dev.estep(Y, 0.7, 0.1)
        wP1, wPt1, wPX, _ = cpd_port.expectation(X, Y, 0.7, 0.1)
        np.testing.assert_allclose(P1, wP1, rtol=1e-12)
        np.testing.assert_allclose(Pt1, wPt1, rtol=1e-12)
        np.testing.assert_allclose(PX, wPX, rtol=1e-11, atol=1e-14)
        dev.close()
    X, Y = rng.normal(size=(40, 2)), rng.normal(size=(9, 2))
    got = cpd.deformable_registration(X=X, Y=Y, alpha=1.0, beta=1.0, num_eig=100, max_iterations=10, tolerance=0.0, ctx=ctx)
    TY, (Q, S, W) = got.register()
    want = cpd_port.DeformableRegistration(X, Y, alpha=1.0, beta=1.0, num_eig=100, max_iterations=10, tolerance=0.0)
    np.testing.assert_allclose(TY, want.register()[0], atol=1e-8)
    assert Q.shape[0] == 9 and Q.shape[1] <= 9 and W.shape == (9, 2)
    a = cpd.affine_registration(X=X, Y=Y, max_iterations=5, tolerance=0.0, ctx=ctx)
    b = cpd_port.AffineRegistration(X, Y, max_iterations=5, tolerance=0.0)
    np.testing.assert_allclose(a.register()[0], b.register()[0], atol=1e-10)
    with pytest.raises(ValueError):
        cpd.affine_registration(X=X, Y=rng.normal(size=(9, 3)), ctx=ctx)


@pytest.mark.gpu
def test_focusr_full_defaults_run_without_cycpd_or_vtk(golden, ctx):
    """`Focusr(target, source)` with EVERY argument at the reference's default (ICP first, curvature features
    requested but unused, affine + deformable CPD on 5000 sampled points, smoothing, both outputs), then
    `align_maps()`: runs on the device without vtk/cycpd, and CPD tightens the spectral match."""
    from pyfocusr_amd import Focusr, PolyMesh

    gt, gs = golden("target_mesh"), golden("source_mesh")
    reg = Focusr(PolyMesh(gt["points"], gt["faces"]), PolyMesh(gs["points"], gs["faces"]), ctx=ctx)
    np.testing.assert_allclose(reg.graph_target.eig_vals, gt["k6_eig_vals"], rtol=1e-8)
    reg.align_maps()
    idx = reg.corresponding_target_idx_for_each_source_pt
    assert idx.shape == (5000,) and idx.min() >= 0 and idx.max() < 5000
    assert len(reg.rigid_params) == 2 and len(reg.non_rigid_params) == 3
    # registration quality: the registered target cloud is closer to the source cloud than before CPD
    before = np.sqrt(ctx.knn(reg.graph_target.eig_vecs[:, :3] * reg.spectral_weights[None, :3],
                             reg.source_spectral_coords_b4_reg, 1)[1]).mean()
    after = np.sqrt(ctx.knn(reg.target_spectral_coords, reg.source_spectral_coords, 1)[1]).mean()
    assert after < before
    assert reg.weighted_avg_transformed_points.shape == (5000, 3)
    assert reg.nearest_neighbor_transformed_points.shape == (5000, 3)
    reg.set_all_mesh_scalars_to_corresp_target_idx()   # focusr.py:572-599
    assert np.array_equal(reg.graph_source.vtk_mesh.scalars, idx) and np.array_equal(reg.weighted_avg_transformed_mesh.scalars, idx)
    assert np.array_equal(reg.graph_target.vtk_mesh.scalars, np.arange(5000))
    # two meshes of the same bone: corresponding points lie within a few percent of the bone's size
    d = np.linalg.norm(reg.weighted_avg_transformed_points - reg.graph_source.points, axis=1)
    assert np.median(d) < 0.05 * np.ptp(gt["points"])


@pytest.mark.gpu
def test_deformable_with_wide_coordinates(ctx):
    """Regression (randomised pipeline sweep): K x d above the initial scratch size (100 eigenvectors x 8 coordinates,
    i.e. spectral coordinates + xyz) must not overflow the device-resident M-step's buffers."""
    from pyfocusr_amd import cpd

    rng = np.random.default_rng(12)
    X = rng.normal(size=(900, 8))
    Y = X[:700] + 0.05 * rng.normal(size=(700, 8))
    kw = dict(alpha=0.5, beta=1.2, num_eig=100, max_iterations=8, tolerance=0.0)
    got = cpd.deformable_registration(X=X, Y=Y, ctx=ctx, **kw)
    TY, (Q, S, W) = got.register()
    want = cpd_port.DeformableRegistration(X, Y, low_rank=True, **kw)
    np.testing.assert_allclose(TY, want.register()[0], atol=1e-7)
    assert Q.shape[1] * 8 > 561


@pytest.mark.gpu
def test_estep_denormal_column_sums(ctx):
    """Regression (randomised sweep): column sums in the denormal range (far-apart clouds in many dimensions, small
    sigma2, w = 0) must not turn into inf * 0 = NaN: numpy's P / den stays finite there."""
    from pyfocusr_amd import _hip

    rng = np.random.default_rng(4)
    found = False
    for trial in range(40):
        D = int(rng.integers(8, 17))
        X, Y = rng.normal(size=(300, D)) * 0.3, rng.normal(size=(200, D)) * 0.3 + rng.uniform(-1, 1)
        s2 = float(10 ** rng.uniform(-2.6, -1.2))
        wP1, wPt1, wPX, _ = cpd_port.expectation(X, Y, s2, 0.0)
        col = np.exp(-np.sum((X[None] - Y[:, None]) ** 2, axis=2) / (2 * s2)).sum(axis=0)
        tiny = np.any((col > 0) & (col < 1e-300))
        dev = _hip.DeviceCpd(X, Y, ctx=ctx)
        P1, Pt1, PX = dev.estep(Y, s2, 0.0)
        dev.close()
        assert np.all(np.isfinite(P1)) and np.all(np.isfinite(Pt1)) and np.all(np.isfinite(PX))
        if tiny:
            found = True
            np.testing.assert_allclose(Pt1, wPt1, rtol=1e-6, atol=1e-300)
            np.testing.assert_allclose(P1, wP1, rtol=1e-6, atol=1e-12)
    assert found, "no trial reached the denormal range: adjust the generator"
