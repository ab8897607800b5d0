"""`Focusr` — orchestration shell around the MI355X spectral hot path.

Mirror of `/root/reference/pyfocusr/focusr.py` (`class Focusr` :22-807): the
constructor takes the same arguments in the same order (target mesh first) with
the same defaults, builds both graphs and their spectra eagerly (:134-170), and
`align_maps()` (:514-568) runs eigenmap sorting -> spectral coordinates ->
[CPD registration] -> nearest-neighbour correspondence -> [smoothing].

On the device: Laplacian assembly + eigensolve (`Graph`), the 3-D NN of `eigsort`,
`get_kd_correspondence` (focusr.py:351-353 -> `pf_knn1`), the graph mean filter
behind `get_smoothed_correspondences` (focusr.py:368-396) and the 3-NN query of
`get_weighted_final_node_locations` (focusr.py:401-426 -> `pf_knn`, k = 3).

"Next" rows of SURVEY.md §8f:
* ICP pre-alignment (:110-131): `pyfocusr_amd/icp.py` (closest-point search on the
  device) unless both meshes are real vtkPolyData and `vtk` is importable;
* CPD registration (:297-334): an injected callable `registration(source_coords,
  target_coords, kind) -> new_target_coords` if given, else the third-party
  `cycpd` when importable (what the reference calls), else `pyfocusr_amd/cpd.py`
  (same algorithm, E-step and affinity products on the device);
* Hungarian point correspondence (:340-349): scipy on the host, as in the reference.
"""
import numpy as np

from . import _hip
from .eigsort import eigsort
from .graph import Graph, compute_spectra, spectral_knn
from .main import print_header
from .vtk_functions import PolyMesh, apply_transform, icp_transform, set_mesh_scalars, vtk_deep_copy

from . import cpd as _native_cpd

try:  # pragma: no cover - not installed in the build image
    import cycpd
except Exception:  # noqa: BLE001
    cycpd = None

__all__ = ["Focusr"]


class Focusr(object):
    def __init__(
        self,
        vtk_mesh_target,
        vtk_mesh_source,
        icp_register_first=True,
        icp_registration_mode="rigid",
        icp_reg_target_to_source=False,
        n_spectral_features=3,
        n_extra_spectral=3,
        target_eigenmap_as_reference=True,
        norm_physical_and_spectral=True,
        n_coords_spectral_ordering=5000,
        n_coords_spectral_registration=5000,
        rigid_before_non_rigid_reg=True,
        rigid_reg_max_iterations=100,
        rigid_tolerance=1e-8,
        non_rigid_max_iterations=1000,
        non_rigid_tolerance=1e-8,
        non_rigid_alpha=0.5,
        non_rigid_beta=3.0,
        non_rigid_n_eigens=100,
        include_points_as_features=False,
        get_weighted_spectral_coords=True,
        graph_smoothing_iterations=300,
        feature_smoothing_iterations=40,
        smooth_correspondences=True,
        return_average_final_points=True,
        return_nearest_final_points=True,
        return_transformed_mesh=True,
        projection_smooth_iterations=40,
        feature_weights=None,
        initial_correspondence_type="kd",
        final_correspondence_type="kd",
        list_features_to_calc=["curvature"],
        list_features_to_get_from_mesh=[],
        use_features_as_coords=False,
        use_features_in_graph=False,
        include_features_in_adj_matrix=False,
        G_matrix_p_function="exp",
        norm_node_features_std=True,
        norm_node_features_cap_std=3,
        norm_node_features_0_1=True,
        verbose=False,
        registration=None,
        ctx=None,
    ):
        self.verbose = verbose
        self._ctx = ctx if ctx is not None else _hip.default_context()
        self.registration = registration
        print("Starting Focusr")
        self.n_spectral_features = n_spectral_features
        self.n_extra_spectral = n_extra_spectral
        self.n_total_spectral_features = self.n_spectral_features + self.n_extra_spectral
        self.target_eigenmap_as_reference = target_eigenmap_as_reference

        self.norm_physical_and_spectral = norm_physical_and_spectral
        self.include_points_as_features = include_points_as_features
        self.get_weighted_spectral_coords = get_weighted_spectral_coords
        self.feature_smoothing_iterations = feature_smoothing_iterations
        self.n_coords_spectral_registration = n_coords_spectral_registration
        self.rigid_before_non_rigid_reg = rigid_before_non_rigid_reg
        self.rigid_reg_max_iterations = rigid_reg_max_iterations
        self.rigid_tolerance = rigid_tolerance
        self.non_rigid_max_iterations = non_rigid_max_iterations
        self.non_rigid_tolerance = non_rigid_tolerance
        self.non_rigid_alpha = non_rigid_alpha
        self.non_rigid_beta = non_rigid_beta
        self.non_rigid_n_eigens = non_rigid_n_eigens
        self.initial_correspondence_type = initial_correspondence_type
        self.smooth_correspondences = smooth_correspondences
        self.return_average_final_points = return_average_final_points
        self.return_nearest_final_points = return_nearest_final_points
        self.graph_smoothing_iterations = graph_smoothing_iterations
        self.projection_smooth_iterations = projection_smooth_iterations
        self.final_correspondence_type = final_correspondence_type
        self.return_transformed_mesh = return_transformed_mesh

        for kind in (initial_correspondence_type, final_correspondence_type):
            if kind not in ("kd", "hungarian"):
                raise ValueError("correspondence type must be 'kd' or 'hungarian'")

        print("Starting ICP")
        self._icp_transform = None
        if icp_register_first is True:  # focusr.py:110-131
            if icp_reg_target_to_source is True:
                icp = icp_transform(target=vtk_mesh_source, source=vtk_mesh_target, transform_mode=icp_registration_mode,
                                    ctx=self._ctx)
                vtk_mesh_target = apply_transform(source=vtk_mesh_target, transform=icp)
            else:
                icp = icp_transform(target=vtk_mesh_target, source=vtk_mesh_source, transform_mode=icp_registration_mode,
                                    ctx=self._ctx)
                vtk_mesh_source = apply_transform(source=vtk_mesh_source, transform=icp)
            self._icp_transform = icp

        if (len(list_features_to_calc) and not (use_features_as_coords or use_features_in_graph
                                                or include_features_in_adj_matrix)):
            # The reference computes them (default: VTK curvature, graph.py:84-119) and, with these switches off,
            # never reads them again (graph.py:191,167; focusr.py:524).  VTK-only: skipped when nothing consumes them.
            print("Node features %s are not consumed by any enabled option: not computed" % list(list_features_to_calc))
            list_features_to_calc = []
        graph_kw = dict(
            n_spectral_features=self.n_total_spectral_features,
            n_rand_samples=n_coords_spectral_ordering,
            list_features_to_calc=list_features_to_calc,
            list_features_to_get_from_mesh=list_features_to_get_from_mesh,
            feature_weights=feature_weights,
            include_features_in_G_matrix=use_features_in_graph,
            include_features_in_adj_matrix=include_features_in_adj_matrix,
            G_matrix_p_function=G_matrix_p_function,
            norm_node_features_std=norm_node_features_std,
            norm_node_features_cap_std=norm_node_features_cap_std,
            norm_node_features_0_1=norm_node_features_0_1,
        )
        # focusr.py:134-170 builds target then source; the two spectra are independent, so their
        # Chebyshev recurrences run in lockstep, two graphs per kernel launch.
        print("Starting to build first graph")
        self.graph_target = Graph(vtk_mesh_target, ctx=self._ctx, **graph_kw)
        print("Loaded Mesh 1")
        self.graph_source = Graph(vtk_mesh_source, ctx=self._ctx, **graph_kw)
        print("Loaded Mesh 2")
        compute_spectra([self.graph_target, self.graph_source])
        print("Computed spectrum 1")
        print("Computed spectrum 2")

        self.Q = None
        self.spec_weights = None
        self.spectral_weights = None
        self._coords = {"source": None, "target": None}  # explicit arrays; None: derived on demand from `_coords_recipe`
        self._coords_recipe = None                       # (n_coords, weights or None) of calc_spectral_coords
        self._coords_derived = {"source": False, "target": False}
        self.source_extra_features = None
        self.target_extra_features = None
        self.use_features_as_coords = use_features_as_coords
        self.source_spectral_coords_after_rigid = None
        self.source_spectral_coords_b4_reg = None
        self.rigid_params = None
        self.non_rigid_params = None
        self.smoothed_target_coords = None
        self.source_projected_on_target = None
        self.weighted_avg_transformed_mesh = None
        self.nearest_neighbour_transformed_mesh = None
        self.corresponding_target_idx_for_each_source_pt = None
        self.nearest_neighbor_transformed_points = None
        self.weighted_avg_transformed_points = None
        self.average_mesh = None

    # ------------------------------------------------------------------ spectral coordinates (focusr.py:459-508)
    # The reference stores two (n, k) arrays.  Here they are derived on first access from the graphs' eigenvectors
    # and the weights, so that a pipeline which only needs the correspondences (KNN on the device-resident
    # eigenvectors, `spectral_knn`) never builds them; assigning an array (CPD registration, appended point
    # coordinates) makes it an ordinary attribute again.
    def _get_coords(self, which):
        state = self.__dict__.setdefault("_coords", {"source": None, "target": None})
        derived = self.__dict__.setdefault("_coords_derived", {"source": False, "target": False})
        if state[which] is None and derived[which] and getattr(self, "_coords_recipe", None) is not None:
            n_coords, weights = self._coords_recipe
            vecs = (self.graph_source if which == "source" else self.graph_target).eig_vecs[:, :n_coords]
            state[which] = vecs if weights is None else vecs * weights[None, :]
        return state[which]

    def _set_coords(self, which, value):
        self.__dict__.setdefault("_coords", {"source": None, "target": None})[which] = value
        self.__dict__.setdefault("_coords_derived", {"source": False, "target": False})[which] = False

    source_spectral_coords = property(lambda self: self._get_coords("source"), lambda self, v: self._set_coords("source", v))
    target_spectral_coords = property(lambda self: self._get_coords("target"), lambda self, v: self._set_coords("target", v))

    def _derive_coords(self, n_coords, weights):
        self._coords_recipe = (n_coords, weights)
        self._coords = {"source": None, "target": None}
        self._coords_derived = {"source": True, "target": True}

    # ------------------------------------------------------------------ point sets
    def append_pts_to_spectral_coords(self):
        """focusr.py:271-295."""
        if self.norm_physical_and_spectral is True:
            self.source_spectral_coords = np.concatenate(
                (self.source_spectral_coords, self.graph_source.normed_points), axis=1)
            self.target_spectral_coords = np.concatenate(
                (self.target_spectral_coords, self.graph_target.normed_points), axis=1)
        elif self.norm_physical_and_spectral is False:
            self.source_spectral_coords = np.concatenate(
                (self.source_spectral_coords * self.graph_source.mean_pts_scale_range, self.graph_source.points), axis=1)
            self.target_spectral_coords = np.concatenate(
                (self.target_spectral_coords * self.graph_target.mean_pts_scale_range, self.graph_target.points), axis=1)

    def register_target_to_source(self, reg_type="deformable"):
        """focusr.py:297-334: CPD moves the TARGET cloud onto the source (X=source, Y=target)."""
        X = self.source_spectral_coords[self.graph_source.get_list_rand_idxs(self.n_coords_spectral_registration), :]
        Y = self.target_spectral_coords[self.graph_target.get_list_rand_idxs(self.n_coords_spectral_registration), :]
        if self.registration is not None:
            self.target_spectral_coords = np.asarray(
                self.registration(self.source_spectral_coords, self.target_spectral_coords, reg_type))
            return
        backend, extra = (cycpd, {}) if cycpd is not None else (_native_cpd, {"ctx": self._ctx})
        if reg_type == "deformable":
            reg = backend.deformable_registration(
                **{"X": X, "Y": Y, "num_eig": self.non_rigid_n_eigens, "max_iterations": self.non_rigid_max_iterations,
                   "tolerance": self.non_rigid_tolerance, "alpha": self.non_rigid_alpha, "beta": self.non_rigid_beta,
                   "verbose": self.verbose}, **extra)
            _, self.non_rigid_params = reg.register()
        elif reg_type == "affine":
            reg = backend.affine_registration(
                **{"X": X, "Y": Y, "max_iterations": self.rigid_reg_max_iterations, "tolerance": self.rigid_tolerance},
                **extra)
            _, self.rigid_params = reg.register()
        self.target_spectral_coords = reg.transform_point_cloud(self.target_spectral_coords)

    # ------------------------------------------------------------------ correspondences
    def get_kd_correspondence(self, target_pts, spectral_pts):
        """focusr.py:351-353: nearest target point of every source point, on the GPU."""
        self.corresponding_target_idx_for_each_source_pt = self._ctx.knn1(target_pts, spectral_pts)

    def get_hungarian_correspondence(self, target_pts, spectral_pts):
        """focusr.py:340-349: optimal one-to-one assignment on the dense distance matrix — scipy on the host
        exactly as the reference issues it (O(N^2) memory, O(N^3) time: small meshes only)."""
        from scipy.optimize import linear_sum_assignment
        from scipy.spatial.distance import cdist

        _, target_idx = linear_sum_assignment(cdist(spectral_pts, target_pts))
        self.corresponding_target_idx_for_each_source_pt = target_idx

    def get_initial_correspondences(self):
        """focusr.py:355-366.  While both coordinate sets are still what `calc_spectral_coords` defined (no
        registration moved them), the nearest-neighbour search reads the eigenvectors where the eigensolve left
        them, in HBM (`spectral_knn`); otherwise, and always for "hungarian", the arrays are used."""
        if self.initial_correspondence_type == "hungarian":
            self.get_hungarian_correspondence(self.target_spectral_coords, self.source_spectral_coords)
            return
        derived = self.__dict__.get("_coords_derived", {})
        if derived.get("source") and derived.get("target") and self.__dict__.get("_coords_recipe") is not None:
            n_coords, weights = self._coords_recipe
            idx = spectral_knn(self.graph_target, self.graph_source, n_coords, weights)
            if idx is not None:
                self.corresponding_target_idx_for_each_source_pt = idx
                return
        self.get_kd_correspondence(self.target_spectral_coords, self.source_spectral_coords)

    def get_smoothed_correspondences(self):
        """focusr.py:368-396 (mean filters and the second NN query on the device)."""
        self.smoothed_target_coords = self.graph_target.mean_filter_graph(
            self.graph_target.points, iterations=self.graph_smoothing_iterations)
        if (self.smoothed_target_coords.shape[0] != self.graph_source.n_points) & (
                self.initial_correspondence_type == "hungarian"):  # focusr.py:377-385
            raise Exception(
                "If number vertices between source & target don't match, initial_correspondence_type must\n"
                "be 'kd' and not 'hungarian'. Current type is: {}".format(self.initial_correspondence_type))
        self.source_projected_on_target = self.graph_source.mean_filter_graph(
            np.take(self.smoothed_target_coords, self.corresponding_target_idx_for_each_source_pt, axis=0),
            iterations=self.projection_smooth_iterations)
        if self.final_correspondence_type == "hungarian":  # focusr.py:391-396
            self.get_hungarian_correspondence(self.smoothed_target_coords, self.source_projected_on_target)
        else:
            self.get_kd_correspondence(self.smoothed_target_coords, self.source_projected_on_target)

    def get_weighted_final_node_locations(self, n_closest_pts=3):
        """focusr.py:401-426: every source point goes to the inverse-distance-weighted average of the
        `n_closest_pts` target vertices nearest to its projection (a coincident vertex wins outright).
        The reference loops over points with one `tree.query(k=3)` each; here one device call."""
        idx, d2 = self._ctx.knn(self.smoothed_target_coords, self.source_projected_on_target, n_closest_pts)
        dist = np.sqrt(d2)
        pts = self.graph_target.points
        with np.errstate(divide="ignore", invalid="ignore"):
            w = 1.0 / dist
            num = np.take(pts, idx[:, 0], axis=0) * w[:, 0:1]  # np.take: the fast path for whole-row gathers
            den = w[:, 0:1].copy()
            for j in range(1, idx.shape[1]):
                num = num + np.take(pts, idx[:, j], axis=0) * w[:, j:j + 1]
                den = den + w[:, j:j + 1]
            out = num / den
        coincident = dist == 0.0
        rows = np.nonzero(coincident.any(axis=1))[0]
        if len(rows):  # focusr.py:415-419: first zero-distance neighbour
            first = np.argmax(coincident[rows], axis=1)
            out[rows, :] = pts[idx[rows, first], :]
        self.weighted_avg_transformed_points = out

    def get_nearest_neighbour_final_node_locations(self):
        """focusr.py:428-431."""
        self.nearest_neighbor_transformed_points = np.take(
            self.graph_target.points, self.corresponding_target_idx_for_each_source_pt, axis=0)

    def _source_mesh_with_points(self, new_points):
        mesh = self.graph_source.vtk_mesh
        if isinstance(mesh, PolyMesh):
            return PolyMesh(new_points, mesh.faces.copy(), list(mesh.point_data))
        out = vtk_deep_copy(mesh)
        points = out.GetPoints()
        for i in range(self.graph_source.n_points):
            points.SetPoint(i, new_points[i])
        return out

    def get_source_mesh_transformed_nearest_neighbour(self):
        """focusr.py:615-625."""
        self.nearest_neighbour_transformed_mesh = self._source_mesh_with_points(self.nearest_neighbor_transformed_points)

    def get_source_mesh_transformed_weighted_avg(self):
        """focusr.py:603-613."""
        self.weighted_avg_transformed_mesh = self._source_mesh_with_points(self.weighted_avg_transformed_points)

    # ------------------------------------------------------------------ mesh scalars for visualisation (focusr.py:572-599)
    def set_transformed_source_scalars_to_corresp_target_idx(self):
        for mesh in (getattr(self, "weighted_avg_transformed_mesh", None), getattr(self, "nearest_neighbour_transformed_mesh", None)):
            if mesh is not None:
                set_mesh_scalars(mesh, self.corresponding_target_idx_for_each_source_pt)

    def set_source_scalars_to_corresp_target_idx(self):
        set_mesh_scalars(self.graph_source.vtk_mesh, self.corresponding_target_idx_for_each_source_pt)

    def set_target_scalars_to_corresp_target_idx(self):
        set_mesh_scalars(self.graph_target.vtk_mesh, np.arange(self.graph_target.n_points))

    def set_all_mesh_scalars_to_corresp_target_idx(self):
        self.set_target_scalars_to_corresp_target_idx()
        self.set_source_scalars_to_corresp_target_idx()
        self.set_transformed_source_scalars_to_corresp_target_idx()

    # ------------------------------------------------------------------ viewers (focusr.py:646-795): itkwidgets, not built
    def _no_viewer(self, *args, **kwargs):
        raise ImportError("itkwidgets viewers are not part of the MI355X hot path; the arrays they display are "
                          "`source_spectral_coords`, `target_spectral_coords`, `corresponding_target_idx_for_each_source_pt`, "
                          "`weighted_avg_transformed_mesh` and `nearest_neighbour_transformed_mesh`")

    view_aligned_spectral_coords = view_meshes_colored_by_spectral_correspondences = _no_viewer
    view_aligned_smoothed_spectral_coords = view_meshes = _no_viewer

    def get_average_shape(self, align_type="weighted"):
        """focusr.py:433-453: mean of each source vertex and its image on the target."""
        if align_type == "nearest":
            moved = self.graph_target.points[self.corresponding_target_idx_for_each_source_pt, :]
        else:
            moved = self.weighted_avg_transformed_points
        self.average_mesh = self._source_mesh_with_points((moved + self.graph_source.points) / 2)

    # ------------------------------------------------------------------ spectral weighting (focusr.py:459-508)
    def calc_c_weighting_spectral(self):
        self.spectral_weights = self.Q[: self.n_spectral_features] * np.max(
            (self.graph_source.eig_vals[: self.n_spectral_features],
             self.graph_target.eig_vals[: self.n_spectral_features]), axis=0)
        sigma = np.mean(self.spectral_weights)
        self.spectral_weights = np.exp(-(self.spectral_weights**2) / (2 * sigma**2))
        self.spec_weights = self.spectral_weights  # the reference initialises this name

    def calc_weighted_spectral_coords(self):
        self.calc_c_weighting_spectral()
        self._derive_coords(self.n_spectral_features, self.spectral_weights)  # eig_vecs[:, :ns] * weights, on demand

    def calc_spectral_coords(self):
        if self.get_weighted_spectral_coords is True:
            self.calc_weighted_spectral_coords()
        elif self.get_weighted_spectral_coords is False:
            self._derive_coords(self.n_spectral_features, None)  # eig_vecs[:, :ns], on demand

    # ------------------------------------------------------------------ align_maps (focusr.py:514-568)
    def align_maps(self):
        eig_map_sorter = eigsort(
            graph_target=self.graph_target,
            graph_source=self.graph_source,
            n_features=self.n_total_spectral_features,
            target_as_reference=self.target_eigenmap_as_reference,
        )
        self.Q = eig_map_sorter.sort_eigenmaps()
        self.calc_spectral_coords()

        if self.include_points_as_features is True:
            self.append_pts_to_spectral_coords()

        self.source_spectral_coords_b4_reg = np.copy(self.source_spectral_coords)
        print("Number of features (including spectral) used for registartion: {}".format(
            self.target_spectral_coords.shape[1]))

        if self.rigid_before_non_rigid_reg is True:
            print_header("Rigid Registration Beginning!")
            self.register_target_to_source(reg_type="affine")
            self.source_spectral_coords_after_rigid = np.copy(self.source_spectral_coords)
        print_header("Non-Rigid (Deformable) Registration Beginning")
        self.register_target_to_source("deformable")

        self.get_initial_correspondences()
        print("Number of unique correspondences: {}".format(self._n_unique_correspondences()))
        if self.smooth_correspondences is True:
            self.get_smoothed_correspondences()
            print("Number of unique correspondences after smoothing: {}".format(self._n_unique_correspondences()))
        if self.return_average_final_points is True:
            if self.smoothed_target_coords is None:  # focusr.py:409 needs the smoothed coordinates
                raise RuntimeError("return_average_final_points needs smooth_correspondences=True")
            self.get_weighted_final_node_locations()
        if self.return_nearest_final_points is True:
            self.get_nearest_neighbour_final_node_locations()
        if self.return_transformed_mesh is True:
            if self.return_average_final_points is True:
                self.get_source_mesh_transformed_weighted_avg()
            if self.return_nearest_final_points is True:
                self.get_source_mesh_transformed_nearest_neighbour()

    def _n_unique_correspondences(self):
        """len(np.unique(idx)) (focusr.py:543-554) without the sort: target indices are small non-negative ints."""
        idx = np.asarray(self.corresponding_target_idx_for_each_source_pt)
        return int(np.count_nonzero(np.bincount(idx, minlength=self.graph_target.n_points)))

    @property
    def icp_transform(self):
        """focusr.py:797-807."""
        return self._icp_transform
