"""Mesh input for the spectral hot path.

Mirrors the part of the reference's VTK adapter the hot path consumes
(`/root/reference/pyfocusr/vtk_functions.py:5-9` `read_vtk_mesh`) without needing
the `vtk` wheel: an ASCII legacy-VTK POLYDATA reader that yields a `PolyMesh`,
a light object that

* exposes `points` (n,3) f64 and `faces` (F,v) i32 arrays directly (fast path
  used by `Graph`), and
* duck-types the slice of the vtkPolyData protocol the reference walks
  (`graph.py:58-62,155-164`): `GetNumberOfPoints/GetPoint/GetNumberOfCells/
  GetCell(c).GetNumberOfEdges()/GetEdge(e).GetPointId(0|1)` with VTK's polygon
  edge order (0,1),(1,2),...,(v-1,0).

ICP (`vtk_functions.py:12-37`): real `vtkPolyData` inputs go to
VTK's own `vtkIterativeClosestPointTransform` when the `vtk` wheel is importable;
everything else (always the case in the build image) runs the same iteration with
the closest-point search on the MI355X (`pyfocusr_amd/icp.py`).  Curvature
(`vtk_functions.py:40-74`) stays VTK-only.
"""
import numpy as np

try:  # pragma: no cover - vtk is not installed in the build image
    import vtk as _vtk
except Exception:  # noqa: BLE001
    _vtk = None


class _Edge(object):
    __slots__ = ("_a", "_b")

    def __init__(self, a, b):
        self._a = a
        self._b = b

    def GetPointId(self, i):
        return self._a if i == 0 else self._b


class _Cell(object):
    __slots__ = ("_ids",)

    def __init__(self, ids):
        self._ids = ids

    def GetNumberOfEdges(self):
        return len(self._ids)

    def GetNumberOfPoints(self):
        return len(self._ids)

    def GetPointId(self, i):
        return int(self._ids[i])

    def GetEdge(self, e):
        n = len(self._ids)
        return _Edge(int(self._ids[e]), int(self._ids[(e + 1) % n]))


class _PointData(object):
    def __init__(self, arrays):
        self._arrays = arrays  # list of (name, ndarray)

    def GetNumberOfArrays(self):
        return len(self._arrays)

    def GetArray(self, idx):
        name, values = self._arrays[idx]
        return _NamedArray(name, values)


class _NamedArray(object):
    def __init__(self, name, values):
        self._name = name
        self.values = values

    def GetName(self):
        return self._name


class PolyMesh(object):
    """Triangle/polygon surface mesh: `points` (n,3) float64, `faces` (F,v) int32."""

    def __init__(self, points, faces, point_data=None):
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        faces = np.asarray(faces)
        if faces.ndim != 2:
            raise ValueError("faces must be (F, verts_per_face)")
        self.faces = np.ascontiguousarray(faces, dtype=np.int32)
        if self.faces.size and (self.faces.min() < 0 or self.faces.max() >= len(self.points)):
            raise ValueError("face index out of range")
        self.point_data = list(point_data or [])

    # --- vtkPolyData protocol subset (graph.py:58-62, 89-104, 155-164) ---
    def GetNumberOfPoints(self):
        return int(self.points.shape[0])

    def GetPoint(self, i):
        p = self.points[i]
        return (float(p[0]), float(p[1]), float(p[2]))

    def GetNumberOfCells(self):
        return int(self.faces.shape[0])

    def GetCell(self, c):
        return _Cell(self.faces[c])

    def GetPointData(self):
        return _PointData(self.point_data)


def _tokens(path):
    with open(path, "r") as fh:
        for line in fh:
            for tok in line.split():
                yield tok


def read_vtk_mesh(path_to_file):
    """Read an ASCII legacy-VTK POLYDATA file (the format of the reference's
    `data/*.vtk`) into a `PolyMesh`.

    Same call signature as `vtk_functions.py:5-9`.  Polygons of mixed size are
    rejected (all bundled data are triangles)."""
    with open(path_to_file, "r") as fh:
        header = [fh.readline() for _ in range(4)]
        if not header[0].startswith("# vtk DataFile"):
            raise ValueError("not a legacy VTK file: %s" % path_to_file)
        if header[2].strip().upper() != "ASCII":
            raise NotImplementedError("only ASCII legacy VTK files are supported")
        if header[3].split()[:2] != ["DATASET", "POLYDATA"]:
            raise NotImplementedError("only DATASET POLYDATA is supported")
        rest = fh.read().split()

    pos = 0
    points = None
    faces = None
    point_data = []
    n_points = 0
    while pos < len(rest):
        key = rest[pos].upper()
        if key == "POINTS":
            n_points = int(rest[pos + 1])
            pos += 3
            points = np.array(rest[pos : pos + 3 * n_points], dtype=np.float64).reshape(n_points, 3)
            pos += 3 * n_points
        elif key == "POLYGONS":
            n_cells = int(rest[pos + 1])
            total = int(rest[pos + 2])
            pos += 3
            flat = np.array(rest[pos : pos + total], dtype=np.int64)
            pos += total
            if n_cells == 0:
                faces = np.zeros((0, 3), dtype=np.int32)
            else:
                v = int(flat[0])
                if total != n_cells * (v + 1) or np.any(flat.reshape(n_cells, v + 1)[:, 0] != v):
                    raise NotImplementedError("mixed polygon sizes are not supported")
                faces = flat.reshape(n_cells, v + 1)[:, 1:].astype(np.int32)
        elif key == "POINT_DATA":
            pos += 2
        elif key == "SCALARS":
            name = rest[pos + 1]
            ncomp = 1
            pos += 3
            if pos < len(rest) and rest[pos].isdigit():
                ncomp = int(rest[pos])
                pos += 1
            if rest[pos].upper() == "LOOKUP_TABLE":
                pos += 2
            vals = np.array(rest[pos : pos + n_points * ncomp], dtype=np.float64)
            pos += n_points * ncomp
            point_data.append((name, vals if ncomp == 1 else vals.reshape(n_points, ncomp)))
        else:
            # unknown section (VERTICES/LINES/CELL_DATA/...): not needed by the hot path.
            pos += 1
    if points is None or faces is None:
        raise ValueError("file has no POINTS/POLYGONS section: %s" % path_to_file)
    return PolyMesh(points, faces, point_data)


def mesh_arrays(mesh):
    """(points (n,3) f64, faces (F,v) i32) of any mesh object the reference
    accepts.  Fast paths: `PolyMesh`, objects with `.points/.faces`; fallback: the
    generic cell/edge walk of `graph.py:58-62,155-164`."""
    if hasattr(mesh, "points") and hasattr(mesh, "faces"):
        return (
            np.ascontiguousarray(mesh.points, dtype=np.float64).reshape(-1, 3),
            np.ascontiguousarray(mesh.faces, dtype=np.int32),
        )
    if _vtk is not None and hasattr(mesh, "GetPolys") and hasattr(mesh, "GetPoints"):
        try:  # a real vtkPolyData: read the arrays in bulk instead of walking cells in Python
            from vtk.util.numpy_support import vtk_to_numpy

            pts = np.ascontiguousarray(vtk_to_numpy(mesh.GetPoints().GetData()), dtype=np.float64).reshape(-1, 3)
            polys = mesh.GetPolys()
            conn = vtk_to_numpy(polys.GetConnectivityArray())
            offs = vtk_to_numpy(polys.GetOffsetsArray())
            widths = np.diff(offs)
            if (len(widths) and np.all(widths == widths[0]) and polys.GetNumberOfCells() == mesh.GetNumberOfCells()):
                return pts, np.ascontiguousarray(conn.reshape(-1, int(widths[0])), dtype=np.int32)
        except Exception:  # noqa: BLE001 - older VTK without offsets/connectivity arrays: generic walk below
            pass
    n = mesh.GetNumberOfPoints()
    pts = np.zeros((n, 3))
    for i in range(n):
        pts[i, :] = mesh.GetPoint(i)
    cells = []
    for c in range(mesh.GetNumberOfCells()):
        cell = mesh.GetCell(c)
        ne = cell.GetNumberOfEdges()
        cells.append([int(cell.GetEdge(e).GetPointId(0)) for e in range(ne)])
    widths = {len(c) for c in cells}
    if len(widths) > 1:
        raise NotImplementedError("mixed polygon sizes are not supported")
    v = widths.pop() if widths else 3
    return pts, np.asarray(cells, dtype=np.int32).reshape(-1, v)


def set_mesh_scalars(mesh, values, name="scalars"):
    """`mesh.GetPointData().SetScalars(numpy_to_vtk(values))` (focusr.py:576-599) for either kind of mesh: a real
    vtkPolyData gets VTK scalars; a `PolyMesh` gets / replaces the point-data array `name` (also `mesh.scalars`)."""
    values = np.asarray(values)
    if _is_vtk_polydata(mesh):
        from vtk.util.numpy_support import numpy_to_vtk

        mesh.GetPointData().SetScalars(numpy_to_vtk(values))
        return
    if len(values) != mesh.GetNumberOfPoints():
        raise ValueError("one scalar per point expected")
    mesh.point_data = [(n, v) for n, v in getattr(mesh, "point_data", []) if n != name] + [(name, values.copy())]
    mesh.scalars = mesh.point_data[-1][1]


def _need_vtk(what):
    if _vtk is None:
        raise NotImplementedError(
            "%s needs the `vtk` package (VTK C++ plumbing outside the MI355X hot path, "
            "SURVEY.md §8 f3)" % what
        )


def _is_vtk_polydata(mesh):
    return _vtk is not None and isinstance(mesh, _vtk.vtkPolyData)


def icp_transform(target, source, numberOfIterations=100, number_landmarks=1000, transform_mode="rigid", ctx=None):
    """`vtk_functions.py:12-29`.  Same arguments; returns an object with `GetMatrix()`."""
    if transform_mode not in ("rigid", "similarity"):
        raise ValueError("Error invalid transform mode")
    if not (_is_vtk_polydata(target) and _is_vtk_polydata(source)):
        from . import icp as _icp

        t_pts, t_faces = mesh_arrays(target)
        s_pts, _ = mesh_arrays(source)
        return _icp.icp_transform(t_pts, t_faces, s_pts, numberOfIterations=numberOfIterations,
                                  number_landmarks=number_landmarks, transform_mode=transform_mode, ctx=ctx)
    icp = _vtk.vtkIterativeClosestPointTransform()
    if transform_mode == "rigid":
        icp.GetLandmarkTransform().SetModeToRigidBody()
    elif transform_mode == "similarity":
        icp.GetLandmarkTransform().SetModeToSimilarity()
    else:
        raise ValueError("Error invalid transform mode")
    icp.SetTarget(target)
    icp.SetSource(source)
    icp.SetMaximumNumberOfIterations(numberOfIterations)
    icp.StartByMatchingCentroidsOn()
    icp.Modified()
    icp.Update()
    icp.SetMaximumNumberOfLandmarks(number_landmarks)
    return icp


def apply_transform(source, transform):
    """`vtk_functions.py:32-37`: a new mesh with the transform applied to the points."""
    if hasattr(transform, "transform_points"):  # pyfocusr_amd.icp.IcpTransform
        if _is_vtk_polydata(source):
            t = _vtk.vtkTransform()
            t.SetMatrix([transform.GetMatrix().GetElement(i, j) for i in range(4) for j in range(4)])
            transform = t
        else:
            pts, faces = mesh_arrays(source)
            return PolyMesh(transform.transform_points(pts), faces, list(getattr(source, "point_data", [])))
    _need_vtk("apply_transform")
    f = _vtk.vtkTransformPolyDataFilter()
    f.SetInputData(source)
    f.SetTransform(transform)
    f.Update()
    return f.GetOutput()


def vtk_deep_copy(mesh):
    """`vtk_functions.py:77-81`."""
    if isinstance(mesh, PolyMesh):
        return PolyMesh(mesh.points.copy(), mesh.faces.copy(), list(mesh.point_data))
    _need_vtk("vtk_deep_copy")
    new_mesh = _vtk.vtkPolyData()
    new_mesh.DeepCopy(mesh)
    return new_mesh
