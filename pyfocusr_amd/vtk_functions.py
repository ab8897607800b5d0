"""Mesh input for the spectral hot path.

Mirrors the part of the reference's VTK adapter the hot path consumes
(`/root/reference/pyfocusr/vtk_functions.py:5-9` `read_vtk_mesh`) without needing
the `vtk` wheel: a legacy-VTK POLYDATA reader (ASCII and BINARY, file versions <= 4.2 and 5.1) that yields a `PolyMesh`,
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


_VTK_TYPES = {"float": ">f4", "double": ">f8", "int": ">i4", "unsigned_int": ">u4", "long": ">i8", "unsigned_long": ">u8",
              "vtktypeint32": ">i4", "vtktypeint64": ">i8", "short": ">i2", "unsigned_short": ">u2", "char": ">i1",
              "unsigned_char": ">u1", "vtkidtype": ">i8"}


class _LegacyVtkStream(object):
    """Token / raw-block reader over a legacy VTK file (ASCII or BINARY: binary blocks are big-endian)."""

    def __init__(self, data, binary):
        self.data, self.pos, self.binary = data, 0, binary

    def token(self):
        d, n = self.data, len(self.data)
        while self.pos < n and d[self.pos] in b" \t\r\n":
            self.pos += 1
        start = self.pos
        while self.pos < n and d[self.pos] not in b" \t\r\n":
            self.pos += 1
        return d[start:self.pos].decode("ascii", "replace") if self.pos > start else None

    def rest_of_line(self):
        end = self.data.find(b"\n", self.pos)
        end = len(self.data) if end < 0 else end
        line = self.data[self.pos:end]
        self.pos = min(end + 1, len(self.data))
        return line.decode("ascii", "replace")

    def array(self, count, vtk_type):
        dt = _VTK_TYPES.get(vtk_type.lower())
        if dt is None:
            raise NotImplementedError("legacy VTK data type %r" % vtk_type)
        if self.binary:
            if self.pos < len(self.data) and self.data[self.pos:self.pos + 1] == b"\n":
                self.pos += 1  # the newline that ends the header line of the block
            nbytes = count * np.dtype(dt).itemsize
            out = np.frombuffer(self.data, dtype=dt, count=count, offset=self.pos)
            self.pos += nbytes
            return out
        vals = []
        while len(vals) < count:
            t = self.token()
            if t is None:
                raise ValueError("unexpected end of file inside a data block")
            vals.append(t)
        return np.array(vals, dtype=np.float64 if dt[1] == "f" else np.int64)


def read_vtk_mesh(path_to_file):
    """Read a legacy-VTK POLYDATA file into a `PolyMesh`: ASCII or BINARY, file versions up to 4.2 (`POLYGONS n size`
    followed by `v i0 .. iv-1` records, the format of the reference's `data/*.vtk`) and 5.1
    (`OFFSETS` / `CONNECTIVITY` blocks, what current VTK writes).  Same call signature as `vtk_functions.py:5-9`.
    Polygons of mixed size are rejected (all bundled data are triangles)."""
    with open(path_to_file, "rb") as fh:
        data = fh.read()
    st = _LegacyVtkStream(data, False)
    if not st.rest_of_line().startswith("# vtk DataFile"):
        raise ValueError("not a legacy VTK file: %s" % path_to_file)
    st.rest_of_line()  # title
    fmt = st.rest_of_line().strip().upper()
    if fmt not in ("ASCII", "BINARY"):
        raise NotImplementedError("legacy VTK format %r" % fmt)
    st.binary = fmt == "BINARY"
    if (st.token() or "").upper() != "DATASET" or (st.token() or "").upper() != "POLYDATA":
        raise NotImplementedError("only DATASET POLYDATA is supported")

    points = faces = None
    point_data, n_points, in_point_data = [], 0, False
    while True:
        key = st.token()
        if key is None:
            break
        key = key.upper()
        if key == "POINTS":
            n_points = int(st.token())
            points = np.asarray(st.array(3 * n_points, st.token()), dtype=np.float64).reshape(n_points, 3)
        elif key in ("POLYGONS", "VERTICES", "LINES", "TRIANGLE_STRIPS"):
            a, b = int(st.token()), int(st.token())
            nxt_pos = st.pos
            nxt = st.token()
            if nxt is not None and nxt.upper() == "OFFSETS":  # version 5.1: a offsets (cells + 1), b connectivity entries
                offs = np.asarray(st.array(a, st.token()), dtype=np.int64)
                if (st.token() or "").upper() != "CONNECTIVITY":
                    raise ValueError("OFFSETS without CONNECTIVITY")
                conn = np.asarray(st.array(b, st.token()), dtype=np.int64)
                if key == "POLYGONS":
                    widths = np.diff(offs)
                    if len(widths) == 0:
                        faces = np.zeros((0, 3), dtype=np.int32)
                    elif np.any(widths != widths[0]):
                        raise NotImplementedError("mixed polygon sizes are not supported")
                    else:
                        faces = conn.reshape(len(widths), int(widths[0])).astype(np.int32)
            else:  # versions <= 4.2: a cells, b integers in all
                st.pos = nxt_pos
                flat = np.asarray(st.array(b, "int"), dtype=np.int64)
                if key == "POLYGONS":
                    if a == 0:
                        faces = np.zeros((0, 3), dtype=np.int32)
                    else:
                        v = int(flat[0])
                        if b != a * (v + 1) or np.any(flat.reshape(a, v + 1)[:, 0] != v):
                            raise NotImplementedError("mixed polygon sizes are not supported")
                        faces = flat.reshape(a, v + 1)[:, 1:].astype(np.int32)
        elif key == "POINT_DATA":
            st.token()
            in_point_data = True
        elif key == "CELL_DATA":
            st.token()
            in_point_data = False
        elif key == "SCALARS":
            name, vtype = st.token(), st.token()
            line = st.rest_of_line().split()
            ncomp = int(line[0]) if line and line[0].isdigit() else 1
            if (st.token() or "").upper() != "LOOKUP_TABLE":
                raise ValueError("SCALARS without LOOKUP_TABLE")
            st.token()
            count = (n_points if in_point_data else 0) * ncomp
            if not in_point_data:
                raise NotImplementedError("cell SCALARS are not supported")
            vals = np.asarray(st.array(count, vtype), dtype=np.float64)
            point_data.append((name, vals if ncomp == 1 else vals.reshape(n_points, ncomp)))
        # anything else (METADATA, NORMALS, FIELD, ...): tokens are skipped one by one; the hot path needs none of it
    if points is None or faces is None:
        raise ValueError("file has no POINTS/POLYGONS section: %s" % path_to_file)
    return PolyMesh(points, faces, point_data)


def write_vtk_mesh(mesh, path_to_file, title="pyfocusr_amd"):
    """Write a mesh (a `PolyMesh`, or anything `mesh_arrays` accepts) as an ASCII legacy-VTK 4.2 POLYDATA file with its
    scalar point-data arrays — e.g. the transformed source meshes of `Focusr` with the correspondence indices set by
    `set_all_mesh_scalars_to_corresp_target_idx`.  Coordinates are written with 17 significant digits (round trip exact)."""
    pts, faces = mesh_arrays(mesh)
    with open(path_to_file, "w") as fh:
        fh.write("# vtk DataFile Version 4.2\n%s\nASCII\nDATASET POLYDATA\nPOINTS %d double\n" % (title, len(pts)))
        for p in pts:
            fh.write("%.17g %.17g %.17g\n" % (p[0], p[1], p[2]))
        fh.write("POLYGONS %d %d\n" % (len(faces), len(faces) * (faces.shape[1] + 1)))
        for f in faces:
            fh.write("%d %s\n" % (len(f), " ".join(str(int(v)) for v in f)))
        arrays = [(n, np.asarray(v)) for n, v in getattr(mesh, "point_data", []) if np.asarray(v).ndim == 1 and len(v) == len(pts)]
        if arrays:
            fh.write("POINT_DATA %d\n" % len(pts))
            for name, vals in arrays:
                fh.write("SCALARS %s double\nLOOKUP_TABLE default\n" % str(name).replace(" ", "_"))
                fh.write("\n".join("%.17g" % float(v) for v in vals) + "\n")


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
