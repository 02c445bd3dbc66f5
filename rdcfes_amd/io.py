"""On-disk formats of the reference, host side: the whitespace-separated nodal / elemental
initial-field files (`fin >> n_ >> c_ >> h_ >> v_ >> a_` per node, src/pihna.C:287-292; `HU RT` per element,
:251-256), the ASCII VTU files of `Paraview_IO::write_nodal_data` (src/paraview.h:30-150) with the same arrays and
names, and the PVD collection (src/paraview.h:160-200)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

SMALLEST_NUMBER = 1.0e-30   # values below are written as 0, src/paraview.h:91 (threshold of src/utils.h)
_VTK_TYPE = {4: 10, 8: 12}  # VTK_TETRA, VTK_HEXAHEDRON; Gmsh/libMesh/VTK node orders coincide for both


def read_field_dat(path, n_rows, n_cols) -> np.ndarray:
    """n_rows x n_cols numbers in stream order, as the reference's `fin >>` loops read them."""
    a = np.array(Path(path).read_text().split(), dtype=np.float64)
    if a.size < n_rows * n_cols:
        raise ValueError(f"{path}: {a.size} numbers, need {n_rows * n_cols}")
    return a[:n_rows * n_cols].reshape(n_rows, n_cols)


def write_field_dat(path, values):
    np.savetxt(path, np.atleast_2d(values), fmt="%.17g")


def _arr(name, typ, data, ncomp=1):
    body = " " + " ".join(("%d" % v) if typ.startswith("Int") else repr(float(v)) for v in np.asarray(data).ravel())
    return (f'        <DataArray type="{typ}" Name="{name}" NumberOfComponents="{ncomp}" format="ascii">\n'
            f"{body}\n        </DataArray>\n")


def write_vtu(path, elem_type, conn, xyz, names, nodal, region_id=None, processor_id=None):
    """One ASCII .vtu with the arrays the reference writes: position; PointData node_ID (1-based) + one scalar per
    variable name; CellData element_ID (1-based), region_ID, processor_ID; Cells connectivity/offsets/types.
    Nodes not used by any element are left out, as upstream (:38-55)."""
    conn = np.asarray(conn, dtype=np.int64)
    xyz = np.asarray(xyz, dtype=np.float64)
    nodal = np.asarray(nodal, dtype=np.float64).reshape(xyz.shape[0], len(names))
    used = np.zeros(xyz.shape[0], bool)
    used[conn.ravel()] = True
    vtk_id = np.full(xyz.shape[0], -1, np.int64)
    vtk_id[used] = np.arange(int(used.sum()))
    ne = conn.shape[0]
    region_id = np.zeros(ne, np.int64) if region_id is None else np.asarray(region_id)
    processor_id = np.zeros(ne, np.int64) if processor_id is None else np.asarray(processor_id)
    vals = np.where(np.abs(nodal) <= SMALLEST_NUMBER, 0.0, nodal)
    out = ['<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">\n', "  <UnstructuredGrid>\n",
           f'    <Piece  NumberOfPoints="{int(used.sum())}" NumberOfCells="{ne}">\n', "      <Points>\n",
           _arr("position", "Float64", xyz[used], 3), "      </Points>\n", "      <PointData>\n",
           _arr("node_ID", "Int32", np.nonzero(used)[0] + 1)]
    out += [_arr(nm, "Float64", vals[used, j]) for j, nm in enumerate(names)]
    out += ["      </PointData>\n", "      <CellData>\n", _arr("element_ID", "Int32", np.arange(ne) + 1),
            _arr("region_ID", "Int32", region_id), _arr("processor_ID", "Int32", processor_id), "      </CellData>\n",
            "      <Cells>\n", _arr("connectivity", "Int32", vtk_id[conn]),
            _arr("offsets", "Int32", elem_type * (np.arange(ne) + 1)), _arr("types", "Int32", np.full(ne, _VTK_TYPE[elem_type])),
            "      </Cells>\n", "    </Piece>\n", "  </UnstructuredGrid>\n", "</VTKFile>\n"]
    Path(path).write_text("".join(out))


class PvdCollection:
    """`<stem>.pvd` listing one .vtu per output time (open_pvd / update_pvd / close_pvd of src/paraview.h)."""

    def __init__(self, stem):
        self.stem = Path(stem)
        self.entries = []

    def add(self, time, elem_type, conn, xyz, names, nodal, **kw):
        fn = self.stem.parent / f"{self.stem.name}_{len(self.entries):06d}.vtu"
        write_vtu(fn, elem_type, conn, xyz, names, nodal, **kw)
        self.entries.append((float(time), fn.name))
        return fn

    def close(self):
        body = "".join(f'    <DataSet timestep="{t!r}" group="" part="0" file="{f}"/>\n' for t, f in self.entries)
        self.stem.with_suffix(".pvd").write_text('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1" '
                                                'byte_order="LittleEndian">\n  <Collection>\n' + body +
                                                "  </Collection>\n</VTKFile>\n")
