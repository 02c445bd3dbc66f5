"""Gmsh 2.x ASCII reader for the subset the reference's models read through libMesh's GmshIO
(`src/pihna.C:44`, `src/solid.C`; the writer side of the same subset is `src/process_mesh.C:21-83`):
nodes, first-order volume elements (TET4 = type 4, HEX8 = type 5) whose first tag is the subdomain id,
and lower-dimensional elements (TRI3 = 2, QUAD4 = 3) whose first tag is a boundary id that libMesh
attaches to the matching element side.  SURVEY §8(f) rank 4 ("on-disk formats")."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_NNODES = {2: 3, 3: 4, 4: 4, 5: 8}
_SIDES = {4: [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)],
          8: [(0, 3, 2, 1), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7)]}


@dataclass
class GmshMesh:
    xyz: np.ndarray            # [n_node][3]
    elem_type: int             # 4 (TET4) or 8 (HEX8)
    conn: np.ndarray           # [n_elem][elem_type] uint32, 0-based, Gmsh == libMesh node order
    subdomain: np.ndarray      # [n_elem] first tag of the volume elements
    face_nodes: np.ndarray     # [n_face][3 or 4] boundary elements
    face_tag: np.ndarray       # [n_face] first tag (boundary id)

    def sides_with_boundary_id(self, bid):
        """(elem, libMesh side number) of every boundary face tagged `bid`."""
        nen = self.elem_type
        lut = {}
        for s, loc in enumerate(_SIDES[nen]):
            keys = np.sort(self.conn[:, loc], axis=1)
            for e, k in enumerate(map(tuple, keys)):
                lut[k] = (e, s)
        es, ss = [], []
        for f in np.nonzero(self.face_tag == bid)[0]:
            k = tuple(sorted(self.face_nodes[f].tolist()))
            if k not in lut:
                raise ValueError(f"boundary face {f} matches no element side")
            e, s = lut[k]
            es.append(e)
            ss.append(s)
        return np.asarray(es, dtype=np.int64), np.asarray(ss, dtype=np.int32)


def read_msh2(path) -> GmshMesh:
    with open(path) as fh:
        lines = fh.read().split("\n")
    i = 0
    ids, xyz, vol, faces = {}, None, {4: [], 5: []}, {2: [], 3: []}
    while i < len(lines):
        ln = lines[i].strip()
        if ln == "$MeshFormat":
            ver = float(lines[i + 1].split()[0])
            if not 2.0 <= ver < 3.0:
                raise ValueError(f"unsupported Gmsh format {ver}")
            i += 3
        elif ln == "$Nodes":
            n = int(lines[i + 1])
            arr = np.array([lines[i + 2 + k].split() for k in range(n)], dtype=np.float64)
            ids = {int(t): k for k, t in enumerate(arr[:, 0])}
            xyz = np.ascontiguousarray(arr[:, 1:4])
            i += n + 3
        elif ln == "$Elements":
            n = int(lines[i + 1])
            for k in range(n):
                t = lines[i + 2 + k].split()
                et, ntags = int(t[1]), int(t[2])
                if et not in _NNODES:
                    continue
                tag = int(t[3]) if ntags > 0 else 0
                nodes = [ids[int(x)] for x in t[3 + ntags:3 + ntags + _NNODES[et]]]
                (vol if et in vol else faces)[et].append((tag, nodes))
            i += n + 3
        else:
            i += 1
    if xyz is None:
        raise ValueError("no $Nodes section")
    if vol[4] and vol[5]:
        raise ValueError("mixed TET4/HEX8 meshes are not supported")
    et = 4 if vol[4] else 5
    if not vol[et]:
        raise ValueError("no TET4 or HEX8 elements")
    nen = _NNODES[et]
    fl = faces[2] + faces[3]
    fn = max((len(f[1]) for f in fl), default=3)
    return GmshMesh(xyz=xyz, elem_type=nen,
                    conn=np.array([v[1] for v in vol[et]], dtype=np.uint32),
                    subdomain=np.array([v[0] for v in vol[et]], dtype=np.int32),
                    face_nodes=np.array([f[1] for f in fl if len(f[1]) == fn], dtype=np.int64).reshape(-1, fn),
                    face_tag=np.array([f[0] for f in fl if len(f[1]) == fn], dtype=np.int32))
