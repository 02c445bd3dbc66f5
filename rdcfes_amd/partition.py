"""Element-based domain decomposition of the assembly path.

The reference partitions its ReplicatedMesh with libMesh's default partitioner (METIS) and every
MPI rank loops over `active_local_element_ptr_range()` (src/pihna.C:383); rows shared with another
rank are completed through PETSc's off-process stash.  Here every GPU instead stores its owned
elements plus ONE layer of ghost elements (all elements touching an owned node), so it assembles
the complete rows of its owned nodes by itself: no matrix-entry exchange at all.  The only
per-step communication left is the halo update of ghost-node DoFs (`system.update()`,
src/pihna.C:801), done by halo.py over RCCL.

METIS is not available in this image; the partitioner here is recursive coordinate bisection on
element centroids (balanced to +-1 element).  Node ownership follows libMesh: the lowest-ranked
partition touching a node owns it.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def partition_rcb(centroids: np.ndarray, nparts: int) -> np.ndarray:
    """part id per element, recursive coordinate bisection along the longest axis."""
    n = centroids.shape[0]
    part = np.zeros(n, dtype=np.int32)
    todo = [(np.arange(n), 0, nparts)]
    while todo:
        idx, base, k = todo.pop()
        if k == 1:
            part[idx] = base
            continue
        kl = k // 2
        c = centroids[idx]
        ax = int(np.argmax(c.max(axis=0) - c.min(axis=0))) if idx.size else 0
        nl = (idx.size * kl) // k
        # O(n) selection instead of a full sort; ties are broken by element id so that every rank
        # computes the same partition
        key = c[:, ax]
        if 0 < nl < idx.size:
            order = np.argpartition(key, nl - 1)
            left, right = order[:nl], order[nl:]
            # make the split deterministic w.r.t. equal keys: elements equal to the pivot go by index
            pivot = key[left].max()
            eq = np.nonzero(key == pivot)[0]
            if eq.size > 1:
                n_left_strict = int((key < pivot).sum())
                take = np.sort(eq)[: nl - n_left_strict]
                mask = key < pivot
                mask[take] = True
                left, right = np.nonzero(mask)[0], np.nonzero(~mask)[0]
        else:
            left, right = np.arange(nl), np.arange(nl, idx.size)
        todo.append((idx[left], base, kl))
        todo.append((idx[right], base + kl, k - kl))
    return part


def node_owners(conn: np.ndarray, elem_part: np.ndarray, n_node: int, nparts: int) -> np.ndarray:
    owner = np.full(n_node, nparts, dtype=np.int32)
    np.minimum.at(owner, conn.ravel(), np.repeat(elem_part.astype(np.int32), conn.shape[1]))
    return owner


@dataclass
class LocalPartition:
    rank: int
    nparts: int
    conn: np.ndarray          # [n_elem_local][nen] LOCAL node ids (owned first, then ghosts)
    xyz: np.ndarray           # [n_node_local][3]
    n_owned: int
    node_global: np.ndarray   # local node id -> global node id
    elem_global: np.ndarray   # local element id -> global element id
    n_elem_owned: int         # elements of this partition in the global partition (metric accounting)
    n_interior: int = 0       # leading owned nodes whose elements contain no ghost node (assembled before the halo lands)
    # halo plan: for every peer q, the local ids to send (owned here, ghost on q) and to receive
    send_ids: dict = field(default_factory=dict)
    recv_ids: dict = field(default_factory=dict)


def build_local(conn: np.ndarray, xyz: np.ndarray, elem_part: np.ndarray, rank: int, nparts: int,
                owner: np.ndarray | None = None) -> LocalPartition:
    """Local mesh of `rank`: owned nodes, owned + ghost-layer elements, halo send/recv lists.

    Every rank holds the global mesh in this harness, so the peers' needs are computed locally
    instead of being negotiated; the lists on both sides are ordered by global node id.
    """
    n_node = xyz.shape[0]
    if owner is None:
        owner = node_owners(conn, elem_part, n_node, nparts)
    owned_mask = owner == rank

    def local_elems(r):
        return np.nonzero((owner[conn] == r).any(axis=1))[0]

    elems = local_elems(rank)
    mine = elem_part[elems] == rank
    elems = np.concatenate([elems[mine], elems[~mine]])  # partition proper first, ghost layer after
    touched = np.unique(conn[elems])
    owned = touched[owned_mask[touched]]
    # owned nodes that no local element touches cannot exist (owner = a touching partition)
    assert owned.size == int(owned_mask.sum())
    ghosts = touched[~owned_mask[touched]]
    ghosts = ghosts[np.lexsort((ghosts, owner[ghosts]))]
    # owned nodes none of whose elements contains a ghost node come first: their rows can be assembled while the halo
    # exchange of the step is still in flight (bench.py overlaps the two); the order inside both groups is kept
    ce = conn[elems]
    near_ghost = np.zeros(n_node, dtype=bool)
    near_ghost[np.unique(ce[(~owned_mask[ce]).any(axis=1)])] = True
    interior = owned[~near_ghost[owned]]
    owned = np.concatenate([interior, owned[near_ghost[owned]]])
    node_global = np.concatenate([owned, ghosts])
    g2l = np.full(n_node, -1, dtype=np.int64)
    g2l[node_global] = np.arange(node_global.size)
    lp = LocalPartition(rank=rank, nparts=nparts, conn=g2l[conn[elems]].astype(np.uint32), xyz=xyz[node_global],
                        n_owned=owned.size, node_global=node_global, elem_global=elems,
                        n_elem_owned=int((elem_part == rank).sum()), n_interior=int(interior.size))
    for q in range(nparts):
        if q == rank:
            continue
        r = ghosts[owner[ghosts] == q]  # already sorted by global id
        if r.size:
            lp.recv_ids[q] = g2l[r]
        # what q needs from me: nodes I own inside q's local elements
        tq = np.unique(conn[local_elems(q)])
        s = tq[owner[tq] == rank]
        if s.size:
            lp.send_ids[q] = g2l[s]
    return lp
