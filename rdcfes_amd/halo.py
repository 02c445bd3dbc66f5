"""Halo exchange of ghost-node DoFs (`system.update()` of the reference, src/pihna.C:801;
SURVEY §2.3, §8e) with torch.distributed: backend "nccl" is RCCL over xGMI on MI355X, "gloo" on CPU
for the tests.  One grouped point-to-point round per step (each GPU talks to its few face
neighbours over its direct xGMI links; the messages are O(0.1-2 MB), latency-bound) -- no
all-reduce, no ring.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class HaloExchange:
    def __init__(self, lp, nvar: int, device, group=None):
        self.group = group
        self.nvar = nvar
        self.rank = lp.rank
        self.peers = sorted(set(lp.send_ids) | set(lp.recv_ids))
        dev = torch.device(device)
        self.send_idx = {q: torch.as_tensor(lp.send_ids[q], dtype=torch.long, device=dev) for q in lp.send_ids}
        self.recv_idx = {q: torch.as_tensor(lp.recv_ids[q], dtype=torch.long, device=dev) for q in lp.recv_ids}
        self.send_buf = {q: torch.empty((i.numel(), nvar), dtype=torch.float64, device=dev) for q, i in self.send_idx.items()}
        self.recv_buf = {q: torch.empty((i.numel(), nvar), dtype=torch.float64, device=dev) for q, i in self.recv_idx.items()}
        self.bytes_per_step = 8 * nvar * sum(i.numel() for i in self.send_idx.values())

    def exchange(self, u: torch.Tensor):
        """u: [n_node_local][nvar]; owned rows are read, ghost rows are overwritten in place."""
        if not self.peers:
            return
        ops = []
        for q in self.peers:
            if q in self.recv_idx:
                ops.append(dist.P2POp(dist.irecv, self.recv_buf[q], q, group=self.group))
        for q in self.peers:
            if q in self.send_idx:
                torch.index_select(u, 0, self.send_idx[q], out=self.send_buf[q])
                ops.append(dist.P2POp(dist.isend, self.send_buf[q], q, group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for q, idx in self.recv_idx.items():
            u.index_copy_(0, idx, self.recv_buf[q])
