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
        # gloo cannot move device tensors: stage through host buffers (test rigs with one GPU only;
        # the production backend is "nccl" = RCCL, which sends the device buffers directly)
        self.host_staged = dev.type == "cuda" and dist.is_initialized() and dist.get_backend(group) == "gloo"
        if self.host_staged:
            self.send_host = {q: torch.empty_like(b, device="cpu").pin_memory() for q, b in self.send_buf.items()}
            self.recv_host = {q: torch.empty_like(b, device="cpu").pin_memory() for q, b in self.recv_buf.items()}

    def exchange(self, u: torch.Tensor):
        """u: [n_node_local][nvar]; owned rows are read, ghost rows are overwritten in place."""
        if not self.peers:
            return
        ops = []
        rbuf = self.recv_host if self.host_staged else self.recv_buf
        sbuf = self.send_host if self.host_staged else self.send_buf
        for q in self.peers:
            if q in self.recv_idx:
                ops.append(dist.P2POp(dist.irecv, rbuf[q], q, group=self.group))
        for q in self.peers:
            if q in self.send_idx:
                torch.index_select(u, 0, self.send_idx[q], out=self.send_buf[q])
                if self.host_staged:
                    self.send_host[q].copy_(self.send_buf[q], non_blocking=True)
        if self.host_staged:
            torch.cuda.current_stream().synchronize()
        for q in self.peers:
            if q in self.send_idx:
                ops.append(dist.P2POp(dist.isend, sbuf[q], q, group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for q, idx in self.recv_idx.items():
            if self.host_staged:
                self.recv_buf[q].copy_(self.recv_host[q], non_blocking=True)
            u.index_copy_(0, idx, self.recv_buf[q])
