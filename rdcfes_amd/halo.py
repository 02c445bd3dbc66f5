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
    """All peers share ONE send and ONE receive buffer (per-peer slices of them are what is sent): a step costs one
    gather kernel, one grouped send/recv and one scatter kernel whatever the number of neighbours -- at 8 GPUs
    the assembly kernel is ~0.4 ms, so per-peer launches would be a visible fraction of the step."""

    def __init__(self, lp, nvar: int, device, group=None):
        self.group = group
        self.nvar = nvar
        self.rank = lp.rank
        self.peers = sorted(set(lp.send_ids) | set(lp.recv_ids))
        dev = torch.device(device)
        as_idx = lambda a: torch.as_tensor(a, dtype=torch.long, device=dev)
        self.send_peers = [q for q in self.peers if q in lp.send_ids and len(lp.send_ids[q])]
        self.recv_peers = [q for q in self.peers if q in lp.recv_ids and len(lp.recv_ids[q])]
        cat = lambda ids, qs: as_idx([i for q in qs for i in ids[q]]) if qs else as_idx([])
        self.send_idx = cat(lp.send_ids, self.send_peers)
        self.recv_idx = cat(lp.recv_ids, self.recv_peers)
        self.send_buf = torch.empty((self.send_idx.numel(), nvar), dtype=torch.float64, device=dev)
        self.recv_buf = torch.empty((self.recv_idx.numel(), nvar), dtype=torch.float64, device=dev)
        self.bytes_per_step = 8 * nvar * int(self.send_idx.numel())
        # gloo cannot move device tensors: stage through host buffers (test rigs with one GPU only;
        # the production backend is "nccl" = RCCL, which sends the device buffers directly)
        self.host_staged = dev.type == "cuda" and dist.is_initialized() and dist.get_backend(group) == "gloo"
        if self.host_staged:
            self.send_host = torch.empty_like(self.send_buf, device="cpu").pin_memory()
            self.recv_host = torch.empty_like(self.recv_buf, device="cpu").pin_memory()
        sb = self.send_host if self.host_staged else self.send_buf
        rb = self.recv_host if self.host_staged else self.recv_buf

        def views(buf, ids, qs):
            out, o = {}, 0
            for q in qs:
                n = len(ids[q])
                out[q] = buf[o:o + n]
                o += n
            return out
        self.send_view = views(sb, lp.send_ids, self.send_peers)
        self.recv_view = views(rb, lp.recv_ids, self.recv_peers)
        self._ops = None   # the P2POp list is the same every step: built once (needs the process group to exist)

    def exchange_many(self, fields):
        """Several nodal fields in ONE grouped message per peer: fields = [tensor [n_node_local][w_k], ...] with
        sum(w_k) == nvar.  The coupled HCC + solid step (BASELINE config 5) moves the HCC unknowns (3 per node) and the
        CURRENT coordinates of the moving mesh (3 per node: the solid system's solution, src/solid_system.C:103-123)
        together: the message count, not the bytes, is what a step pays for."""
        if sum(int(f.shape[1]) for f in fields) != self.nvar:
            raise ValueError("field widths do not add up to the exchange's nvar")
        if not self.send_peers and not self.recv_peers:
            return
        if self.send_idx.numel():
            o = 0
            for f in fields:
                w = int(f.shape[1])
                self.send_buf[:, o:o + w].copy_(f.index_select(0, self.send_idx))
                o += w
        self._round()
        if self.recv_idx.numel():
            o = 0
            for f in fields:
                w = int(f.shape[1])
                f.index_copy_(0, self.recv_idx, self.recv_buf[:, o:o + w])
                o += w

    def _round(self):
        """send_buf -> peers -> recv_buf (one grouped isend / irecv round)"""
        if self.send_idx.numel() and self.host_staged:
            self.send_host.copy_(self.send_buf, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        if self._ops is None:
            self._ops = [dist.P2POp(dist.irecv, self.recv_view[q], q, group=self.group) for q in self.recv_peers]
            self._ops += [dist.P2POp(dist.isend, self.send_view[q], q, group=self.group) for q in self.send_peers]
        for w in dist.batch_isend_irecv(self._ops):
            w.wait()
        if self.recv_idx.numel() and self.host_staged:
            self.recv_buf.copy_(self.recv_host, non_blocking=True)

    def exchange(self, u: torch.Tensor):
        """u: [n_node_local][nvar]; owned rows are read, ghost rows are overwritten in place."""
        if not self.send_peers and not self.recv_peers:
            return
        if self.send_idx.numel():
            torch.index_select(u, 0, self.send_idx, out=self.send_buf)
            if self.host_staged:
                self.send_host.copy_(self.send_buf, non_blocking=True)
                torch.cuda.current_stream().synchronize()
        if self._ops is None:
            self._ops = [dist.P2POp(dist.irecv, self.recv_view[q], q, group=self.group) for q in self.recv_peers]
            self._ops += [dist.P2POp(dist.isend, self.send_view[q], q, group=self.group) for q in self.send_peers]
        for w in dist.batch_isend_irecv(self._ops):
            w.wait()
        if self.recv_idx.numel():
            if self.host_staged:
                self.recv_buf.copy_(self.recv_host, non_blocking=True)
            u.index_copy_(0, self.recv_idx, self.recv_buf)
