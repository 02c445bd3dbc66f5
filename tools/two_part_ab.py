"""Cost of the two-part assembly (rdc_set_option "part") that bench.py --gpus N uses to overlap the halo exchange:
one process, one GPU, a K(n) mesh of per-GPU size, the trailing `--boundary` fraction of the nodes declared "near a
ghost", and a stand-in for the exchange (gather + device copy + scatter of those nodes on the side stream; a real
exchange adds the RCCL latency on top).

  whole        one launch (N = 1 baseline, no exchange)
  seq          exchange, then one launch              (bench.py --overlap 0)
  two_same     exchange on the side stream | part 1, wait, part 2 on the main stream
  two_streams  part 1 on the main stream | exchange + part 2 on the side stream   (bench.py --overlap 1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rdcfes_amd import AssemblyContext, synth  # noqa: E402
from rdcfes_amd.context import FIELD_OLD_SOLUTION  # noqa: E402
from rdcfes_amd.params import pihna_params_from_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh-n", dest="n", type=int, default=60)
    ap.add_argument("--boundary", type=float, default=0.1)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    conn, xyz = synth.kuhn_tet_mesh(a.n, order="lex")
    u = synth.pihna_fields(xyz)
    p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
    n_node = xyz.shape[0]
    n_int = int((1.0 - a.boundary) * n_node)
    ctx = AssemblyContext(0)
    main_s, side_s = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
    ctx.set_stream(main_s.cuda_stream)
    ctx.mesh_upload(4, conn, xyz, 5)
    u_t = torch.from_numpy(u).to(dev)
    ctx.field_bind_device(FIELD_OLD_SOLUTION, u_t.data_ptr(), u_t.numel())
    ctx.set_option("interior_nodes", n_int)
    idx = torch.arange(n_int, n_node, device=dev)
    sbuf = torch.empty((idx.numel(), 5), dtype=torch.float64, device=dev)
    rbuf = torch.empty_like(sbuf)

    def exchange():
        torch.index_select(u_t, 0, idx, out=sbuf)
        rbuf.copy_(sbuf)
        u_t.index_copy_(0, idx, rbuf)

    def whole():
        ctx.set_option("part", 0)
        ctx.assemble_pihna(p)

    def seq():
        exchange()
        whole()

    def two_same():
        side_s.wait_stream(main_s)
        with torch.cuda.stream(side_s):
            exchange()
        ctx.set_option("part", 1)
        ctx.assemble_pihna(p)
        main_s.wait_stream(side_s)
        ctx.set_option("part", 2)
        ctx.assemble_pihna(p)

    def two_streams():
        side_s.wait_stream(main_s)
        with torch.cuda.stream(side_s):
            exchange()
        ctx.set_option("part", 1)
        ctx.assemble_pihna(p)
        ctx.set_stream(side_s.cuda_stream)
        ctx.set_option("part", 2)
        ctx.assemble_pihna(p)
        ctx.set_stream(main_s.cuda_stream)
        main_s.wait_stream(side_s)

    out = {"mesh": f"K({a.n})", "tets": int(conn.shape[0]), "boundary_fraction": a.boundary}
    for name, fn in [("whole", whole), ("seq", seq), ("two_same", two_same), ("two_streams", two_streams), ("whole_again", whole)]:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) / a.steps * 1e3
    ctx.set_option("part", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
