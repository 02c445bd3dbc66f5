"""RCCL smoke on the one GPU: torch.distributed with backend "nccl" (= RCCL on ROCm), world size 1 -- the library is loaded,
a communicator is created and a device collective runs -- then the overlap step of bench.py (part 1, exchange, part 2 on
two streams) with a partition that has no peers.  The N > 1 runs are the driver's; this removes the first-ever-RCCL-load
risk from them ("dist.init_process_group('nccl') has never executed")."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RDC_ROOT"])
from rdcfes_amd import AssemblyContext, partition, pihna_params_from_dict, synth
from rdcfes_amd.context import FIELD_OLD_SOLUTION
from rdcfes_amd.halo import HaloExchange
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.arange(8, dtype=torch.float64, device=dev)
dist.all_reduce(t)                     # RCCL communicator + one device collective
dist.barrier()
assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float64))
conn, xyz = synth.kuhn_tet_mesh(10, order="lex")
u = synth.pihna_fields(xyz)
p = pihna_params_from_dict(synth.pihna_param_dict("shipped"))
lp = partition.build_local(conn, xyz, np.zeros(conn.shape[0], dtype=np.int32), 0, 1)
assert lp.n_owned == xyz.shape[0] and not lp.send_ids and not lp.recv_ids and lp.n_interior == lp.n_owned
hx = HaloExchange(lp, 5, dev)
u_t = torch.from_numpy(u[lp.node_global]).to(dev)
main_s, halo_s = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
with AssemblyContext(0) as ctx:
    ctx.set_stream(main_s.cuda_stream)
    n_int = int(0.7 * lp.n_owned)       # no ghosts: any prefix is interior; a real split so that both parts launch
    ctx.set_option("interior_nodes", n_int)
    ctx.mesh_upload(4, lp.conn, lp.xyz, 5, n_owned=lp.n_owned)
    ctx.field_bind_device(FIELD_OLD_SOLUTION, u_t.data_ptr(), u_t.numel())
    ctx.assemble_pihna(p)
    val0, rhs0 = ctx.csr_download()
    for _ in range(3):                  # the step of bench.py --gpus N --overlap 1
        halo_s.wait_stream(main_s)
        ctx.assemble_pihna_part(p, 1, main_s.cuda_stream)
        with torch.cuda.stream(halo_s):
            hx.exchange(u_t)
        ctx.assemble_pihna_part(p, 2, halo_s.cuda_stream)
        main_s.wait_stream(halo_s)
    torch.cuda.synchronize()
    val, rhs = ctx.csr_download()
    n1 = ctx.part1_nodes()
err = float(np.linalg.norm(val - val0) / np.linalg.norm(val0))
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"ok": True, "backend": "nccl", "err": err, "part1_nodes": int(n1)}))
'''


def test_rccl_world_size_one_and_the_overlap_step():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RDC_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["err"] < 1e-13 and out["part1_nodes"] > 0
