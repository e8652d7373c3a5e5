"""CPU, world_size 2, gloo: the N>1 path of bench.py (weight broadcast, batch split, result gather)."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import __graft_entry__ as ge
ge.import_package()
sh = importlib.import_module("paddle_lite_amd.sharding")
wl = importlib.import_module("paddle_lite_amd.workloads")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W = wl.make_mobilenet_v1_weights(seed=77) if rank == 0 else None
W = sh.broadcast_weights(W, dist, torch.device("cpu"), rank, world)
ref = wl.make_mobilenet_v1_weights(seed=77)
for name, v in ref.items():
    if isinstance(v, dict):
        for f, a in v.items():
            assert np.array_equal(W[name][f], a) and W[name][f].dtype == a.dtype, (name, f)
    else:
        assert W[name] == v
# batch split: ragged global batch 7 over 2 ranks -> [0,4) and [4,7); equal shards are gathered rank-major
lo, hi = sh.shard_range(7, rank, world)
assert (lo, hi) == ((0, 4), (4, 7))[rank]
lo, hi = sh.shard_range(8, rank, world)
local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1).repeat(1, 3)
g = sh.all_gather_rows(local, dist, world)
assert g.shape == (8, 3) and torch.equal(g[:, 0], torch.arange(8, dtype=torch.float32))
# pipelined gather (bench.py N > 1): 5 steps through 2 buffers, every step's result must be that step's shards
pg = sh.PipelinedGather(local, dist, world, depth=2)
outs = []
for step in range(5):
    buf = pg.stage_buffer()
    buf.copy_(local + 100.0 * step)
    outs.append((step, pg.launch()))
    if step >= 1:  # the previous step's buffer is not reused before the next stage_buffer(): check it after a wait
        pstep, pout = outs[-2]
        h = pg.pending[pstep & 1]
        if h is not None:
            h.wait()
        assert torch.equal(pout[:, 0], torch.arange(8, dtype=torch.float32) + 100.0 * pstep), (pstep, pout[:, 0])
pg.drain()
assert torch.equal(outs[-1][1][:, 0], torch.arange(8, dtype=torch.float32) + 400.0)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_broadcast_split_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    report = "\n".join("---- rank %d (rc %s) ----\n%s" % (r, p.returncode, o) for r, (p, o) in enumerate(zip(procs, outs)))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok" % r in o, report


def test_pack_unpack_roundtrip_and_shards(pkg):
    sh = importlib.import_module("paddle_lite_amd.sharding")
    W = {"a": {"w": np.arange(7, dtype=np.int8), "s": np.float32(0.25)}, "b": np.float32(3.0),
         "c": {"m": np.ones((2, 3), np.float32)}}
    items, blob = sh.pack_weights(W)
    assert all(off % 16 == 0 for (_, _, _, _, off, _) in items)
    U = sh.unpack_weights(items, blob)
    assert np.array_equal(U["a"]["w"], W["a"]["w"]) and U["a"]["s"] == np.float32(0.25) and U["b"] == np.float32(3.0)
    assert U["c"]["m"].shape == (2, 3)
    for gb, world in ((1024, 8), (7, 4), (3, 8), (128, 1)):
        rs = [sh.shard_range(gb, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == gb and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
