"""bench.py's own N > 1 launch path on the CPU: `--gpus 2` without a launcher environment must start two rank
processes, rendezvous (gloo here, RCCL on GPUs), broadcast the network, split one global batch with shard_range, gather
every step's result and print ONE JSON line with n_gpus == 2.  PLHIP_BENCH_DRYRUN=1 replaces the predictor by a stand-in
without a device (the control flow is what is tested; the numbers mean nothing and the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(PLHIP_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_gpus_2_spawns_two_ranks_weak():
    rc, out, err = _run(["--gpus", "2", "--steps", "7", "--warmup", "3", "--inflight", "2", "--batch", "5", "--res", "32"])
    assert rc == 0, err[-3000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 10
    assert line["steps"] == 7 and line["data"] == "dryrun-no-compute" and line["value"] > 0
    assert line["metric"] == "INT8 images/sec MobileNetV1 224x224"
    # the default (weak) multi-GPU run also carries the strong-scaling point: one large batch split over the ranks
    st = line["strong"]
    assert st and st["scaling"] == "strong" and st["global_batch"] == 1024 and st["images_per_gpu"] == 512 and st["value"] > 0


def test_strong_scaling_ragged_global_batch_three_ranks():
    # 7 images over 3 ranks -> shards of 3, 2, 2 rows: padded staging slots, rank-major order checked inside bench.py
    rc, out, err = _run(["--gpus", "3", "--steps", "9", "--warmup", "2", "--inflight", "3", "--scaling", "strong",
                         "--global-batch", "7", "--config", "c5", "--res", "32"])
    assert rc == 0, err[-3000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["scaling"] == "strong" and line["config"]["global_batch"] == 7 and line["strong"] is None
    assert "MobileNetV2" in line["metric"]


def test_eight_ranks_strong_scaling_ragged_shards():
    """The driver's N = 8 point of BASELINE config C5 (`bench.py --config c5 --gpus 8`: strong scaling by default), rehearsed on
    the CPU with gloo: 8 rank processes, a global batch that does NOT divide by 8 (1021 -> shards of 128 x 5 + 127 x 3),
    network broadcast, scatter, one gather per step in rank-major order (checked inside bench.py), ONE JSON line."""
    rc, out, err = _run(["--gpus", "8", "--steps", "5", "--warmup", "2", "--inflight", "2", "--config", "c5", "--global-batch", "1021",
                         "--res", "32", "--windows", "2"], timeout=600)
    assert rc == 0, err[-3000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["global_batch"] == 1021
    assert line["config"]["parallelism"].startswith("batch split x8") and line["value"] > 0 and "MobileNetV2" in line["metric"]


def test_a_predictor_far_ahead_of_the_others_does_not_take_their_staging_slots():
    """Round 4's hang of the N > 1 path: with 4 predictors and 10 staging slots, predictor 0 (no start offset) could run three of
    its steps (0, 4, 8, 12) before predictor 2 woke up, take the slot step 2 needed (12 % 10), and the coordinator, which issues
    the collectives in step order, waited for step 2 for ever.  Here every predictor starts 30 ms after the previous one and a
    step takes ~1 ms: predictor 0 is 12 steps ahead of predictor 2 at once.  The slots are handed on in step order now."""
    rc, out, err = _run(["--gpus", "2", "--steps", "17", "--warmup", "13", "--inflight", "4", "--batch", "3", "--res", "32", "--windows", "2"],
                        {"PLHIP_BENCH_STAGGER_US": "30000"}, timeout=240)
    assert rc == 0, err[-3000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 17 and line["value"] > 0


def test_world_size_mismatch_is_an_error():
    rc, out, err = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and "WORLD_SIZE=2" in err and out.strip() == ""


def test_single_process_line_shape():
    rc, out, err = _run(["--steps", "3", "--warmup", "1", "--batch", "4", "--res", "32"])
    assert rc == 0, err[-3000:]
    line = json.loads(out.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line
    assert line["n_gpus"] == 1 and line["config"]["name"] == "c3"


def test_a_rank_dying_before_the_rendezvous_ends_the_launch_quickly():
    """spawn_ranks polls every child: rank 1 exits 7 before init_process_group, rank 0 would otherwise sit in the c10d
    rendezvous until its timeout; the launcher must terminate it and return non-zero within seconds."""
    import time
    t0 = time.time()
    rc, out, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--res", "32"],
                        {"PLHIP_BENCH_DIE_RANK": "1"}, timeout=120)
    assert rc != 0 and "rank exit codes" in err
    assert time.time() - t0 < 60
    assert not [l for l in out.splitlines() if l.startswith("{")]
