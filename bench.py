#!/usr/bin/env python3
"""bench.py — INT8 images/sec of the BASELINE.json graphs on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c4|c5|c2] [--scaling weak|strong]
                    [--batch B | --global-batch B] [--inflight P]

Default (what the driver runs): config C3 = MobileNetV1-INT8 full graph, batch 128 per GPU (BASELINE.json metric).
  c4 = ResNet50-INT8 batch 256, c5 = MobileNetV2-INT8 global batch 1024 split over the GPUs (strong scaling),
  c2 = the single conv2d_int8 op (N=32 Cin=64 Cout=128 56x56 k3 s1), N = 1 only.
A step = one pass of the lowered program (calib -> int8 convs [+ fp32 pool / residual adds] -> pool -> calib -> fc ->
softmax) over one synthetic batch, through the C++ kHIP kernel classes and libplhip.so.  By
default 4 predictors per GPU (one host thread + HIP stream each, the reference's predictor-per-thread serving model)
run whole steps, dealt round-robin, so four steps are in flight and fill each other's dispatch gaps (`--inflight 1`:
strictly serial; also reported as `single_stream`; 3 was the default until the end of round 3); inside a timed window
predictor i submits its first step i/4 of a step time late, so that they do not run the same layer at the same time.  Inputs are resident in HBM when the timed region
starts.

N > 1: one process per GPU (torch.distributed "nccl" == RCCL over xGMI).  `--gpus N` without a launcher environment
makes this script start its own N rank processes (fresh children, before anything touches the GPU); under
`torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE.  Rank 0 builds the network and broadcasts it, generates
ONE global batch and scatters the shards (sharding.shard_range); every step each rank runs its shard (no collective in
the layer loop) and the [rows, 1000] probabilities are all-gathered asynchronously, overlapping the next steps.
weak: global batch = --batch x N; strong: global batch fixed (--global-batch).

The JSON line carries `roofline` for the dominant kernel family (launch time measured live with HIP events on the launch
stream) and `cpu_baseline` (oracle port of the reference algorithm on the host cores, bounded sample; rank 0, N = 1).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Denominators (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8.0 TB/s spec; int8 MFMA dense =
# 2x the bf16 rate = 256 CU x 4 SIMD x 2048 op/clk x 2.4 GHz = 5.03 POP/s (sparsity figures not used).
HBM_PEAK_GBS = 8000.0
MFMA_I8_PEAK_TOPS = 256 * 4 * 2048 * 2.4e9 / 1e12
BALANCE_OPS_PER_BYTE = MFMA_I8_PEAK_TOPS * 1e12 / (HBM_PEAK_GBS * 1e9)

CONFIGS = {
    "c3": dict(model="mobilenet_v1", batch=128, title="MobileNetV1 INT8 full graph 224x224 (27 int8 convs + pool + fc + softmax)",
               metric="INT8 images/sec MobileNetV1 224x224"),
    "c4": dict(model="resnet50", batch=256, title="ResNet50 INT8 full graph 224x224 (53 int8 convs, max pool, 16 residual adds)",
               metric="INT8 images/sec ResNet50 224x224"),
    "c5": dict(model="mobilenet_v2", batch=128, global_batch=1024, scaling="strong",
               title="MobileNetV2 INT8 full graph 224x224 (52 int8 convs, relu6, 10 residual adds)",
               metric="INT8 images/sec MobileNetV2 224x224"),
    "c2": dict(model="conv", batch=32, title="single conv2d_int8 op N=32 Cin=64 Cout=128 HW=56 k=3 s=1 p=1, int8 out",
               metric="INT8 images/sec conv2d 64->128 3x3 56x56"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)   # ~0.13 s timed at 0.42 ms / step: steadier than a 20 ms window
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU and step (weak scaling); default: the config's")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None)
    ap.add_argument("--global-batch", type=int, default=None, help="strong scaling: images per step over all GPUs")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each, back to back; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the oracle comparison of the last step's output")
    ap.add_argument("--layer-table", action="store_true", help="print the per-instruction timing table to stderr")
    ap.add_argument("--inflight", type=int, default=4,
                    help="predictors per GPU, each on its own host thread + HIP stream, each running WHOLE steps; the steps "
                         "are dealt round-robin, so --inflight steps are in flight at once (the reference's "
                         "one-predictor-per-thread serving model, cxx_api.h:103-137). 1 = strictly serial steps.")
    ap.add_argument("--res", type=int, default=224, help=argparse.SUPPRESS)  # tests only
    return ap.parse_args(argv)


# =====================================================================================================================
# self-launch: `bench.py --gpus N` without a launcher environment
# =====================================================================================================================
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """Start N fresh rank processes of this script (nothing in this process has touched the GPU: torch is not even
    imported yet), relay rank 0's JSON line, fail if any rank failed or fewer than N joined."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    # poll ALL children: a rank that dies before the rendezvous must not leave the others waiting in init_process_group
    # (fresh processes only: nothing that touched the GPU is ever re-executed)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = (chunks[0] if chunks else b"").decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(rc != 0 for rc in rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        return 1
    try:
        line = json.loads([l for l in out0.splitlines() if l.startswith("{")][-1])
    except Exception:  # noqa: BLE001
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    if line.get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d GPUs, %s ranks joined\n" % (args.gpus, line.get("n_gpus")))
        return 1
    return 0


# =====================================================================================================================
# engines: one predictor on its own stream (HipEngine), or a stand-in without a device (DryEngine; PLHIP_BENCH_DRYRUN=1,
# used by the CPU tests of the multi-process control flow: spawn, rendezvous, scatter, per-step gather, timing)
# =====================================================================================================================
class HipEngine:
    def __init__(self, torch, lite, wl, local_rank, dev, net, rows, image, cfg):
        self.torch = torch
        self.stream = torch.cuda.Stream(dev)
        self.pred = lite.Predictor(local_rank, stream=self.stream.cuda_stream)
        if cfg["model"] == "conv":
            self.pred.add_feed("image", image.shape, lite.PREC_INT8)
            self.pred.add_io_copy("image", "xd", True)
            o = net["ops"][0]
            p = o["pad"]
            self.pred.add_conv("conv2d", "xd", o["name"], o["w"], o["bias"], (o["stride"],) * 2, (p, p, p, p), (1, 1), 1, o["act"],
                               0.0, float(o["in_scale"]), o["w_scale"], float(o["out_scale"]), True)
            self.out_var, self.plan = o["name"], None
        else:
            wl.emit_graph(self.pred, net, rows, fuse_dwpw={"0": False, "1": True}.get(os.environ.get("PLHIP_BENCH_FUSE_DWPW", ""), None))
            self.plan = self.pred.graph_plan()
            self.pred.graph_lower()
            self.out_var = net["output"]
        self.pred.set_input("image", image)
        self.pred.run(skip_io_copy=False)  # uploads the feed; PrepareForRun (weight pack, scale fold) everywhere
        self.pred.sync()
        self.use_graph = os.environ.get("PLHIP_BENCH_GRAPH", "0") == "1"

    def run(self):
        if self.use_graph:
            self.pred.run_graph()  # the step as ONE recorded launch graph (hipGraph): recorded at the first call
        else:
            self.pred.run(skip_io_copy=True)

    def stage(self, dst, nbytes):
        self.pred.copy_var_to_device(self.out_var, dst.data_ptr(), nbytes)

    def record(self):
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def wait_event(self, ev):
        if ev is not None:
            self.stream.wait_event(ev)

    def close(self):
        self.pred.close()


class DryEngine:
    """No device, no compute: "probabilities" whose first column is the global image index (so that the gathered result
    can be checked for order and completeness)."""

    def __init__(self, torch, lo, rows, classes):
        self.fake = torch.zeros((rows, classes), dtype=torch.float32)
        self.fake[:, 0] = torch.arange(lo, lo + rows, dtype=torch.float32)
        self.plan = None

    def run(self):
        time.sleep(0.0005)

    def stage(self, dst, nbytes):
        dst[:self.fake.shape[0]].copy_(self.fake)

    def record(self):
        return None

    def wait_event(self, ev):
        pass

    def close(self):
        pass


def build_net(wl, cfg, res):
    import numpy as np
    if cfg["model"] == "mobilenet_v1":
        return wl.mobilenet_v1_net(seed=1234, res=res)
    if cfg["model"] == "resnet50":
        return wl.resnet50_net(res=res)
    if cfg["model"] == "mobilenet_v2":
        return wl.mobilenet_v2_net(res=res)
    # c2: the reference test's scale convention (conv_int8_compute_test.cc:190-203)
    rng = np.random.default_rng(2000)
    cin, cout, k = 64, 128, 3
    op = dict(op="conv2d", name="y", src="image", w=rng.integers(-127, 128, (cout, cin, k, k)).astype(np.int8),
              bias=rng.uniform(-1, 1, cout).astype(np.float32), stride=1, pad=1, groups=1, act=1, act_coef=0.0,
              in_scale=np.float32(1 / 127.0), w_scale=np.full(cout, 1 / 127.0, np.float32), out_scale=np.float32(cin * k * k / 127.0))
    return dict(ops=[op], input="image", input_shape=(cin, 56, 56), output="y", shapes={"y": (cout, 56, 56)})


def by_family(fam_out):
    """Every kernel family of the step against both roofs (the `roofline` object is the one with the largest summed time)."""
    out = {}
    for k, f in (fam_out or {}).items():
        if not f.get("ms"):
            continue
        out[k] = {"ms": f["ms"], "launches": f["launches"], "hbm_frac": round(f["GB/s"] / HBM_PEAK_GBS, 4),
                  "mfma_frac": round(f["TOP/s"] / MFMA_I8_PEAK_TOPS, 4)}
    return out


def csrc_sha256():
    """Hash of the kernel sources: `roofline.traffic` comes from a committed PMC pass and is only valid for the kernels that
    pass measured (tools/pmc_traffic.py stores the same hash; there is no git on the GPU box)."""
    import hashlib
    d = os.path.join(ROOT, "paddle-lite_amd", "csrc")
    hsh = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            hsh.update(f.encode())
            hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def cpu_baseline(cfg, net, seconds):
    """The reference algorithm restated for the host (oracle/): im2col + int8 GEMM over (batch, group) for the dense convs
    (conv_impl.cc:490-598 structure), direct loops for depthwise, fused float epilogue, fp32 pool / add, OpenMP.
    Bounded by time; batch 1 per pass like the reference's own benchmark (benchmark.md:35-39)."""
    import numpy as np
    from oracle import graph_oracle, plref
    rng = np.random.default_rng(99)
    c, h, w = net["input_shape"]
    done, t0, per_pass = 0, time.perf_counter(), []
    while True:
        tp = time.perf_counter()
        if cfg["model"] == "conv":
            o = net["ops"][0]
            x = rng.integers(-127, 128, (1, c, h, w)).astype(np.int8)
            sh = plref.shape(1, c, h, w, o["w"].shape[0], 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1)
            plref.conv2d(sh, x, o["w"], o["bias"], float(o["in_scale"]), o["w_scale"], float(o["out_scale"]), 1, 0.0, True, via_gemm=True)
        else:
            graph_oracle.forward(plref, net, rng.uniform(-1, 1, (1, c, h, w)).astype(np.float32), keep=set(), via_gemm=True)
        done += 1
        per_pass.append(time.perf_counter() - tp)
        el = time.perf_counter() - t0
        if el >= seconds or done >= 100000:
            break
    c1 = None
    if cfg["model"] == "mobilenet_v1":
        # BASELINE config C1 beside it: the same network in fp32 through the reference's x86 path restated (im2col + SGEMM per
        # image and group, batch_norm / relu as separate fp32 passes; all-ones 1x3x224x224 input, test_mobilenetv1_lite_x86.cc)
        from oracle import x86_path
        c1 = x86_path.time_c1(net, seconds=min(4.0, max(1.0, seconds / 3)))
        c1["cores"] = int(os.environ["OMP_NUM_THREADS"])
        c1["note"] = ("MobileNetV1 fp32 1x3x%dx%d, oracle/x86_path.py: restatement of lite/kernels/x86/conv_compute.h:48-150 "
                      "(the reference's x86 build needs MKLML / gflags / protobuf downloads: unbuildable here)" % (h, w))
    return {"value": round(done / el, 2), "unit": "img/s", "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
            "ms_per_image_min_avg_max": [round(1e3 * min(per_pass), 3), round(1e3 * sum(per_pass) / len(per_pass), 3), round(1e3 * max(per_pass), 3)],
            "threads": int(os.environ["OMP_NUM_THREADS"]),
            "fp32_x86_path_ms": c1["avg_ms"] if c1 else None, "fp32_x86_path": c1,
            "sample": "%d images of the same graph, batch 1 each, %.1f s; oracle/ restatement of the reference's im2col+GEMM int8 "
                      "path (its ARM NEON kernels cannot run on x86; its x86 backend has no INT8 kernels)" % (done, el)}


def oracle_selfcheck(np, pred, net, cfg, images, out_var):
    """Compare the predictor's variables for images[0:2] (the first rows of the last timed step) with oracle/ on the same
    bytes.  Test infrastructure use of the oracle: the checker, never the thing measured."""
    from oracle import graph_oracle, plref
    n = images.shape[0]
    if cfg["model"] == "conv":
        o = net["ops"][0]
        c, h, w = net["input_shape"]
        sh = plref.shape(n, c, h, w, o["w"].shape[0], 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1)
        want, _ = plref.conv2d(sh, images, o["w"], o["bias"], float(o["in_scale"]), o["w_scale"], float(o["out_scale"]), 1, 0.0, True, via_gemm=True)
        got = pred.get_var(out_var, np.int8)[:n]
        bad = int((got != want).sum())
        return {"ok": bad == 0, "images": n, "checked": {out_var: "int8 bit-exact"}, "mismatches": bad}
    ref = graph_oracle.forward(plref, net, images, keep=None, via_gemm=True)
    checked, bad = {}, 0
    names = [v for v in ref if v != net["input"]]
    # every int8 variable that survives in the lowered program, plus the fp32 output
    for v in names:
        if ref[v].dtype != np.int8 and v != out_var:
            continue
        try:
            got = pred.get_var(v, ref[v].dtype, max_bytes=(1 << 30) if v == out_var else (96 << 20))[:n]
        except Exception:  # noqa: BLE001  (fused away in the lowered program)
            continue
        if got.shape != ref[v].shape:
            continue
        if ref[v].dtype == np.int8:
            nb = int((got != ref[v]).sum())
            checked[v] = "int8 bit-exact"
        else:
            nb = int((~np.isclose(got, ref[v], rtol=1e-4, atol=1e-7)).sum())
            checked[v] = "fp32 rtol 1e-4 atol 1e-7"
        bad += nb
    return {"ok": bad == 0 and out_var in checked, "images": n, "variables_checked": len(checked), "output": checked.get(out_var), "mismatches": bad}


def main():
    args = parse()
    wd = os.environ.get("PLHIP_BENCH_WATCHDOG")  # seconds: dump every thread's Python stack and exit if still running by then
    if wd:
        import faulthandler
        faulthandler.dump_traceback_later(float(wd), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    if os.environ.get("PLHIP_BENCH_DIE_RANK") == str(rank) and world > 1:  # tests only: a rank that dies before the rendezvous
        sys.exit(7)
    # stdout carries exactly ONE line, the JSON record: everything else that libraries print there (RCCL's version banner,
    # c10d notices) is sent to stderr by pointing fd 1 at fd 2 and keeping the real stdout aside
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    cfg = dict(CONFIGS[args.config])
    if cfg["model"] == "conv" and world > 1:
        sys.stderr.write("bench.py: config c2 is a single-GPU op benchmark\n")
        sys.exit(2)
    dry = os.environ.get("PLHIP_BENCH_DRYRUN") == "1"
    # the CPU baseline's OpenMP team = the cores this process may actually use (capped at 32: the per-layer GEMMs of one
    # image are too small to feed more threads)
    os.environ.setdefault("OMP_NUM_THREADS", str(min(32, len(os.sched_getaffinity(0)))))
    import numpy as np
    import torch
    import torch.distributed as dist
    import queue
    import threading
    import __graft_entry__ as ge
    pkg_ = ge.import_package()
    lite = importlib.import_module("paddle_lite_amd.liteapi")
    if not dry:
        pkg_.capi.load()  # the same libplhip.so the predictor library links: forwards PLHIP_<KNOB> diagnostics (printed as debug_knobs)
    wl = importlib.import_module("paddle_lite_amd.workloads")
    sharding = importlib.import_module("paddle_lite_amd.sharding")

    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box (never used by the driver): PLHIP_BENCH_SAME_GPU=1 puts every rank on device 0 and
    # PLHIP_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
    if os.environ.get("PLHIP_BENCH_SAME_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("PLHIP_BENCH_BACKEND", "gloo" if dry else "nccl")
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    # PLHIP_BENCH_FORCE_DIST=1: run the multi-rank code path (RCCL init, network broadcast, batch scatter, per-step
    # all_gather, barriers) even with ONE rank: the only way to exercise the RCCL calls on a one-GPU box
    use_dist = world > 1 or os.environ.get("PLHIP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("only %d of %d ranks joined" % (dist.get_world_size(), args.gpus))

    def run_case(scaling, global_batch_arg, per_gpu_arg, details):
        """One timed case: build the predictors for this split, warm up, time `windows` x `steps` steps, tear down.
        details: also the serial rate, the per-kernel table / roofline, the oracle self-check and the CPU baseline (headline
        case only; the extra strong-scaling point of a multi-GPU run carries its rate alone)."""
        # ---- work split ----
        per_gpu = per_gpu_arg or cfg["batch"]
        if scaling == "strong":
            global_batch = global_batch_arg or cfg.get("global_batch", per_gpu)
        else:
            global_batch = per_gpu * world
        lo, hi = sharding.shard_range(global_batch, rank, world)
        rows = hi - lo
        max_rows = max(sharding.shard_range(global_batch, r, world)[1] - sharding.shard_range(global_batch, r, world)[0]
                       for r in range(world))
        if rows < 1:
            raise SystemExit("global batch %d leaves rank %d without work" % (global_batch, rank))

        # ---- network: rank 0 generates, RCCL broadcast over xGMI, every rank builds its predictors from the bytes ----
        net = build_net(wl, cfg, args.res) if rank == 0 else None
        net = sharding.broadcast_net(net, dist, dev, rank, world, force=use_dist)
        classes = net["shapes"][net["output"]][0] * net["shapes"][net["output"]][1] * net["shapes"][net["output"]][2]

        # ---- ONE global batch, generated on rank 0 and scattered (device to device over xGMI), resident before timing ----
        c, h, w = net["input_shape"]
        images = None
        if rank == 0 and not dry:
            rng = np.random.default_rng(1000)
            if cfg["model"] == "conv":
                images = rng.integers(-127, 128, (global_batch, c, h, w)).astype(np.int8)
            else:
                images = rng.uniform(-1, 1, (global_batch, c, h, w)).astype(np.float32)
        check_images = images[:2].copy() if images is not None else None  # rank 0's shard starts at image 0
        if dry:
            image = None
        elif cfg["model"] == "conv":
            image = images
        else:
            image = sharding.scatter_batch(images, global_batch, (c, h, w), dist, dev, rank, world, force=use_dist)
            images = None

        P = max(1, args.inflight)
        # Start offset between the predictors of a window: predictor i submits its first step i * stagger late, stagger = a
        # P-th of the step time the previous window (first window: the warm-up) measured.  All P predictors are released
        # together at a window's start; starting level they run the same layers at the same time (three depthwise kernels, then
        # three GEMMs) until they drift apart, which a 20-step window barely has time for: 308.3 / 309.4 k -> 315.6 / 315.5 k
        # img/s in the driver's form `--steps 20 --warmup 5` (A/B twice in one call, 3 predictors), no change for 300-step windows.
        # With the offset a fourth predictor pays (without it 3 and 4 read the same): driver form 308.6 / 306.3 k -> 312.4 / 310.1 k
        # (c3), 57.1 -> 57.5-57.7 k (c4), 220.0 -> 223.8 k (c5).
        # PLHIP_BENCH_STAGGER_US: a fixed offset in us (0 = none).
        stagger_env = os.environ.get("PLHIP_BENCH_STAGGER_US")
        stagger = [float(stagger_env) * 1e-6 if stagger_env is not None else 0.0]
        out_bytes = rows * classes * (1 if cfg["model"] == "conv" else 4)
        engines, errs = [None] * P, []

        # ---- per-step result gather (N > 1): DEPTH staging slots; step s uses slot s % DEPTH.  The predictor thread stages
        # its probabilities behind the slot's previous collective (event) and tells the coordinator (queue, no reply needed);
        # the coordinator (this thread) issues the collectives in step order on a stream that carries no compute, so every
        # rank issues them in the same order and the all_gather of step s overlaps the kernels of the following steps. ----
        DEPTH = 2 * P + 2
        if use_dist:
            local = [torch.zeros((max_rows, classes), dtype=torch.float32, device=dev) for _ in range(DEPTH)]
            gathered = [torch.empty((world * max_rows, classes), dtype=torch.float32, device=dev) for _ in range(DEPTH)]
            # a slot is handed on in STEP ORDER: step g may stage into slot g % DEPTH once the collective of step g - DEPTH has been
            # issued (`issued` = number of steps whose collective the coordinator has issued, all windows).  A free-for-all
            # semaphore per slot deadlocked: predictor 0, three steps ahead of a predictor that was still asleep in its start
            # offset, took the slot of that predictor's first step, and the coordinator, which issues in step order, waited for
            # that first step for ever (round 4: the N > 1 path hung twice in the evidence script, never stand-alone)
            issued = [0]
            issued_cv = threading.Condition()
            slot_event = [None] * DEPTH
            pending = [None] * DEPTH
        coord_stream = None if dry else torch.cuda.Stream(dev)

        class Flight(threading.Thread):
            """Predictor i runs the steps i, i+P, i+2P, ... as whole steps on its own stream."""

            def __init__(self, i):
                super().__init__(daemon=True)
                self.i, self.cmd, self.done = i, threading.Semaphore(0), threading.Semaphore(0)
                self.doneq = queue.Queue()
                self.n, self.base, self.alive, self.err = 0, 0, True, None

            def run(self):
                try:
                    if dry:
                        engines[self.i] = DryEngine(torch, lo, rows, classes)
                    else:
                        torch.cuda.set_device(local_rank)
                        engines[self.i] = HipEngine(torch, lite, wl, local_rank, dev, net, rows, image, cfg)
                except Exception as e:  # noqa: BLE001
                    self.err = e
                self.done.release()
                while True:
                    self.cmd.acquire()
                    if not self.alive:
                        break
                    try:
                        e = engines[self.i]
                        if stagger[0] >= 2e-5 and self.i:  # predictors 1.. start their share i * stagger late (see above;
                            # offsets under 20 us — the single conv of c2 — are below what a host sleep can keep)
                            time.sleep(self.i * stagger[0])
                        for s_ in range(self.i, self.n, P):
                            e.run()
                            if use_dist:
                                gstep = self.base + s_
                                b = gstep % DEPTH
                                with issued_cv:              # host: the collective that last read this slot (step g - DEPTH) has been ISSUED
                                    while issued[0] < gstep - DEPTH + 1 and self.alive:
                                        issued_cv.wait(1.0)
                                e.wait_event(slot_event[b])  # device: ... and will have FINISHED before the copy below
                                e.stage(local[b], out_bytes)
                                self.doneq.put(e.record())
                    except Exception as ex:  # noqa: BLE001
                        self.err = ex
                    self.done.release()

        flights = [Flight(i) for i in range(P)]
        for f_ in flights:
            f_.start()
        for f_ in flights:
            f_.done.acquire()
            if f_.err:
                raise f_.err
        step_base = [0]
        last_gather = [None]

        def run_steps(n):
            """n steps in total, dealt round-robin over the P predictors; this thread only coordinates."""
            for f_ in flights:
                f_.n, f_.base = n, step_base[0]
                f_.cmd.release()
            if use_dist:
                for s_ in range(n):
                    f_ = flights[s_ % P]
                    while True:
                        try:
                            ev = f_.doneq.get(timeout=1.0)
                            break
                        except queue.Empty:
                            if f_.err:
                                raise f_.err
                    b = (step_base[0] + s_) % DEPTH
                    if dry:
                        pending[b] = dist.all_gather_into_tensor(gathered[b], local[b], async_op=True)
                        pending[b].wait()
                    else:
                        with torch.cuda.stream(coord_stream):
                            coord_stream.wait_event(ev)
                            pending[b] = dist.all_gather_into_tensor(gathered[b], local[b], async_op=True)
                            pending[b].wait()  # stream-level: coord_stream (no compute on it) is ordered behind the collective
                            fe = torch.cuda.Event()
                            fe.record(coord_stream)
                        slot_event[b] = fe
                    last_gather[0] = gathered[b]
                    with issued_cv:
                        issued[0] = step_base[0] + s_ + 1
                        issued_cv.notify_all()
            for f_ in flights:
                f_.done.acquire()
                if f_.err:
                    raise f_.err
            step_base[0] += n
            if use_dist and not dry:
                coord_stream.synchronize()  # every step's result is complete inside the timed region

        def sync_all():
            if not dry:
                torch.cuda.synchronize(dev)
            if use_dist:
                dist.barrier()
            if not dry:
                torch.cuda.synchronize(dev)

        sync_all()
        tw = time.perf_counter()
        run_steps(args.warmup)
        sync_all()
        if stagger_env is None and P > 1 and args.warmup >= P and not dry:
            # (the warm-up's own time may include first-touch costs: its estimate is capped low; later windows measure)
            stagger[0] = min((time.perf_counter() - tw) / args.warmup / P, 0.0002)
        # EXACTLY args.steps steps per timed window, barrier + synchronize on both sides (the contract); the window is repeated
        # back to back and the MEDIAN window is reported: a single 20-step window is ~9 ms here and moved the figure by 3-7 %
        # from run to run.  min / median / max over the windows are in the line.
        wins = []
        for _ in range(max(1, args.windows)):
            sync_all()
            t0 = time.perf_counter()
            run_steps(args.steps)
            sync_all()
            el = time.perf_counter() - t0
            if use_dist:
                tt = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            wins.append(el)
            if stagger_env is None and P > 1 and not dry:
                stagger[0] = min(el / args.steps / P, 0.005)
        elapsed = float(np.median(wins))

        # ---- outside the timed region: the gathered result of the last step is complete and in rank-major image order ----
        if use_dist and last_gather[0] is not None:
            g_ = last_gather[0].float().cpu().numpy().reshape(world, max_rows, classes)
            for r in range(world):
                l_, h_ = sharding.shard_range(global_batch, r, world)
                blk = g_[r, :h_ - l_]
                if dry:
                    assert np.array_equal(blk[:, 0], np.arange(l_, h_, dtype=np.float32)), "gathered rows out of order"
                else:
                    assert np.all(np.isfinite(blk)) and np.allclose(blk.sum(-1), 1.0, rtol=1e-3), "gathered probabilities are not distributions"

        # ---- self-check, outside the timed region and BEFORE anything is re-run: what the last timed step of EVERY predictor
        # left in its variables for the first two images of this rank's shard must equal the oracle's result for the same bytes
        # (int8 tensors bit for bit, fp32 within the tolerance of the parity tests; predictor 0: every surviving variable, the
        # others: the output); a mismatch makes the run fail instead of printing a rate for garbage
        selfcheck = None
        if details and rank == 0 and not dry and check_images is not None and not args.no_selfcheck:
            selfcheck = oracle_selfcheck(np, engines[0].pred, net, cfg, check_images, engines[0].out_var)
            others = []
            for e_ in engines[1:]:
                if e_ is None:
                    continue
                got = e_.pred.get_var(e_.out_var, np.int8 if cfg["model"] == "conv" else np.float32, max_bytes=1 << 30)[:check_images.shape[0]]
                ref0 = engines[0].pred.get_var(engines[0].out_var, got.dtype, max_bytes=1 << 30)[:check_images.shape[0]]
                others.append(bool(np.array_equal(got, ref0)))  # same program, same bytes in: identical to the verified predictor
            selfcheck["predictors_checked"] = 1 + len(others)
            selfcheck["ok"] = bool(selfcheck["ok"] and all(others))
            if not selfcheck["ok"]:
                sys.stderr.write("bench.py: SELF-CHECK FAILED: %s\n" % json.dumps(selfcheck))
                real_stdout.flush()
                os._exit(3)

        eng0 = engines[0]
        pred = None if dry else eng0.pred
        serial, roof, fam_out = None, None, {}
        if rank == 0 and not dry and details:
            stream = eng0.stream
            # ---- informational: the same steps strictly serial on one stream (what one predictor alone delivers) ----
            if P > 1:
                for _ in range(3):
                    eng0.run()
                torch.cuda.synchronize(dev)
                ts = time.perf_counter()
                for _ in range(args.steps):
                    eng0.run()
                torch.cuda.synchronize(dev)
                es = time.perf_counter() - ts
                serial = {"value": round(rows * args.steps / es, 1), "unit": "img/s", "ms_per_step": round(1e3 * es / args.steps, 4),
                          "note": "one predictor, one stream, steps back to back (this GPU's shard only): a step LATENCY"}

            # ---- per-launch kernel time, live, HIP events on the launch stream.  One event pair brackets INNER back-to-back
            # launches of the same instruction: a pair around nothing already reads ~4.7 us on this stack, so a pair per
            # launch overstated every kernel by ~3 us against rocprofv3's averages; with 4 launches per pair the residue is < 1 us.
            names = pred.kernel_names()
            body = [i for i, n_ in enumerate(names) if not n_.startswith("io_copy")]
            reps, INNER = 5, 4
            with torch.cuda.stream(stream):
                ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in body] for _ in range(reps)]
                for r in range(reps):
                    for j, i in enumerate(body):
                        ev[r][j][0].record(stream)
                        for _ in range(INNER):
                            pred.run_instruction(i)
                        ev[r][j][1].record(stream)
            torch.cuda.synchronize(dev)
            per_inst = {i: float(np.median([ev[r][j][0].elapsed_time(ev[r][j][1]) for r in range(reps)])) / INNER
                        for j, i in enumerate(body)}
            if cfg["model"] == "conv":
                o = net["ops"][0]
                cout, cin, k, _ = o["w"].shape
                costs = [dict(name="image", family="io_copy", ops=0, bytes=0),
                         dict(name="y", family="conv3x3", ops=2 * rows * cout * 56 * 56 * cin * k * k,
                              bytes=rows * (cin * 56 * 56 + cout * 56 * 56) + o["w"].size)]
            else:
                costs = wl.program_costs(net, rows, eng0.plan)
            assert len(costs) == len(names), (len(costs), len(names))
            for i in body:
                f = fam_out.setdefault(costs[i]["family"], {"launches": 0, "ms": 0.0, "alg_bytes": 0, "ops": 0})
                f["launches"] += 1
                f["ms"] += per_inst[i]
                f["alg_bytes"] += costs[i]["bytes"]
                f["ops"] += costs[i]["ops"]
            for f in fam_out.values():
                f["GB/s"] = round(f["alg_bytes"] / f["ms"] / 1e6, 1)
                f["TOP/s"] = round(f["ops"] / f["ms"] / 1e9, 2)
                f["ms"] = round(f["ms"], 4)
            if args.layer_table:
                for i in body:
                    print("%-28s %-16s %8.4f ms  %7.1f GB/s %7.1f TOP/s  %s" % (
                        costs[i]["name"], costs[i]["family"], per_inst[i], costs[i]["bytes"] / per_inst[i] / 1e6,
                        costs[i]["ops"] / per_inst[i] / 1e9, names[i]), file=sys.stderr)
            # the dominant family = the one with the largest summed launch time of a step, nothing else (families within a few
            # per cent of each other may swap between runs: `roofline_by_family` in the line carries all of them)
            dom = max(fam_out, key=lambda k_: fam_out[k_]["ms"])
            d = fam_out[dom]
            ai = d["ops"] / max(1, d["alg_bytes"])
            mfma_bound = ai > BALANCE_OPS_PER_BYTE
            # HBM-side traffic per launch: from the PMC passes committed under profiles/ (tools/pmc_traffic.py over
            # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script: FETCH_SIZE x2 per the gfx950 correction
            # + WRITE_SIZE); a file, not a live counter -> named in traffic_source, null when absent for this config
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, "profiles", {"c3": "pmc_traffic.json", "c4": "pmc_traffic_c4.json", "c2": "pmc_traffic_c2.json"}.get(args.config, "none"))
            if os.path.exists(tpath) and rows == cfg["batch"]:
                try:
                    tj = json.load(open(tpath))
                    if tj.get("csrc_sha256") != csrc_sha256():
                        raise ValueError("kernel sources changed since the PMC pass")
                    t_ = tj.get(dom)
                    if not t_ and dom == "conv3x3" and "conv3x3_patch" in tj:  # one conv = the padded copy + the patch kernel
                        t_ = {k_: tj["conv3x3_patch"][k_] + tj.get("conv3x3_patch_pad", {}).get(k_, 0.0)
                              for k_ in ("fetch_bytes_per_launch_x2", "write_bytes_per_launch")}
                    if t_:
                        traffic = round(t_["fetch_bytes_per_launch_x2"] + t_["write_bytes_per_launch"])
                        tsrc = "profiles/%s @ %s (not measured in this run)" % (os.path.basename(tpath), tj.get("commit", "round 1"))
                except Exception as e_:  # noqa: BLE001
                    traffic, tsrc = None, "profiles/%s refused: %s" % (os.path.basename(tpath), e_)
            roof = {"kernel": dom, "bound": "mfma" if mfma_bound else "hbm",
                    "achieved": d["TOP/s"] if mfma_bound else d["GB/s"], "peak": round(MFMA_I8_PEAK_TOPS, 1) if mfma_bound else HBM_PEAK_GBS,
                    "unit": "TOP/s" if mfma_bound else "GB/s",
                    "frac": round((d["TOP/s"] / MFMA_I8_PEAK_TOPS) if mfma_bound else (d["GB/s"] / HBM_PEAK_GBS), 4),
                    "traffic": traffic, "traffic_source": tsrc,
                    "avg_launch_ms": round(d["ms"] / d["launches"], 5), "launches_per_step": d["launches"],
                    "alg_bytes_per_launch": round(d["alg_bytes"] / d["launches"]), "alg_ops_per_byte": round(ai, 1),
                    "hbm_GB/s": d["GB/s"], "hbm_frac": round(d["GB/s"] / HBM_PEAK_GBS, 4),
                    "mfma_TOP/s": d["TOP/s"], "mfma_frac_of_dense_i8_peak": round(d["TOP/s"] / MFMA_I8_PEAK_TOPS, 4),
                    "note": "family aggregate: algorithmic bytes = unique input + weights + output once per launch (SURVEY.md 8d), "
                            "summed over the family's launches of one step, / summed launch time (HIP events on the launch "
                            "stream, 4 launches per event pair); bound = mfma iff ops/byte > %.0f" % BALANCE_OPS_PER_BYTE}

        cpu = None
        if details and rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
            cpu = cpu_baseline(cfg, net, args.cpu_seconds)

        for f_ in flights:
            f_.alive = False
            f_.cmd.release()
        for e in engines:
            if e is not None:
                e.close()
        return dict(scaling=scaling, global_batch=global_batch, rows=rows, P=P, elapsed=elapsed, wins=wins, serial=serial,
                    roof=roof, fam_out=fam_out, selfcheck=selfcheck, cpu=cpu, net=net)

    head_scaling = args.scaling or cfg.get("scaling", "weak")
    case = run_case(head_scaling, args.global_batch, args.batch, True)
    # a multi-GPU run of the default (weak: fixed images per GPU) config also times the STRONG point the north star names: one
    # large batch (1024 images) split over the GPUs; it goes into the line as `strong`, the headline stays as configured
    strong = None
    if world > 1 and head_scaling == "weak" and args.scaling is None and cfg["model"] != "conv":
        sc_ = run_case("strong", args.global_batch or 1024, None, False)
        strong = {"global_batch": sc_["global_batch"], "images_per_gpu": sc_["rows"], "value": round(sc_["global_batch"] * args.steps / sc_["elapsed"], 1),
                  "unit": "img/s", "ms_per_step": round(1e3 * sc_["elapsed"] / args.steps, 4), "scaling": "strong",
                  "note": "same run, same ranks: global batch fixed and split by shard_range; compare with the 1-GPU line of `--scaling strong --global-batch %d`" % sc_["global_batch"]}
    scaling, global_batch, rows, P, elapsed, wins = (case[k] for k in ("scaling", "global_batch", "rows", "P", "elapsed", "wins"))
    serial, roof, fam_out, selfcheck, cpu, net = (case[k] for k in ("serial", "roof", "fam_out", "selfcheck", "cpu", "net"))
    if rank == 0:
        total_imgs = global_batch * args.steps
        val = total_imgs / elapsed
        stats = wl.net_stats(net) if cfg["model"] != "conv" else {"conv_kxk": 128 * 56 * 56 * 64 * 9}
        ops_per_img = 2 * sum(stats.values())
        line = {
            "metric": cfg["metric"], "value": round(val, 1), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "int8", "data": "synthetic" if not dry else "dryrun-no-compute",
            "config": {"workload": "%s, random-init weights, %d images per step and GPU (%s), input resident in HBM" % (
                cfg["title"], rows, ("%d predictors / HIP streams, each running whole steps: %d steps in flight" % (P, P))
                if P > 1 else "1 predictor, serial steps"),
                       "name": args.config, "steps_in_flight": P, "global_batch": global_batch,
                       "parallelism": "batch split x%d (shard_range), RCCL network broadcast + input scatter at init, "
                                      "asynchronous all_gather of the probabilities per step" % world},
            "ms_per_step_note": "elapsed / steps with %d steps in flight: an aggregate issue interval, not a step latency "
                                "(see single_stream)" % P,
            "whole_graph_TOP/s": round(val * ops_per_img / 1e12, 2),
            "whole_graph_frac_of_i8_mfma_peak": round(val * ops_per_img / 1e12 / MFMA_I8_PEAK_TOPS, 4),
            "windows": {"count": len(wins), "steps_each": args.steps,
                        "ms_per_step_min_median_max": [round(1e3 * min(wins) / args.steps, 4), round(1e3 * elapsed / args.steps, 4),
                                                       round(1e3 * max(wins) / args.steps, 4)]},
            "selfcheck": selfcheck, "strong": strong,
            "debug_knobs": dict(getattr(sys.modules.get("paddle_lite_amd.capi"), "KNOBS_SET", {})), "single_stream": serial, "roofline": roof, "roofline_by_family": by_family(fam_out), "kernels": fam_out, "cpu_baseline": cpu,
        }
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
