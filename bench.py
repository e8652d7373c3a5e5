#!/usr/bin/env python3
"""bench.py — INT8 images/sec of the MobileNetV1 224x224 graph on N MI355X GPUs (BASELINE.json metric).

A step = one pass of the hot path (calib -> 27 int8 convs -> pool -> calib -> fc -> softmax, the program of
SURVEY.md Appendix D) over one synthetic batch (128 images) per GPU, through the C++ kHIP kernel classes and libplhip.so.
By default 3 predictors per GPU (one host thread + HIP stream each) run whole batches, the timed K steps being dealt
round-robin, so three steps are in flight at once and fill each other's dispatch gaps, launch ramps and tails
(`--inflight 1`: strictly serial; the serial rate is also reported as `single_stream`).
Inputs are resident in HBM when the timed region starts (the host->device io_copy instruction is skipped).
N > 1: one process per GPU (torch.distributed, backend "nccl" == RCCL): rank 0's weights are broadcast over xGMI
at init, every rank runs its own batch shard (no collective inside the layer loop) and the logits are all-gathered
each step.  Weak scaling: the per-GPU batch is fixed.

The JSON line carries `roofline` for the dominant kernel family (time measured live with HIP events on the launch
stream, one event pair per launch) and `cpu_baseline` (the oracle's im2col+GEMM port of the reference algorithm,
OpenMP on the host cores, bounded sample; rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Denominators (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8.0 TB/s spec; int8 MFMA dense =
# 2x the bf16 rate = 256 CU x 4 SIMD x 2048 op/clk x 2.4 GHz = 5.03 POP/s (sparsity figures not used).
HBM_PEAK_GBS = 8000.0
MFMA_I8_PEAK_TOPS = 256 * 4 * 2048 * 2.4e9 / 1e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)   # ~0.13 s timed at 0.42 ms / step: steadier than a 20 ms window
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (BASELINE config: batch=128 on 1 GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--layer-table", action="store_true", help="print the per-layer timing table to stderr")
    ap.add_argument("--inflight", type=int, default=3,
                    help="predictors per GPU, each on its own host thread + HIP stream, each running WHOLE batches of "
                         "--batch images; the steps are dealt round-robin, so --inflight steps are in flight at once "
                         "(the reference's one-predictor-per-thread serving model, cxx_api.h:103-137). Kernels of "
                         "different steps overlap: dispatch gaps, launch ramps and tails of one step are filled by the "
                         "others. 1 = strictly serial steps.")
    ap.add_argument("--streams", type=int, default=1,
                    help="predictors per GPU, each on its own host thread + HIP stream with batch/streams images "
                         "(the reference's one-predictor-per-thread model, cxx_api.h:103-137); kernels of different "
                         "streams overlap, hiding per-launch fill/drain")
    return ap.parse_args()


def layer_costs(wl, batch):
    """Algorithmic ops / bytes per instruction family for one step (unique in + weights + out once)."""
    fam = {}
    for (name, op, cin, cout, k, s, p, g, hin) in wl.mobilenet_v1_layers():
        ho = (hin + 2 * p - k) // s + 1
        macs = batch * ho * ho * cout * (cin // g) * k * k
        out_b = 4 if name == "pw14" else 1
        byts = batch * (cin * hin * hin + cout * ho * ho * out_b) + cout * (cin // g) * k * k
        key = "conv3x3s2_first" if name == "conv1" else ("depthwise3x3" if g > 1 else "pointwise1x1")
        fam.setdefault(key, {"ops": 0, "bytes": 0, "names": []})
        fam[key]["ops"] += 2 * macs
        fam[key]["bytes"] += byts
        fam[key]["names"].append(name)
    return fam


def cpu_baseline(wl, W, seconds):
    """The reference algorithm restated for the host (oracle/plref.c): im2col + int8 GEMM over (batch, group) for the
    dense convs (conv_impl.cc:490-598 structure), direct loops for depthwise, fused float epilogue; OpenMP."""
    from oracle import plref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    rng = np.random.default_rng(99)
    layers = wl.mobilenet_v1_layers()
    done, t0 = 0, time.perf_counter()
    while True:
        x = plref.calib_f32_to_i8(rng.uniform(-1, 1, (1, 3, 224, 224)).astype(np.float32), float(W["input_scale"]))
        for i, (name, op, cin, cout, k, s, p, g, hin) in enumerate(layers):
            L = W[name]
            sh = plref.shape(1, cin, x.shape[2], x.shape[3], cout, k, k, (p, p, p, p), (s, s), (1, 1), g)
            x, _ = plref.conv2d(sh, x, L["w"], L["bias"], float(L["in_scale"]), L["w_scale"], float(L["out_scale"]), 1, 0.0,
                                i != len(layers) - 1, via_gemm=(g == 1))
        pool = plref.global_avg_pool(x)
        q = plref.calib_f32_to_i8(pool, float(W["pool_scale"]))
        F = W["fc"]
        logits, _ = plref.fc(q.reshape(1, -1), F["w"], F["bias"], (F["w_scale"] * np.float32(F["in_scale"])).astype(np.float32), False, False)
        plref.softmax(logits)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 100000:  # bounded by time (default 12 s of CPU work)
            break
    return {"value": round(done / el, 2), "unit": "img/s", "cores": int(os.environ["OMP_NUM_THREADS"]),
            "kind": "port",
            "sample": "%d images of the same MobileNetV1-INT8 graph, batch 1 each, %.1f s; oracle/plref.c restatement of the "
                      "reference's im2col+GEMM int8 path (its ARM NEON kernels cannot run on x86; its x86 backend has no "
                      "INT8 kernels)" % (done, el)}


def main():
    args = parse()
    # the CPU baseline's OpenMP team = the cores this process may actually use (not every core of the host)
    # (capped at 32: the per-layer GEMMs of one image are too small to feed more threads — 256 threads measured slower)
    os.environ.setdefault("OMP_NUM_THREADS", str(min(32, len(os.sched_getaffinity(0)))))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.import_package()
    lite = importlib.import_module("paddle_lite_amd.liteapi")
    wl = importlib.import_module("paddle_lite_amd.workloads")
    sharding = importlib.import_module("paddle_lite_amd.sharding")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box (never used by the driver): PLHIP_BENCH_SAME_GPU=1 puts every rank on device 0 and
    # PLHIP_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so that the N > 1 control flow
    # (weight broadcast, staging copies, pipelined gather, max-over-ranks timing) can be exercised end to end.
    if os.environ.get("PLHIP_BENCH_SAME_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("PLHIP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- weights: rank 0 generates, RCCL broadcast over xGMI, every rank builds its predictor from the bytes ----
    W = wl.make_mobilenet_v1_weights(seed=1234) if rank == 0 else None
    W = sharding.broadcast_weights(W, dist, dev, rank, world)

    import queue
    import threading
    P = max(1, args.inflight)
    S = max(1, args.streams)
    if S > 1:
        P = 1  # --streams (batch split inside one step) and --inflight (whole batches) are alternatives
    assert args.batch % S == 0, "--batch must be divisible by --streams"
    sub = args.batch // S
    main_stream = torch.cuda.current_stream(dev)
    streams = [main_stream] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    preds = [None] * S
    rng = np.random.default_rng(1000 + rank)
    images = [rng.uniform(-1, 1, (sub, 3, 224, 224)).astype(np.float32) for _ in range(S)]

    p_bytes = (args.batch if P > 1 else sub) * wl.NUM_CLASSES * 4
    # N > 1: per-step result gather, double buffered so that the RCCL all_gather of step s overlaps step s+1
    gather = (sharding.PipelinedGather(torch.empty((args.batch, wl.NUM_CLASSES), dtype=torch.float32, device=dev), dist, world)
              if world > 1 else None)
    loc_ptr = [0]  # device address of the current step's staging buffer (written by run_steps, read by the workers)
    loc_ready = [None]  # event on the main stream: the collective that last read that buffer has been waited for

    class Worker(threading.Thread):
        """One predictor, one host thread, one HIP stream (TargetWrapperHip state is per thread)."""

        def __init__(self, i):
            super().__init__(daemon=True)
            self.i, self.cmd, self.done, self.err = i, threading.Semaphore(0), threading.Semaphore(0), None
            self.step_done = threading.Semaphore(0)
            self.go = threading.Semaphore(0)
            self.events = []
            self.n = 0
            self.alive = True

        def run(self):
            try:
                torch.cuda.set_device(local_rank)
                p = lite.Predictor(local_rank, stream=streams[self.i].cuda_stream)
                wl.build_mobilenet_v1(p, W, sub)
                p.set_input("image", images[self.i])
                p.run(skip_io_copy=False)  # uploads the feed; PrepareForRun (weight pack, scale fold) everywhere
                p.sync()
                preds[self.i] = p
            except Exception as e:  # noqa: BLE001
                self.err = e
            self.done.release()
            while True:
                self.cmd.acquire()
                if not self.alive:
                    break
                try:
                    for s_ in range(self.n):
                        preds[self.i].run(skip_io_copy=True)
                        if world > 1:  # stage this shard's probabilities for the all_gather of step s_
                            self.go.acquire()  # the main thread has picked this step's staging buffer
                            streams[self.i].wait_event(loc_ready[0])  # ... and its previous collective is done
                            preds[self.i].copy_var_to_device("prob", loc_ptr[0] + self.i * p_bytes, p_bytes)
                            self.events[s_].record(streams[self.i])
                            self.step_done.release()
                except Exception as e:  # noqa: BLE001
                    self.err = e
                self.done.release()

    class FlightWorker(threading.Thread):
        """--inflight: predictor i runs the steps i, i+P, i+2P, ... as whole batches on its own stream."""

        def __init__(self, i):
            super().__init__(daemon=True)
            self.i, self.cmd, self.done, self.err = i, threading.Semaphore(0), threading.Semaphore(0), None
            self.ptrq, self.doneq = queue.Queue(), queue.Queue()
            self.stream = torch.cuda.Stream(dev)
            self.n = 0
            self.alive = True
            self.pred = None

        def run(self):
            try:
                torch.cuda.set_device(local_rank)
                p = lite.Predictor(local_rank, stream=self.stream.cuda_stream)
                wl.build_mobilenet_v1(p, W, args.batch)
                p.set_input("image", images[0])
                p.run(skip_io_copy=False)
                p.sync()
                self.pred = p
            except Exception as e:  # noqa: BLE001
                self.err = e
            self.done.release()
            while True:
                self.cmd.acquire()
                if not self.alive:
                    break
                try:
                    for s_ in range(self.i, self.n, P):
                        self.pred.run(skip_io_copy=True)
                        if world > 1:
                            # stage this step's probabilities into the buffer the coordinator picked for step s_
                            ptr, ready = self.ptrq.get()
                            self.stream.wait_event(ready)  # the collective that last read this buffer has finished
                            self.pred.copy_var_to_device("prob", ptr, p_bytes)
                            ev = torch.cuda.Event()
                            ev.record(self.stream)
                            self.doneq.put(ev)
                except Exception as e:  # noqa: BLE001
                    self.err = e
                self.done.release()

    flights = [FlightWorker(i) for i in range(P)] if P > 1 else []
    for f_ in flights:
        f_.start()
    workers = [Worker(i) for i in range(1, S)]
    for w_ in workers:
        w_.start()
    pred = lite.Predictor(local_rank, stream=main_stream.cuda_stream)
    out_name = wl.build_mobilenet_v1(pred, W, sub)
    pred.set_input("image", images[0])
    pred.run(skip_io_copy=False)
    pred.sync()
    preds[0] = pred
    for w_ in workers + flights:
        w_.done.acquire()
        if w_.err:
            raise w_.err
    n_inst = pred.num_instructions()
    names = pred.kernel_names()
    io_idx = [i for i, n in enumerate(names) if n.startswith("io_copy")]
    body = [i for i in range(n_inst) if i not in io_idx]

    def run_steps_inflight(n):
        """n steps in total, dealt round-robin over the P predictors; this thread only coordinates.  N > 1: the
        collectives are issued here, in step order (every rank issues them in the same order), on the main stream, which
        carries no compute: the RCCL all_gather of step s overlaps the kernels of the following steps."""
        for f_ in flights:
            f_.n = n
            f_.cmd.release()
        if world > 1:
            tr = [0.0, 0.0, 0.0] if os.environ.get("PLHIP_BENCH_TRACE") else None
            for s_ in range(n):
                f_ = flights[s_ % P]
                t_a = time.perf_counter()
                buf = gather.stage_buffer()  # main stream is now ordered behind the collective that last read `buf` ...
                ready = torch.cuda.Event()
                ready.record(main_stream)    # ... and the predictor's stream will be, through this event
                f_.ptrq.put((buf.data_ptr(), ready))
                t_b = time.perf_counter()
                while True:
                    try:
                        ev = f_.doneq.get(timeout=1.0)
                        break
                    except queue.Empty:
                        if f_.err:
                            raise f_.err
                t_c = time.perf_counter()
                main_stream.wait_event(ev)
                gather.launch()
                if tr:
                    tr[0] += t_b - t_a
                    tr[1] += t_c - t_b
                    tr[2] += time.perf_counter() - t_c
            if tr:
                print("rank %d coordinator: %d steps, stage_buffer %.1f ms, wait for predictor %.1f ms, launch %.1f ms" % (
                    rank, n, 1e3 * tr[0], 1e3 * tr[1], 1e3 * tr[2]), file=sys.stderr)
        for f_ in flights:
            f_.done.acquire()
            if f_.err:
                raise f_.err
        if world > 1:
            gather.drain()

    def run_steps(n):
        """n steps on every stream of this GPU; the other predictors run on their own host threads."""
        if P > 1:
            return run_steps_inflight(n)
        for w_ in workers:
            w_.n = n
            w_.events = [torch.cuda.Event() for _ in range(n)] if world > 1 else []
            w_.cmd.release()
        for s_ in range(n):
            pred.run(skip_io_copy=True)
            if world > 1:
                # result gather over xGMI: every predictor stages its [sub, 1000] probabilities into this step's buffer
                # (device-to-device, on its own stream); the main stream waits for them and starts the asynchronous RCCL
                # all_gather, which overlaps the next step (the buffer is waited for two steps later)
                loc_ptr[0] = gather.stage_buffer().data_ptr()
                loc_ready[0] = torch.cuda.Event()
                loc_ready[0].record(main_stream)
                for w_ in workers:
                    w_.go.release()
                pred.copy_var_to_device("prob", loc_ptr[0], p_bytes)
                for w_ in workers:
                    w_.step_done.acquire()
                    main_stream.wait_event(w_.events[s_])
                gather.launch()
        for w_ in workers:
            w_.done.acquire()
            if w_.err:
                raise w_.err
        if world > 1:
            gather.drain()  # every step's result is complete inside the timed region

    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stream = main_stream

    # ---- informational: the same steps strictly serial on one stream (what one predictor alone delivers) ----
    serial = None
    if rank == 0 and P > 1:
        for _ in range(3):
            pred.run(skip_io_copy=True)
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(args.steps):
            pred.run(skip_io_copy=True)
        torch.cuda.synchronize(dev)
        es = time.perf_counter() - ts
        serial = {"value": round(args.batch * args.steps / es, 1), "unit": "img/s", "ms_per_step": round(1e3 * es / args.steps, 4),
                  "note": "one predictor, one stream, steps back to back (this GPU only)"}

    # ---- per-launch kernel time, live, HIP events on the launch stream (rank 0) ----
    roof, fam_out = None, {}
    if rank == 0:
        # One event pair brackets INNER back-to-back launches of the same instruction: a pair around nothing already
        # reads ~4.7 us on this stack, so a pair per launch overstated every kernel by ~3 us (rocprofv3's per-kernel
        # averages of the serial run were 18.3 us vs 21.6 us here); with 4 launches per pair the residue is < 1 us.
        reps, INNER = 5, 4
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
        costs = layer_costs(wl, sub)
        layer_names = [l[0] for l in wl.mobilenet_v1_layers()]
        conv_idx = {}
        for i in body:
            if names[i].startswith("conv2d") or names[i].startswith("depthwise_conv2d"):
                conv_idx[layer_names[len(conv_idx)]] = i
        for key, c in costs.items():
            ms = sum(per_inst[conv_idx[n]] for n in c["names"])
            fam_out[key] = {"launches": len(c["names"]), "ms": round(ms, 4), "GB/s": round(c["bytes"] / ms / 1e6, 1),
                            "TOP/s": round(c["ops"] / ms / 1e9, 2), "alg_bytes": c["bytes"], "ops": c["ops"]}
        if args.layer_table:
            for n in layer_names:
                print("%-6s %8.4f ms  %s" % (n, per_inst[conv_idx[n]], names[conv_idx[n]]), file=sys.stderr)
            other = sum(v for i, v in per_inst.items() if i not in conv_idx.values())
            print("other (calib/pool/fc/softmax) %.4f ms" % other, file=sys.stderr)
        # the dominant family; pointwise and depthwise are within a few % of each other, so the MFMA GEMM family keeps the
        # label unless another one is clearly (> 10 %) larger -- otherwise `roofline.kernel` would flip from run to run
        dom = max(fam_out, key=lambda k: fam_out[k]["ms"])
        if dom != "pointwise1x1" and fam_out[dom]["ms"] < 1.10 * fam_out["pointwise1x1"]["ms"]:
            dom = "pointwise1x1"
        d = fam_out[dom]
        hbm_frac = d["GB/s"] / HBM_PEAK_GBS
        # HBM-side traffic per launch from the committed PMC passes (tools/pmc_traffic.py over `rocprofv3 --pmc FETCH_SIZE`
        # and `--pmc WRITE_SIZE` runs of this script, batch 128): FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B;
        # verified here on calib_f32_to_i8 and on the 32->64 pointwise layer, both = 0.50 of the known bytes) + WRITE_SIZE
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and args.batch == 128:
            try:
                t_ = json.load(open(tpath)).get(dom)
                traffic = round(t_["fetch_bytes_per_launch_x2"] + t_["write_bytes_per_launch"]) if t_ else None
            except Exception:
                traffic = None
        roof = {"kernel": dom, "bound": "hbm", "achieved": d["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(hbm_frac, 4), "traffic": traffic,
                "avg_launch_ms": round(d["ms"] / d["launches"], 5), "launches_per_step": d["launches"],
                "alg_bytes_per_launch": round(d["alg_bytes"] / d["launches"]),
                "mfma_TOP/s": d["TOP/s"], "mfma_frac_of_dense_i8_peak": round(d["TOP/s"] / MFMA_I8_PEAK_TOPS, 4),
                "note": "algorithmic bytes = int8 in + out + weights once per layer (SURVEY.md 8d), summed over the "
                        "family's launches of one step, / summed launch time (HIP events on the launch stream, 4 launches per event pair)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl, W, args.cpu_seconds)  # OMP_NUM_THREADS was pinned to the usable cores in main()

    if rank == 0:
        total_imgs = world * args.batch * args.steps
        val = total_imgs / elapsed
        ops_per_img = 2 * (sum(v for k, v in wl.mobilenet_v1_macs().items() if k != "act_bytes"))
        line = {
            "metric": "INT8 images/sec MobileNetV1 224x224", "value": round(val, 1), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": "MobileNetV1 INT8 full graph 224x224 (27 int8 convs + pool + fc + softmax), "
                                   "random-init weights, batch %d per step and GPU (%s), input resident in HBM" % (
                                       args.batch, ("%d predictors / HIP streams, each running whole batches: %d steps in flight" % (P, P))
                                       if P > 1 else ("%d predictor thread(s)/stream(s) x %d images" % (S, sub))),
                       "steps_in_flight": P,
                       "global_batch": world * args.batch, "parallelism": "batch-split x%d, RCCL weight broadcast + logits all_gather" % world},
            "whole_graph_TOP/s": round(val * ops_per_img / 1e12, 2),
            "whole_graph_frac_of_i8_mfma_peak": round(val * ops_per_img / 1e12 / MFMA_I8_PEAK_TOPS, 4),
            "single_stream": serial, "roofline": roof, "kernels": fam_out, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    for w_ in workers + flights:
        w_.alive = False
        w_.cmd.release()
    pred.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
