"""Batch-split data parallelism for inference: one process per GPU, full weight replica per GPU.

  * once at init: rank 0's packed int8 weights + fp32 scales/biases are broadcast (RCCL over xGMI on GPUs; gloo in
    the CPU tests) — MobileNetV1 is ~4.3 MB, latency-bound, not on the steady-state path;
  * per step: every rank runs images [r*B/G, (r+1)*B/G) with no collective inside the layer loop, then the
    [B/G, 1000] fp32 probabilities are all-gathered.
The reference has no multi-device code at all; this module is new.
"""
import numpy as np


def pack_weights(W):
    """Flatten {name: array | {field: array}} into (manifest, uint8 blob); every entry 16-byte aligned."""
    items, blobs, off = [], [], 0
    for name in sorted(W):
        v = W[name]
        fields = sorted(v.items()) if isinstance(v, dict) else [("", v)]
        for f, arr in fields:
            arr = np.asarray(arr)
            shape = tuple(arr.shape)  # ascontiguousarray would promote 0-d scalars to (1,)
            b = np.ascontiguousarray(arr).tobytes()
            items.append((name, f, str(arr.dtype), shape, off, len(b)))
            blobs.append(b)
            off += len(b)
            pad = (-off) % 16
            blobs.append(b"\0" * pad)
            off += pad
    return items, np.frombuffer(b"".join(blobs), np.uint8).copy()


def unpack_weights(items, blob):
    W = {}
    for name, f, dt, shape, off, n in items:
        arr = np.frombuffer(blob[off:off + n].tobytes(), dtype=np.dtype(dt)).reshape(shape).copy()
        if arr.shape == ():
            arr = arr.dtype.type(arr)
        if f == "":
            W[name] = arr
        else:
            W.setdefault(name, {})[f] = arr
    return W


def shard_range(global_batch, rank, world):
    """Images [lo, hi) of rank `rank`: contiguous, sizes differ by at most one (ragged batches allowed)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(W, dist, device, rank, world, src=0, force=False):
    """Rank `src` passes its weight dict, the others None; everyone returns the same dict.
    `dist` is torch.distributed (initialised); `device` the torch device of the transport buffer.
    force: run the collectives even with one rank (exercises the transport on a one-GPU box)."""
    import torch
    if world == 1 and not force:
        return W
    if rank == src:
        items, blob = pack_weights(W)
        meta = [items, int(blob.size)]
    else:
        items, blob, meta = None, None, [None, None]
    dist.broadcast_object_list(meta, src=src)
    items, nbytes = meta
    t = torch.from_numpy(blob).to(device) if rank == src else torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    if rank == src and not force:
        return W
    return unpack_weights(items, t.cpu().numpy())


def all_gather_rows(local, dist, world):
    """all_gather of equally sized [rows, cols] tensors into [world*rows, cols] (rank-major order)."""
    import torch
    if world == 1:
        return local
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


class PipelinedGather:
    """Result gather that never stalls the compute stream: step s stages its shard into buffer s % depth and starts an
    asynchronous all_gather (RCCL runs it on its own stream, ordered after the staging copy); the buffer is only waited
    for when it comes round again `depth` steps later, so the collective of step s overlaps the kernels of steps
    s+1 .. s+depth-1.  `local_like` gives the shard shape / dtype / device."""

    def __init__(self, local_like, dist, world, depth=2):
        import torch
        self.dist, self.world, self.depth = dist, world, depth
        shape = tuple(local_like.shape)
        self.local = [torch.empty_like(local_like) for _ in range(depth)]
        self.out = [torch.empty((world * shape[0],) + shape[1:], dtype=local_like.dtype, device=local_like.device)
                    for _ in range(depth)]
        self.pending = [None] * depth
        self.step = 0

    def stage_buffer(self):
        """Shard buffer of the current step (free: its previous collective has been waited for)."""
        b = self.step % self.depth
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        return self.local[b]

    def launch(self):
        """Start the all_gather of the current step's staged shard; returns the [world*rows, ...] output tensor, which
        is complete after drain() (or after this buffer's next stage_buffer())."""
        b = self.step % self.depth
        if self.world > 1:
            self.pending[b] = self.dist.all_gather_into_tensor(self.out[b], self.local[b], async_op=True)
        else:
            self.out[b].copy_(self.local[b])
        self.step += 1
        return self.out[b]

    def drain(self):
        for b in range(self.depth):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None



# ---- op-list networks (workloads.py) over the same transport ----
def net_to_blob(net):
    """Split an op-list network into a picklable skeleton (no arrays) and an {key: array} dict for pack_weights."""
    arrays, ops = {}, []
    for i, o in enumerate(net["ops"]):
        sk = {}
        for k, v in o.items():
            if isinstance(v, np.ndarray):
                arrays.setdefault("op%04d" % i, {})[k] = v
                sk[k] = None
            elif isinstance(v, np.generic):
                sk[k] = ("np", str(v.dtype), v.item())
            else:
                sk[k] = v
        ops.append(sk)
    skeleton = {k: v for k, v in net.items() if k != "ops"}
    skeleton["ops"] = ops
    return skeleton, arrays


def net_from_blob(skeleton, arrays):
    net = {k: v for k, v in skeleton.items() if k != "ops"}
    ops = []
    for i, sk in enumerate(skeleton["ops"]):
        o = {}
        for k, v in sk.items():
            if v is None and k in arrays.get("op%04d" % i, {}):
                o[k] = arrays["op%04d" % i][k]
            elif isinstance(v, tuple) and len(v) == 3 and v[0] == "np":
                o[k] = np.dtype(v[1]).type(v[2])
            else:
                o[k] = v
        ops.append(o)
    net["ops"] = ops
    return net


def broadcast_net(net, dist, device, rank, world, src=0, force=False):
    """Rank `src` passes the network, the others None: skeleton by broadcast_object_list, every weight / scale / bias
    array in ONE blob by dist.broadcast (RCCL over xGMI on GPUs)."""
    if world == 1 and not force:
        return net
    if rank == src:
        skeleton, arrays = net_to_blob(net)
        meta = [skeleton]
    else:
        arrays, meta = None, [None]
    dist.broadcast_object_list(meta, src=src)
    arrays = broadcast_weights(arrays, dist, device, rank, world, src, force=force)
    if rank == src and not force:
        return net
    return net_from_blob(meta[0], arrays)


def scatter_batch(images, global_batch, sample_shape, dist, device, rank, world, src=0, force=False):
    """One global batch generated on rank `src` -> every rank's contiguous shard [lo, hi) (shard_range; ragged sizes
    allowed: shards are padded to the largest one for the collective).  Returns a numpy array [hi - lo, ...]."""
    import torch
    lo, hi = shard_range(global_batch, rank, world)
    if world == 1 and not force:
        return np.ascontiguousarray(images[lo:hi])
    rows = max(shard_range(global_batch, r, world)[1] - shard_range(global_batch, r, world)[0] for r in range(world))
    recv = torch.empty((rows,) + tuple(sample_shape), dtype=torch.float32, device=device)
    parts = None
    if rank == src:
        parts = []
        for r in range(world):
            l, h = shard_range(global_batch, r, world)
            t = torch.zeros((rows,) + tuple(sample_shape), dtype=torch.float32)
            t[:h - l] = torch.from_numpy(np.ascontiguousarray(images[l:h]))
            parts.append(t.to(device))
    dist.scatter(recv, parts, src=src)
    return recv[:hi - lo].cpu().numpy()
