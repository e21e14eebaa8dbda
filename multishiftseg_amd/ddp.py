"""Data parallelism for the training loop: one process per GPU, gradients of the *trainable subset*
averaged with bucketed all-reduces (RCCL over xGMI; backend "nccl" is RCCL on ROCm) launched from
inside the backward as soon as each gradient exists, on a side stream, so they overlap with the
remaining dgrad/wgrad kernels.

The reference only has single-process ``nn.DataParallel`` (train_deeplab.py:91): replicate +
scatter + gather of the full-resolution logits to device 0 every step. Here every rank keeps its own
(original, augmented) pairs -- local batch layout [orig...; aug...] so the loss's ``i <-> i + B/2``
pairing stays rank-local -- and only gradients travel: 19 KB in stage 1, 123 MB in
stage 2. xGMI is point-to-point (7 links x ~153 GB/s per GPU), so buckets are few and large.
"""
import torch
import torch.distributed as dist


class GradAllReduce:
    """Bucketed, overlapped gradient averaging. Use as ``model.grad_sink``.

    ``order``: parameter names in the order the backward produces them (heads first, ASPP last);
    consecutive names are grouped into buckets of at most ``bucket_bytes``. A bucket is flattened
    and all-reduced when its last gradient arrives. ``backward_done`` (called by the model at the
    end of its backward) makes the compute stream wait for the communication stream, then the
    averaged values are copied back into the gradient tensors autograd is about to hand out.
    """

    def __init__(self, named_params, bucket_bytes=64 << 20, group=None, force=False):
        """force: run the collectives even in a one-rank group (exercises the RCCL path on a single GPU; tests)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.buckets = []       # list of lists of names (fixed at construction: every rank reduces the same layout)
        self.where = {}         # name -> bucket index
        self.shapes = {}        # name -> (shape, dtype, device) for zero-filling a gradient that did not arrive
        cur, cur_bytes = [], 0
        for name, p in named_params:
            self.shapes[name] = (tuple(p.shape), p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(name)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        for i, b in enumerate(self.buckets):
            for n in b:
                self.where[n] = i
        self.comm_stream = None
        self._reset()

    def _reset(self):
        self.pending = [dict() for _ in self.buckets]
        self.inflight = []      # (flat, [(name, tensor)], work)

    def __call__(self, name, grad):
        if not self.active or name not in self.where:
            return
        i = self.where[name]
        self.pending[i][name] = grad
        if len(self.pending[i]) == len(self.buckets[i]):
            self._launch(i)

    def _launch(self, i):
        # a gradient that is absent this step (parameter frozen after construction, unused branch) travels as zeros
        # so that the message layout never depends on which gradients a rank happened to produce
        items = []
        for n in self.buckets[i]:
            g = self.pending[i].get(n)
            if g is None:
                shape, dtype, dev = self.shapes[n]
                g = torch.zeros(shape, dtype=dtype, device=dev)
                items.append((n, g, False))
            else:
                items.append((n, g, True))
        self.pending[i] = {}
        if items[0][1].is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()
            flat = torch.cat([g.reshape(-1) for _, g, _ in items])
            flat.div_(self.world)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                work = dist.all_reduce(flat, group=self.group, async_op=True)
            flat.record_stream(self.comm_stream)
        else:
            flat = torch.cat([g.reshape(-1) for _, g, _ in items])
            flat.div_(self.world)
            work = dist.all_reduce(flat, group=self.group, async_op=True)
        self.inflight.append((flat, items, work))

    def flush(self):
        """Launch buckets that never filled (some of their parameters produced no gradient this step). The bucket
        layout itself is never changed."""
        for i, pend in enumerate(self.pending):
            if pend:
                self._launch(i)

    def backward_done(self):
        if not self.active:
            return
        self.flush()
        for flat, items, work in self.inflight:
            work.wait()          # CUDA: makes the current stream wait for the collective, no host block
            off = 0
            for _, g, present in items:
                n = g.numel()
                if present:
                    g.copy_(flat[off:off + n].view_as(g))
                off += n
        self._reset()

    def abort(self):
        """The backward is unwinding from an exception: start no further collective (the other ranks may never reach
        theirs), wait for the ones already in flight and forget this step's gradients."""
        for _, _, work in self.inflight:
            try:
                work.wait()
            except Exception:
                pass
        self._reset()


def init_from_env():
    """(rank, world, local_rank, device) from the torchrun environment; initialises the default
    process group when WORLD_SIZE > 1 (backend nccl = RCCL on a GPU box, gloo otherwise)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % ndev)
        device = torch.device("cuda", local_rank % ndev)
        backend = "nccl"
    else:
        device = torch.device("cpu")
        backend = "gloo"
    backend = os.environ.get("MSS_DIST_BACKEND", backend)   # tests: gloo with several ranks on one GPU
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank, device


def shard_pairs(num_pairs, rank, world):
    """Pairs owned by `rank`: {k : k mod world == rank}."""
    return list(range(rank, num_pairs, world))
