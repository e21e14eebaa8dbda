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

    ``named_params``: (name, parameter) in the order the backward produces the gradients (heads first, ASPP last);
    consecutive names are grouped into buckets of at most ``bucket_bytes``. Every bucket owns ONE persistent flat buffer:
    a gradient is scaled by 1/world straight into its slice the moment it exists (one pass), the bucket is all-reduced IN
    PLACE on a side stream as soon as it is complete, and what autograd receives is the slice itself -- no concatenation,
    no division pass, no copy back (round 2 made three passes over the 123 MB of stage 2).

    The sequence of collectives never depends on which gradients a rank produced: buckets are launched strictly in index
    order (a complete bucket waits for its incomplete predecessors until ``backward_done``), every bucket is launched
    every step, and a gradient that did not arrive travels as zeros and still receives the average (it is assigned to
    ``param.grad`` of a parameter that requires grad) -- ranks can therefore never pair collectives of different sizes.

    Contract: every handled parameter's ``.grad`` must be None when its gradient arrives (``zero_grad(set_to_none=True)``,
    what TrainStep does): ``param.grad`` aliases the bucket buffer between steps. Violations raise.
    """

    def __init__(self, named_params, bucket_bytes=64 << 20, group=None, force=False, close_after_bytes=16 << 20):
        """close_after_bytes: a gradient at least this large closes its bucket, i.e. its all-reduce starts the moment it exists
        instead of waiting for the small gradients behind it (stage 2: the three 37.7 MB ASPP weight gradients each end a bucket
        and travel beside the next branch's GEMMs; the last bucket is the 8 MB of the two small branches).
        force: run the collectives even in a one-rank group (exercises the RCCL path on a single GPU; tests).
        MSS_DDP_NO_COMM=1 (bench / tests only): everything but the collective itself -- the exposed-communication probe."""
        import os
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.no_comm = os.environ.get("MSS_DDP_NO_COMM") == "1"
        self.buckets = []       # list of lists of names (fixed at construction: every rank reduces the same layout)
        self.where = {}         # name -> (bucket index, offset, numel)
        self.params = {}        # name -> parameter
        cur, cur_bytes = [], 0
        for name, p in named_params:
            self.params[name] = p
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(name)
            cur_bytes += nbytes
            if close_after_bytes and nbytes >= close_after_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.sizes = []
        for i, b in enumerate(self.buckets):
            off = 0
            for n in b:
                self.where[n] = (i, off, self.params[n].numel())
                off += self.params[n].numel()
            self.sizes.append(off)
        self.flat = [None] * len(self.buckets)          # persistent, allocated on first use
        self.comm_stream = None
        self.bytes_per_step = sum(self.sizes) * 4
        self._reset()

    def _reset(self):
        self.arrived = [set() for _ in self.buckets]
        self.next_bucket = 0
        self.inflight = []      # (bucket index, work)

    def _buffer(self, i):
        if self.flat[i] is None:
            p = self.params[self.buckets[i][0]]
            self.flat[i] = torch.zeros(self.sizes[i], dtype=p.dtype, device=p.device)
        return self.flat[i]

    def __call__(self, name, grad):
        """Returns the tensor to hand to autograd in place of `grad` (a view of the bucket's flat buffer, holding the
        average once backward_done() has returned), or None when this sink does not handle `name`."""
        if not self.active or name not in self.where:
            return None
        i, off, n = self.where[name]
        if self.params[name].grad is not None:
            # autograd steals the returned view as param.grad, so a gradient still defined from an earlier backward IS this
            # buffer: writing the new one would overwrite it and AccumulateGrad would then add the buffer to itself (silently
            # doubled gradients), and an accumulated sum would be all-reduced a second time. Accumulation over several
            # backwards is not what the reference loop does (train_deeplab.py:198-204: zero_grad / backward / step).
            raise RuntimeError(f"GradAllReduce: {name}.grad is still defined at backward time; call "
                               "optimizer.zero_grad(set_to_none=True) before every backward (gradient accumulation and "
                               "zero_grad(set_to_none=False) are not supported with the in-place bucket buffers)")
        view = self._buffer(i)[off:off + n].view(grad.shape)
        torch.mul(grad, 1.0 / self.world, out=view)
        self.arrived[i].add(name)
        while self.next_bucket < len(self.buckets) and len(self.arrived[self.next_bucket]) == len(self.buckets[self.next_bucket]):
            self._launch(self.next_bucket)
            self.next_bucket += 1
        return view

    def _launch(self, i):
        flat = self._buffer(i)
        for n in self.buckets[i]:                     # a gradient that is absent this step travels as zeros
            if n not in self.arrived[i]:
                _, off, cnt = self.where[n]
                flat[off:off + cnt].zero_()
        if self.no_comm:
            self.inflight.append((i, None))
            return
        if flat.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                work = dist.all_reduce(flat, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, group=self.group, async_op=True)
        self.inflight.append((i, work))

    def flush(self):
        """Launch, in index order, every bucket that has not been launched yet (some gradient of it, or of a predecessor,
        did not arrive this step)."""
        while self.next_bucket < len(self.buckets):
            self._launch(self.next_bucket)
            self.next_bucket += 1

    def backward_done(self):
        if not self.active:
            return
        self.flush()
        for i, work in self.inflight:
            if work is not None:
                work.wait()      # CUDA: makes the current stream wait for the collective, no host block
            for n in self.buckets[i]:
                if n not in self.arrived[i]:
                    p = self.params[n]
                    if p.requires_grad:           # no local gradient this step, but the other ranks' average is this rank's too
                        _, off, cnt = self.where[n]
                        p.grad = self.flat[i][off:off + cnt].view(p.shape).clone()
        self._reset()

    def abort(self):
        """The backward is unwinding from an exception: start no further collective (the other ranks may never reach
        theirs), wait for the ones already in flight and forget this step's gradients."""
        for _, work in self.inflight:
            try:
                if work is not None:
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
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if local_world > ndev and os.environ.get("MSS_DIST_BACKEND") != "gloo":
            # a mis-sized launch: ranks would silently share GPUs (local_rank % ndev below) and the "per-GPU" numbers mean nothing
            import warnings
            warnings.warn(f"multishiftseg_amd.ddp: {local_world} local ranks on {ndev} visible GPU(s): ranks share devices "
                          f"(rank {rank} -> cuda:{local_rank % ndev}); launch with --nproc-per-node <= {ndev}", RuntimeWarning, stacklevel=2)
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
