"""world_size-2 gloo tests of the data-parallel host logic (runs on CPU): bucketing, in-backward
launch order, averaging, and the pair sharding that keeps the loss's i <-> i+B/2 pairing rank-local."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from multishiftseg_amd import ddp
    r, w, lr, dev = ddp.init_from_env()
    assert (r, w) == (rank, world) and dev.type == "cpu"
    torch.manual_seed(0)
    shapes = {"ood_head.weight": (19, 256, 1, 1), "bot_fine.weight": (48, 128, 1, 1), "aspp.features.1.0.weight": (8, 16, 3, 3),
              "aspp.features.1.1.bias": (8,)}
    params = [(n, torch.zeros(s, requires_grad=True)) for n, s in shapes.items()]
    sync = ddp.GradAllReduce(params, bucket_bytes=30000)          # forces several buckets
    assert len(sync.buckets) >= 2
    grads = {n: torch.full(s, float(rank + 1)) + torch.arange(int(torch.tensor(s).prod())).reshape(s) for n, s in shapes.items()}
    expect = {n: (sum(torch.full(s, float(k + 1)) for k in range(world)) / world
                  + torch.arange(int(torch.tensor(s).prod())).reshape(s)) for n, s in shapes.items()}
    # the sink hands back what autograd should receive: the gradient's slice of the bucket's persistent flat buffer,
    # holding the average once backward_done() has returned
    got = {n: sync(n, grads[n]) for n in shapes}                  # arrival order = backward order
    sync.backward_done()
    ok = all(torch.allclose(got[n], expect[n]) for n in shapes)
    ok = ok and all(got[n].data_ptr() != grads[n].data_ptr() for n in shapes)
    flat_ptrs = [f.data_ptr() for f in sync.flat]
    # second step reuses the object AND its buffers
    got = {n: sync(n, torch.full(shapes[n], float(rank))) for n in shapes}
    sync.backward_done()
    ok = ok and all(torch.allclose(got[n], torch.full_like(got[n], (world - 1) / 2)) for n in shapes)
    ok = ok and flat_ptrs == [f.data_ptr() for f in sync.flat]
    # a parameter that never arrives (frozen later) must not dead-lock the bucket
    sync2 = ddp.GradAllReduce(params, bucket_bytes=1 << 30)
    g = sync2("ood_head.weight", torch.full((19, 256, 1, 1), float(rank)))
    sync2.backward_done()
    ok = ok and torch.allclose(g, torch.full_like(g, (world - 1) / 2))
    # a gradient produced by ONE rank only: same message sizes on both ranks (absent = zeros), and the rank without a local
    # gradient still receives the average (as param.grad)
    layout = [list(b) for b in sync2.buckets]
    pd = dict(params)
    pd["bot_fine.weight"].grad = None
    g1 = sync2("ood_head.weight", torch.full((19, 256, 1, 1), float(rank + 1)))
    g2 = sync2("bot_fine.weight", torch.full((48, 128, 1, 1), 10.0)) if rank == 0 else None
    sync2.backward_done()
    ok = ok and sync2.buckets == layout
    ok = ok and torch.allclose(g1, torch.full_like(g1, (1 + world) / 2))
    if rank == 0:
        ok = ok and torch.allclose(g2, torch.full_like(g2, 10.0 / world))
    else:
        ok = ok and pd["bot_fine.weight"].grad is not None and torch.allclose(pd["bot_fine.weight"].grad, torch.full((48, 128, 1, 1), 10.0 / world))
    for p_ in pd.values():
        p_.grad = None                                             # zero_grad(set_to_none=True), as every step does
    # ranks that produce their gradients in DIFFERENT orders (rank 1 delivers the last bucket first) still issue the
    # collectives in bucket order: no hang, no mis-paired sizes
    sync3 = ddp.GradAllReduce(params, bucket_bytes=30000)
    order = list(shapes) if rank == 0 else list(reversed(list(shapes)))
    got = {n: sync3(n, torch.full(shapes[n], float(rank + 1))) for n in order}
    sync3.backward_done()
    ok = ok and all(torch.allclose(got[n], torch.full_like(got[n], (1 + world) / 2)) for n in shapes)
    # an exception during the backward starts no collective and leaves the object reusable
    g3 = torch.full((19, 256, 1, 1), 7.0)
    sync2("ood_head.weight", g3)
    sync2.abort()
    ok = ok and not any(sync2.arrived) and sync2.next_bucket == 0 and not sync2.inflight and torch.allclose(g3, torch.full_like(g3, 7.0))
    for p_ in pd.values():
        p_.grad = None
    got = {n: sync2(n, torch.full(shapes[n], float(rank))) for n in shapes}
    sync2.backward_done()
    ok = ok and all(torch.allclose(got[n], torch.full_like(got[n], (world - 1) / 2)) for n in shapes)
    # param.grad aliases the bucket buffer between steps, so a second backward without zero_grad(set_to_none=True)
    # would overwrite the live gradient and then add the buffer to itself -- it must raise instead of doubling silently
    w = torch.nn.Parameter(torch.ones(4, 4))
    sync4 = ddp.GradAllReduce([("w", w)])

    class _Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.sum() * 2.5
        @staticmethod
        def backward(ctx, g):
            r = sync4("w", torch.full((4, 4), 2.5) * g)
            return r
    _Node.apply(w).backward()
    sync4.backward_done()
    ok = ok and torch.allclose(w.grad, torch.full((4, 4), 2.5)) and w.grad.data_ptr() == sync4.flat[0].data_ptr()
    raised = False
    try:
        _Node.apply(w).backward()                                  # no zero_grad in between
    except RuntimeError as e:
        raised = "zero_grad(set_to_none=True)" in str(e)
    sync4.abort()
    ok = ok and raised and torch.allclose(w.grad, torch.full((4, 4), 2.5))      # and the live gradient is untouched
    w.grad = None
    _Node.apply(w).backward()
    sync4.backward_done()
    ok = ok and torch.allclose(w.grad, torch.full((4, 4), 2.5))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_pair_sharding_keeps_pairs_local():
    from multishiftseg_amd import ddp
    pairs = 8
    owned = [ddp.shard_pairs(pairs, r, 8) for r in range(8)]
    assert sorted(sum(owned, [])) == list(range(pairs))
    assert all(len(o) == 1 for o in owned)                        # C3: one pair (2 images) per GPU
    owned = [ddp.shard_pairs(8, r, 2) for r in range(2)]
    assert owned == [[0, 2, 4, 6], [1, 3, 5, 7]]


def test_trainer_stage_sets():
    """a-7: substring freezing gives the reference's two trainable sets (4 864 / 30 749 952 params)."""
    from multishiftseg_amd.deepv3 import DeepWV3Plus
    from multishiftseg_amd.trainer import BACKWARD_ORDER, STAGE_TRAINABLE, configure_trainable_params
    with torch.device("meta"):
        m = DeepWV3Plus(19)
    p1, n1 = configure_trainable_params(m, STAGE_TRAINABLE[1])
    assert n1 == ["ood_head.weight"] and sum(p.numel() for p in p1) == 4864
    p2, n2 = configure_trainable_params(m, STAGE_TRAINABLE[2])
    assert sum(p.numel() for p in p2) == 30749952
    assert set(n2) <= set(BACKWARD_ORDER)
    assert not any(p.requires_grad for n, p in m.named_parameters() if n.startswith("mod"))


def test_stage2_bucket_layout_leaves_a_small_tail():
    """Stage 2 under data parallelism: the gradients are produced heads -> dilated ASPP branches -> the two 4 MB branches, every
    37.7 MB branch gradient closes its bucket (its all-reduce starts beside the next branch's GEMMs) and what is left after the
    last kernel of the backward is the small bucket only. Layout logic only: no process group needed."""
    from multishiftseg_amd import ddp
    from multishiftseg_amd.deepv3 import DeepWV3Plus
    from multishiftseg_amd.trainer import BACKWARD_ORDER, STAGE_TRAINABLE, configure_trainable_params
    with torch.device("meta"):
        m = DeepWV3Plus(19)
    params, names = configure_trainable_params(m, STAGE_TRAINABLE[2])
    named = dict(zip(names, params))
    sync = ddp.GradAllReduce([(n, named[n]) for n in BACKWARD_ORDER if n in named], 64 << 20)
    assert sum(sync.sizes) == 30749952 and len(sync.buckets) == 4
    assert [b[-1] for b in sync.buckets[:3]] == [f"aspp.features.{i}.0.weight" for i in (3, 2, 1)]
    assert 4 * sync.sizes[-1] < (9 << 20) and sync.buckets[-1][-1] == "aspp.img_conv.0.weight"
    assert [n for b in sync.buckets for n in b] == [n for n in BACKWARD_ORDER if n in named]


def test_bench_self_launch_gloo():
    """`python bench.py --gpus 2` without a launcher starts two fresh rank processes itself (torch.distributed.run as a
    child, never exec) and relays ONE JSON line from rank 0. launchcheck is the model-free workload that runs on CPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MSS_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "launchcheck"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["config"]["gloo_ranks"] == 2 and j["config"]["averages_correct"] is True
    assert j["config"]["rccl_ranks"] == 0
