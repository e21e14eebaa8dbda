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
    params = [(n, torch.zeros(s)) for n, s in shapes.items()]
    sync = ddp.GradAllReduce(params, bucket_bytes=30000)          # forces several buckets
    assert len(sync.buckets) >= 2
    grads = {n: torch.full(s, float(rank + 1)) + torch.arange(int(torch.tensor(s).prod())).reshape(s) for n, s in shapes.items()}
    expect = {n: (sum(torch.full(s, float(k + 1)) for k in range(world)) / world
                  + torch.arange(int(torch.tensor(s).prod())).reshape(s)) for n, s in shapes.items()}
    for n in shapes:                                              # arrival order = backward order
        sync(n, grads[n])
    sync.backward_done()
    ok = all(torch.allclose(grads[n], expect[n]) for n in shapes)
    # second step reuses the object
    for n in shapes:
        grads[n].fill_(float(rank))
        sync(n, grads[n])
    sync.backward_done()
    ok = ok and all(torch.allclose(grads[n], torch.full_like(grads[n], (world - 1) / 2)) for n in shapes)
    # a parameter that never arrives (frozen later) must not dead-lock the bucket
    sync2 = ddp.GradAllReduce(params, bucket_bytes=1 << 30)
    g = torch.full((19, 256, 1, 1), float(rank))
    sync2("ood_head.weight", g)
    sync2.backward_done()
    ok = ok and torch.allclose(g, torch.full_like(g, (world - 1) / 2))
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
