"""End-to-end data-parallel training step on the GPU box: two ranks (gloo, both on the one GPU the
box has -- RCCL itself needs distinct devices) run TrainStep on different pairs; the averaged
gradients must equal the mean of the per-rank gradients computed without DDP, and both ranks must
end with identical parameters."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(stage):
    from multishiftseg_amd import synth
    from multishiftseg_amd.deepv3 import DeepWV3Plus
    from multishiftseg_amd.loss import RelContrastiveLoss
    from multishiftseg_amd.trainer import LOSS_PARAMS, TrainStep
    m = DeepWV3Plus(19)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.deepwv3plus_params(0).items()})
    m = m.cuda()
    m.uncertainty_func_init()
    crit = RelContrastiveLoss(LOSS_PARAMS, pairing="reference")
    return m, TrainStep(m, crit, stage=stage)


def _data(rank):
    from multishiftseg_amd import synth
    img = torch.from_numpy(synth.synth_image(40 + rank, 2, 64, 128)).cuda()
    tgt = torch.from_numpy(synth.synth_targets(40 + rank, 1, 64, 128)).cuda()
    masks = {"mod6": torch.ones(2, 1024), "mod7": torch.ones(2, 2048)}      # deterministic "dropout"
    return img, tgt, masks


def _worker(rank, world, port, stage, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MSS_DIST_BACKEND="gloo")
    from multishiftseg_amd import ddp
    ddp.init_from_env()
    model, step = _build(stage)
    img, tgt, masks = _data(rank)
    model.dropout_masks = masks
    torch.manual_seed(7)
    loss = step(img, tgt.clone())
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    params = {n: p.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    out[rank] = dict(loss=float(loss.detach()), grads=grads, params=params)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stage", [1, 2])
def test_two_rank_step_matches_manual_average(stage):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), stage, out), nprocs=world, join=True)
    res = dict(out)
    assert np.isfinite(res[0]["loss"]) and np.isfinite(res[1]["loss"])
    for n in res[0]["params"]:                      # replicas stay in lock-step
        assert torch.equal(res[0]["params"][n], res[1]["params"][n]), n
        assert torch.equal(res[0]["grads"][n], res[1]["grads"][n]), n
    # the same two micro-batches without any process group: gradients averaged by hand
    local = []
    for rank in range(world):
        model, step = _build(stage)
        assert step.sync is None
        img, tgt, masks = _data(rank)
        model.dropout_masks = masks
        torch.manual_seed(7)
        score, logit = model(img)
        loss = step.criterion(logit, score, tgt.clone()).mean()
        loss.backward()
        local.append({n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad})
        np.testing.assert_allclose(float(loss.detach()), res[rank]["loss"], rtol=1e-5)
    for n in local[0]:
        mean = (local[0][n] + local[1][n]) / 2
        scale = mean.abs().max().item() + 1e-20
        assert (res[0]["grads"][n] - mean).abs().max().item() / scale < 1e-3, n


def _rccl_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("MSS_DIST_BACKEND", None)
    from multishiftseg_amd import ddp
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    torch.manual_seed(0)
    params = [(f"p{i}", torch.nn.Parameter(torch.randn(n, device="cuda"))) for i, n in enumerate((1000, 70000, 3, 250000))]
    sink = ddp.GradAllReduce(params, bucket_bytes=300000, force=True)
    assert len(sink.buckets) >= 2 and sink.active
    grads = {n: torch.randn_like(p) for n, p in params}
    want = {n: g.clone() for n, g in grads.items()}
    got = {}
    for n, _ in reversed(params):              # delivered in reverse bucket order: launches still go out in bucket order
        if n != "p2":                          # one gradient never arrives: travels as zeros, flush() sends its bucket
            got[n] = sink(n, grads[n])
    sink.backward_done()
    torch.cuda.synchronize()
    ok = all(torch.equal(got[n], want[n]) for n in got)             # the mean over one rank is the gradient itself
    ok = ok and dict(params)["p2"].grad is not None and not dict(params)["p2"].grad.any()
    t = torch.full((5,), 3.0, device="cuda")
    dist.all_reduce(t)
    out["ok"] = bool(ok) and bool((t == 3.0).all().item()) and sink.comm_stream is not None
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_one_rank_runs_the_bucketed_allreduce():
    """The `nccl` (= RCCL) branch of ddp.GradAllReduce -- side stream, record_stream, async work handles, flush of an
    unfilled bucket -- on the one GPU this box has: a one-rank group, collectives forced on."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert dict(out).get("ok") is True


def _rccl_step_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", MSS_DDP_FORCE="1")
    os.environ.pop("MSS_DIST_BACKEND", None)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res = {}
    for forced in (True, False):
        os.environ["MSS_DDP_FORCE"] = "1" if forced else "0"
        model, step = _build(2)
        assert (step.sync is not None and step.sync.active) == forced
        img, tgt, masks = _data(0)
        model.dropout_masks = masks
        torch.manual_seed(7)
        loss = step(img, tgt.clone())
        res[forced] = (float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad})
    out["ok"] = res[True][0] == res[False][0] and all(torch.equal(res[True][1][n], res[False][1][n]) for n in res[True][1])
    out["n"] = len(res[True][1])
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_one_rank_train_step():
    """A whole stage-2 TrainStep with the gradient all-reduce running over RCCL (one-rank group, forced): bucket launches
    from inside the backward on the side stream, backward_done, Adam -- identical loss and gradients to the un-synchronised
    step."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_step_worker, args=(_free_port(), out), nprocs=1, join=True)
    o = dict(out)
    assert o.get("ok") is True and o.get("n") == 18
