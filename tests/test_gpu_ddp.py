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
