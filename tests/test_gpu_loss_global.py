"""sync="global": two ranks (gloo, both on the box's one GPU) must reproduce what ONE process computes
on the concatenated batch [orig_r0, orig_r1; aug_r0, aug_r1] -- the reference's gathered-batch semantics
(train_deeplab.py:194-198) -- with gradients pre-multiplied by the world size."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
PARAMS = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
          "inoutaug_contras_margins_tri": [10, 5, 5]}
H, W = 24, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_rank_data(rank, crafted):
    rng = np.random.default_rng(100 + rank)
    logits = rng.standard_normal((2, 19, H, W), dtype=np.float32) * 3
    score = rng.standard_normal((2, H, W), dtype=np.float32) * 4
    if crafted:      # every set has 512 elements per rank, so all elements are paired exactly once
        t = np.full((2, H, W), 254, dtype=np.int64)
        t[:, :16, :] = rng.integers(0, 19, size=(2, 16, W))
    else:
        from multishiftseg_amd import synth
        t = synth.synth_targets(100 + rank, 1, H, W)
    return logits, score, t


def _worker(rank, world, port, crafted, params, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MSS_DIST_BACKEND="gloo")
    from multishiftseg_amd import ddp
    from multishiftseg_amd.loss import RelContrastiveLoss
    ddp.init_from_env()
    logits, score, t = make_rank_data(rank, crafted)
    lt = torch.from_numpy(logits).cuda().requires_grad_(True)
    st = torch.from_numpy(score).cuda().requires_grad_(True)
    tt = torch.from_numpy(t).cuda()
    crit = RelContrastiveLoss(params, pairing="device", seed=5, sync="global")
    loss = crit(lt, st, tt)
    loss.backward()
    out[rank] = dict(loss=float(loss.detach()), terms=crit.last_terms.cpu().numpy(), dl=lt.grad.cpu().numpy(),
                     ds=st.grad.cpu().numpy(), tmut=tt.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("crafted,select", [(True, True), (False, True), (True, False), (False, False)])
def test_global_sync_equals_single_process_on_concatenated_batch(crafted, select):
    """select=False: conduct_pixel_selection off -- the augmented CE is a plain mean over the (global) half, whose
    pre-multiplied gradient is w1 / local half (it used to come out world-size times too large)."""
    from multishiftseg_amd.loss import RelContrastiveLoss
    world = 2
    params = dict(PARAMS, inoutaug_contras_margins_tri=[100, 100, 5]) if crafted else dict(PARAMS)
    params["conduct_pixel_selection"] = select
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), crafted, params, out), nprocs=world, join=True)
    res = dict(out)
    data = [make_rank_data(r, crafted) for r in range(world)]
    cat = lambda k: np.concatenate([data[0][k][:1], data[1][k][:1], data[0][k][1:], data[1][k][1:]])
    lt = torch.from_numpy(cat(0)).cuda().requires_grad_(True)
    st = torch.from_numpy(cat(1)).cuda().requires_grad_(True)
    tt = torch.from_numpy(cat(2)).cuda()
    crit = RelContrastiveLoss(params, pairing="device", seed=5, sync="local")
    loss = crit(lt, st, tt)
    loss.backward()
    terms = crit.last_terms.cpu().numpy()
    dl, ds, tm = lt.grad.cpu().numpy(), st.grad.cpu().numpy(), tt.cpu().numpy()
    for r in range(world):
        np.testing.assert_allclose(res[r]["terms"][[1, 2, 5]], terms[[1, 2, 5]], rtol=1e-5)   # ce_orig, ce_aug, c_in
        # this rank's images inside the concatenated batch: orig at r, aug at world + r
        np.testing.assert_allclose(res[r]["dl"] / world, dl[[r, world + r]], rtol=1e-4, atol=1e-9)
        np.testing.assert_array_equal(res[r]["tmut"], tm[[r, world + r]])
        if crafted:   # every element paired once and every hinge active: pairing-independent
            np.testing.assert_allclose(res[r]["terms"][[3, 4]], terms[[3, 4]], rtol=1e-5)
            np.testing.assert_allclose(res[r]["loss"], float(loss.detach()), rtol=1e-5)
            np.testing.assert_allclose(res[r]["ds"] / world, ds[[r, world + r]], rtol=1e-4, atol=1e-9)
        else:
            np.testing.assert_allclose(res[r]["terms"][[3, 4]], terms[[3, 4]], rtol=0.15)
            assert np.isfinite(res[r]["ds"]).all()
    assert res[0]["loss"] == res[1]["loss"]                     # every rank reports the global value
