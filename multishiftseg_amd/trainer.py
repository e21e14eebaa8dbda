"""Training / evaluation loop semantics of the reference harness, for the build's own drivers.

Mirrors what matters on the hot path of train_deeplab.py:113-216 and test_deeplab.py:74-117:
  * trainable set chosen by substring match on parameter names (train_deeplab.py:124-130), stage 1
    = ["ood_head"], stage 2 = ["aspp", "bot_fine", "bot_aspp", "ood_head"] (exps/DeepLab.yaml:10-11);
  * Adam(lr 1e-4 -> 1e-6, weight_decay 1e-4) rebuilt from scratch at the stage switch (:151-166);
  * model.train() with the frozen trunk in batch-statistics mode (:182);
  * batch = cat([img, div_img]), cat([target, div_target]) (:194-195); loss.mean(); zero_grad /
    backward / step (:198-204);
  * per-rank data parallelism instead of nn.DataParallel (multishiftseg_amd/ddp.py).
Datasets, logging, checkpoint policy and metrics are the reference's harness and stay out of scope.
"""
import os

import torch

from . import ddp
from .optim import Adam

STAGE_TRAINABLE = {1: ["ood_head"], 2: ["aspp", "bot_fine", "bot_aspp", "ood_head"]}   # exps/DeepLab.yaml:10-11
STAGE_LR = {1: 1.0e-4, 2: 1.0e-6}                                                       # exps/DeepLab.yaml:21-22
WEIGHT_DECAY = 1.0e-4                                                                    # exps/DeepLab.yaml:24
LOSS_PARAMS = {"ce_weights": [50, 10], "conduct_pixel_selection": True, "selection_ratio": 0.8,
               "inoutaug_contras_margins_tri": [10, 5, 5]}                               # exps/DeepLab.yaml:28-35

# order in which DeepWV3Plus._head_backward produces gradients: heads first, then the three dilated ASPP branches (37.7 MB
# each), the 1x1 and image-pooling branches (4 MB each) last, so that the bucket left after the last kernel is small
BACKWARD_ORDER = ["ood_head.weight", "final.6.weight", "final.4.weight", "final.4.bias", "final.3.weight",
                  "final.1.weight", "final.1.bias", "final.0.weight", "bot_fine.weight", "bot_aspp.weight"] + \
    [f"aspp.features.{i}.{s}" for i in (3, 2, 1, 0) for s in ("1.weight", "1.bias", "0.weight")] + \
    ["aspp.img_conv.1.weight", "aspp.img_conv.1.bias", "aspp.img_conv.0.weight"]


def configure_trainable_params(model, patterns):
    """train_deeplab.py:113-132: requires_grad by substring, returns (params, names)."""
    params, names = [], []
    for name, p in model.named_parameters():
        if any(s in name for s in patterns):
            p.requires_grad = True
            params.append(p)
            names.append(name)
        else:
            p.requires_grad = False
    return params, names


class TrainStep:
    """One optimisation step of either stage on this rank's pairs."""

    def __init__(self, model, criterion, stage=2, bucket_bytes=64 << 20):
        """criterion: RelContrastiveLoss; with sync="global" it reproduces the reference's gathered-batch
        loss across ranks (its gradients come pre-multiplied by the world size, which the averaging
        all-reduce of ddp.GradAllReduce undoes); sync="local" needs no loss communication."""
        self.model, self.criterion = model, criterion
        self.keep_outputs, self.last_outputs = False, None
        self.set_stage(stage, bucket_bytes)

    def set_stage(self, stage, bucket_bytes=64 << 20):
        """update_trainable_params (train_deeplab.py:151-166): new trainable set, new Adam."""
        self.stage = stage
        params, names = configure_trainable_params(self.model, STAGE_TRAINABLE[stage])
        self.optimizer = Adam(params, lr=STAGE_LR[stage], weight_decay=WEIGHT_DECAY)
        self.names = names
        named = dict(zip(names, params))
        order = [(n, named[n]) for n in BACKWARD_ORDER if n in named]
        assert len(order) == len(named), sorted(set(named) - set(n for n, _ in order))
        # MSS_DDP_FORCE=1: run the collectives in a one-rank group too (the RCCL path on a single GPU; tests)
        self.sync = ddp.GradAllReduce(order, bucket_bytes, force=os.environ.get("MSS_DDP_FORCE") == "1") \
            if torch.distributed.is_initialized() else None
        self.model.grad_sink = self.sync
        self.model.train()

    def __call__(self, img, target, perms=None):
        """img [2p,3,H,W] laid out [orig...; aug...], target int64 [2p,H,W] (mutated by the loss). perms: the three pairing
        permutations of loss.py:129-131 when a parity test replays recorded ones (RelContrastiveLoss.forward(perms=))."""
        score, logit = self.model(img)
        if self.keep_outputs:                    # parity tests read the step's own forward outputs; off by default (318 MB)
            self.last_outputs = (score.detach(), logit.detach())
        self.optimizer.zero_grad()
        if hasattr(self.criterion, "value_and_grads") and logit.requires_grad:
            # the fused loss hands out its value AND both gradients; d(loss.mean())/d(loss) = 1 (train_deeplab.py:198-202),
            # so they go to autograd as they are -- no `grad * 1` pass over the 318 MB logit gradient
            loss, dlogit, dscore = self.criterion.value_and_grads(logit, score, target, perms=perms)
            torch.autograd.backward((score, logit), (dscore, dlogit))
        else:
            loss = (self.criterion(logit, score, target, perms=perms) if perms is not None
                    else self.criterion(logit, score, target)).mean()
            loss.backward()
        self.optimizer.step()
        return loss


@torch.no_grad()
def ood_scores(model, img, score_only=False):
    """test_deeplab.py:86-90: eval-mode forward -> (anomaly_score, logit). score_only: (anomaly_score, None), the form the
    reference's evaluation actually consumes (test_deeplab.py:92-96) -- the upsampled logit volume is never written."""
    was, was_so = model.training, model.score_only
    model.eval()
    model.score_only = bool(score_only)
    try:
        return model(img)
    finally:
        model.score_only = was_so
        model.train(was)


class GraphedEval:
    """The eval forward (test_deeplab.py:86-90: image -> (anomaly_score, logit)) captured ONCE into a hipGraph and replayed:
    ~350 kernel launches per image become one graph launch, which removes the host launch path from the OOD-score
    throughput (the kernels take raw pointers and sizes, allocate nothing themselves and never synchronise, so the whole
    forward is capturable; scratch tensors come from the graph's private pool). Fixed input shape.

    What a replay reads by raw pointer, and who keeps it alive: BatchNorm buffers and parameters (the model), and the
    PACKED copies of every conv weight (MFMA / Winograd-domain layouts, made during the warm-up outside the graph's pool):
    this object holds references to exactly the packed tensors that were current at capture (`_keep`), so a later re-pack by
    an eager forward cannot free them under the graph. Packed copies do NOT follow in-place weight updates, so every call
    compares each parameter's (version, data_ptr) with the capture-time signature and re-captures on any change
    (optimizer step, load_state_dict, .to()); refresh() forces it."""

    def __init__(self, model, shape, warmup=2, score_only=False):
        self.model, self.shape, self.score_only = model, tuple(shape), bool(score_only)
        self.static_in = torch.zeros(self.shape, device="cuda", dtype=torch.float32)
        self.captures = 0
        self._capture(warmup)

    def _signature(self):
        return [(p._version, p.data_ptr()) for p in self.model.parameters()] + \
               [(b._version, b.data_ptr()) for b in self.model.buffers() if b.dtype.is_floating_point]

    def _capture(self, warmup):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up off the capture: packs weights, fills one-time caches
            for _ in range(warmup):
                ood_scores(self.model, self.static_in, self.score_only)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = ood_scores(self.model, self.static_in, self.score_only)
        # own what the graph reads: the packed forms current right now (+ the fused heads weight of the model)
        self._keep = [dict(p.__dict__.get("_mss_packed", {})) for p in self.model.parameters()] + [self.model._heads_cache]
        self._sig = self._signature()
        self.captures += 1

    def refresh(self):
        self._capture(1)

    def __call__(self, img):
        if tuple(img.shape) != self.shape:
            raise ValueError(f"GraphedEval was captured for {self.shape}, got {tuple(img.shape)}")
        if self._signature() != self._sig:            # weights or BatchNorm buffers changed since the capture
            self._capture(1)
        self.static_in.copy_(img)
        self.graph.replay()
        return self.static_out
