"""RelContrastiveLoss on MI355X -- host side of the fused loss kernels (csrc/loss.hip).

Mirrors lib/loss.py:7-156: same constructor (``param_dict`` keys), same
``forward(logits, anomaly_score, targets) -> 0-dim tensor``, same in-place mutation of ``targets``
(loss.py:110-111,115), NaN when a batch has no OOD pixel (mean of an empty tensor). The value and
both gradients come out of one sequence of streaming kernels; autograd only sees one node.

Pair sampling (loss.py:129-131 draws three ``torch.randperm`` on the CPU generator):
  * ``pairing="reference"`` (default for parity): draws the same three CPU permutations, in the same
    order, from torch's default generator -> with the same ``torch.manual_seed`` the loss equals the
    reference's to fp32 rounding. Costs one device->host read of the three set sizes.
  * ``pairing="device"``: keyed Feistel bijections evaluated inside the kernel; no host round trip
    and no materialised permutation (same distribution, different sample). Used by bench.py.
  * explicit ``perms=(orig, aug, ood)`` int64 tensors: parity tests inject the recorded ones.

Data parallelism (``sync``): the reference computes the loss once over the batch gathered on device 0
(train_deeplab.py:197-198). With one process per GPU,
  * ``sync="local"``: every rank applies the loss to its own (orig, aug) pairs -- no communication;
    the easiest-80 % threshold and the random pairing are per rank (same expectation, other sample).
  * ``sync="global"``: the reference's semantics over the union of all ranks' pairs with a few tiny
    collectives: one all-reduce of 7 partial sums, four all-reduces of a 256-bin histogram (exact global
    k-th smallest CE), an all-gather of the OOD scores (a few MB) and an all-reduce of their gradients.
    Each rank returns the GLOBAL loss value and gradients already multiplied by the world size, so
    that DDP's gradient averaging yields exactly d(global loss)/d(theta).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._lib import MssRclArgs, call, ptr


class _RclFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, score, targets, mod, perms):
        need_dl = logits.requires_grad
        need_ds = score.requires_grad
        out, dlogit, dscore = mod._run(logits.detach(), score.detach(), targets, need_dl, need_ds, perms)
        ctx.dlogit, ctx.dscore = dlogit, dscore
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        dl = ctx.dlogit * g if ctx.dlogit is not None else None
        ds = ctx.dscore * g if ctx.dscore is not None else None
        return dl, ds, None, None, None


class RelContrastiveLoss(nn.Module):
    def __init__(self, param_dict, pairing="reference", seed=0, sync="local", group=None):
        super().__init__()
        if sync not in ("local", "global"):
            raise ValueError(sync)
        self.sync, self.group = sync, group
        self.inoutaug_contras_margins_tri = param_dict.get("inoutaug_contras_margins_tri", None)
        self.sample_ratio = param_dict.get("sample_ratio", 1)
        self.conduct_pixel_selection = param_dict.get("conduct_pixel_selection", False)
        self.selection_ratio = param_dict.get("selection_ratio", 1.0)
        self.ce_weights = param_dict.get("ce_weights", [1, 1])
        self.contras_weight = param_dict.get("contras_weight", 1.0)
        self.in_id = 99
        self.void_id = 255
        if pairing not in ("reference", "device"):
            raise ValueError(pairing)
        self.pairing = pairing
        self._step = int(seed)
        self.last_terms = None   # device tensor [8]: loss, ce_orig, ce_aug, c_orig, c_aug, c_in

    def forward(self, logits, anomaly_score, targets, perms=None):
        if not logits.is_cuda:
            raise RuntimeError("RelContrastiveLoss (multishiftseg_amd) runs on an MI355X only; there is no CPU path")
        if targets.dtype != torch.int64 or not targets.is_contiguous():
            raise RuntimeError("targets must be a contiguous int64 tensor (it is mutated in place, loss.py:110-111)")
        return _RclFn.apply(logits, anomaly_score, targets, self, perms)

    def value_and_grads(self, logits, anomaly_score, targets, perms=None):
        """(loss, dloss/dlogits, dloss/dscore) in one call, for a training loop that feeds the two gradients straight into
        `torch.autograd.backward((logits, score), (dlogit, dscore))`: the kernels produce them together with the value
        anyway, and going through the autograd node costs `grad * upstream` -- a full extra read + write of the 318 MB logit
        gradient at 2x19x1024x2048 (0.13 ms per step) for an upstream gradient that is exactly 1 (train_deeplab.py:198-202:
        `loss.mean()` of a 0-dim tensor, `loss.backward()`). Same mutation of `targets`, same NaN behaviour as forward()."""
        if not logits.is_cuda:
            raise RuntimeError("RelContrastiveLoss (multishiftseg_amd) runs on an MI355X only; there is no CPU path")
        if targets.dtype != torch.int64 or not targets.is_contiguous():
            raise RuntimeError("targets must be a contiguous int64 tensor (it is mutated in place, loss.py:110-111)")
        out, dlogit, dscore = self._run(logits.detach(), anomaly_score.detach(), targets, True, True, perms)
        return out[0], dlogit, dscore

    def _run(self, logits, score, targets, need_dl, need_ds, perms):
        import torch.distributed as dist
        if self.sync == "global" and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if perms is not None:
                raise ValueError("explicit permutations are a single-process (parity) feature")
            return self._run_global(logits, score, targets, need_dl, need_ds)
        return self._run_local(logits, score, targets, need_dl, need_ds, perms)

    def _args(self, logits, score, targets, w_aug_scale=1.0, batch_scale=1):
        B, C, H, W = logits.shape
        a = MssRclArgs()
        a.logit, a.score, a.target = ptr(logits), ptr(score), ptr(targets)
        a.B, a.C, a.H, a.W = B * batch_scale, C, H, W
        a.w_ce_orig, a.w_ce_aug = float(self.ce_weights[0]), float(self.ce_weights[1]) * w_aug_scale
        a.w_contras = float(self.contras_weight)
        m = self.inoutaug_contras_margins_tri
        a.m0, a.m1, a.m2 = float(m[0]), float(m[1]), float(m[2])
        select = bool(self.conduct_pixel_selection and 0.0 < self.selection_ratio < 1.0)
        a.select, a.selection_ratio = int(select), float(self.selection_ratio)
        return a, select

    def _run_global(self, logits, score, targets, need_dl, need_ds):
        """Reference semantics over the union of all ranks' pairs (module docstring, sync="global")."""
        import torch.distributed as dist
        g = self.group
        Wn, rank = dist.get_world_size(g), dist.get_rank(g)
        logits = logits.contiguous().float()
        score = score.contiguous().float()
        B, C, H, W = logits.shape
        if B < 2:
            raise RuntimeError("RelContrastiveLoss needs an (original, augmented) pair: batch >= 2")
        dev = logits.device
        total, half = B * H * W, (B // 2) * H * W
        if B % 2:
            raise RuntimeError(f"RelContrastiveLoss needs [orig...; aug...] pairs: batch {B} is odd")
        # Gradients come out multiplied by the world size (see docstring). With the easiest-k selection the kernel divides
        # by the GLOBAL k, so the factor goes into the weight; without it the kernel divides by the LOCAL half, which
        # already is (global half) / world, so no extra factor is needed (and none for the original half either way).
        select_on = bool(self.conduct_pixel_selection and 0.0 < self.selection_ratio < 1.0)
        a, select = self._args(logits, score, targets, w_aug_scale=float(Wn) if select_on else 1.0)
        ra = ctypes.byref(a)
        lse = torch.empty(total, device=dev, dtype=torch.float32)
        ce_aug = torch.empty(half, device=dev, dtype=torch.float32)
        kind = torch.empty(total, device=dev, dtype=torch.uint8)
        counters = torch.empty(16, device=dev, dtype=torch.float64)
        sel = torch.zeros(8, device=dev, dtype=torch.int32)
        hist = torch.empty(256, device=dev, dtype=torch.int32)
        dlogit = torch.empty_like(logits) if need_dl else None
        call("mss_rcl_pass1_f32", ra, ptr(lse), ptr(ce_aug), ptr(kind), ptr(counters), ptr(dlogit))
        dist.all_reduce(counters, group=g)                     # slots 0..6 and 12 are sums/counts; the rest still zero
        if select:
            call("mss_rcl_select_init_f32", ptr(counters), float(self.selection_ratio), ptr(hist), ptr(sel))
            local_last = None
            for shift in (24, 16, 8, 0):
                call("mss_rcl_select_hist_f32", ptr(ce_aug), half, ptr(sel), shift, ptr(hist))
                if shift == 0:
                    local_last = hist.clone()
                dist.all_reduce(hist, group=g)
                call("mss_rcl_select_pick_f32", ptr(sel), ptr(hist), shift)
            # elements equal to the threshold: rank r takes what ranks < r left of the global quota
            local_eq = local_last[(sel[0] & 255).long()].reshape(1)
            eqs = [torch.empty_like(local_eq) for _ in range(Wn)]
            dist.all_gather(eqs, local_eq, group=g)
            before = torch.stack(eqs[:rank]).sum() if rank else torch.zeros((), device=dev, dtype=torch.int32)
            sel[3] = torch.clamp(sel[3] - before, min=0).minimum(local_eq[0])
            call("mss_rcl_pass2_f32", ra, ptr(lse), ptr(ce_aug), ptr(kind), ptr(sel), ptr(counters), 1.0, ptr(dlogit))
        nb = _lib.value("mss_rcl_num_compact_blocks", B, H, W)
        idx = torch.empty((3, total), device=dev, dtype=torch.int32)
        block_counts = torch.empty(3 * nb, device=dev, dtype=torch.int32)
        n_out = torch.zeros(4, device=dev, dtype=torch.int32)
        call("mss_rcl_compact_f32", ptr(kind), B, H, W, ptr(idx[0]), ptr(idx[1]), ptr(idx[2]), ptr(block_counts), ptr(n_out))
        wc = float(self.contras_weight)
        dscore = torch.empty_like(score) if need_ds else None
        if need_ds:
            call("mss_rcl_cin_bwd_f32", ra, ptr(kind), ptr(counters), wc * Wn, ptr(dscore))
        # set sizes of every rank (the one host read of this mode; the reference syncs several times here)
        all_n = [torch.empty_like(n_out) for _ in range(Wn)]
        dist.all_gather(all_n, n_out, group=g)
        sizes = torch.stack(all_n)[:, :3].tolist()             # [rank][orig, aug, ood]
        n_orig_g, n_aug_g, n_ood_g = (sum(s[k] for s in sizes) for k in range(3))
        n_pairs = min(int(total * Wn * self.sample_ratio), n_ood_g, n_orig_g, n_aug_g)
        cap = max(1, max(s[2] for s in sizes))
        mine = torch.zeros(cap, device=dev, dtype=torch.float32)
        call("mss_rcl_gather_f32", ptr(score), ptr(idx[2]), sizes[rank][2], ptr(mine))
        parts = [torch.empty_like(mine) for _ in range(Wn)]
        dist.all_gather(parts, mine, group=g)
        ood_all = torch.cat(parts)
        offs = [0]
        for s_ in sizes:
            offs.append(offs[-1] + s_[2])
        ood_off = torch.tensor(offs, device=dev, dtype=torch.int32)
        g_ood = torch.zeros(Wn * cap, device=dev, dtype=torch.float32) if need_ds else None
        self._step += 1
        s0 = (self._step * 0x9E3779B1) & 0xFFFFFFFF            # same on every rank: seed and step count are shared
        coef = wc * Wn / n_pairs if n_pairs else 0.0
        for slot, (set_a, n_a_g, margin) in enumerate(((0, n_orig_g, a.m0), (1, n_aug_g, a.m1))):
            a_off = sum(s_[set_a] for s_ in sizes[:rank])
            call("mss_rcl_pairs_global_f32", ptr(score), ptr(idx[set_a]), a_off, sizes[rank][set_a], n_a_g, ptr(ood_all),
                 ptr(ood_off), Wn, cap, n_pairs, (s0 + 1 + slot) & 0xFFFFFFFF, (s0 + 7) & 0xFFFFFFFF, float(margin),
                 ptr(counters), slot, coef, ptr(dscore), ptr(g_ood))
        if need_ds:
            dist.all_reduce(g_ood, group=g)
            call("mss_rcl_scatter_add_f32", ptr(g_ood[rank * cap:]), ptr(idx[2]), sizes[rank][2], ptr(dscore))
        part = counters[7:11].clone()                            # selected-CE sum/count, the two hinge sums
        dist.all_reduce(part, group=g)
        counters[7:11] = part
        out = torch.empty(8, device=dev, dtype=torch.float32)
        fa, _ = self._args(logits, score, targets, batch_scale=Wn)   # true weights, global pixel count
        call("mss_rcl_finalize_f32", ctypes.byref(fa), ptr(counters), ptr(sel), ptr(out))
        self.last_terms = out
        return out, dlogit, dscore

    def _run_local(self, logits, score, targets, need_dl, need_ds, perms):
        logits = logits.contiguous().float()
        score = score.contiguous().float()
        B, C, H, W = logits.shape
        if B < 2:
            raise RuntimeError("RelContrastiveLoss needs an (original, augmented) pair: batch >= 2")
        if B % 2:
            raise RuntimeError(f"RelContrastiveLoss needs [orig...; aug...] pairs: batch {B} is odd")
        dev = logits.device
        h = B // 2
        total, half = B * H * W, h * H * W
        a = MssRclArgs()
        a.logit, a.score, a.target = ptr(logits), ptr(score), ptr(targets)
        a.B, a.C, a.H, a.W = B, C, H, W
        a.w_ce_orig, a.w_ce_aug, a.w_contras = float(self.ce_weights[0]), float(self.ce_weights[1]), float(self.contras_weight)
        m = self.inoutaug_contras_margins_tri
        a.m0, a.m1, a.m2 = float(m[0]), float(m[1]), float(m[2])
        select = bool(self.conduct_pixel_selection and 0.0 < self.selection_ratio < 1.0)
        a.select, a.selection_ratio = int(select), float(self.selection_ratio)
        ra = ctypes.byref(a)

        dlogit = torch.empty_like(logits) if need_dl else None
        if perms is None and self.pairing == "device":
            # no host round trip anywhere in this mode: ONE call issues the whole launch sequence (pass 1, radix select, pass 2,
            # compaction, the three hinge terms, finalize) out of one workspace -- issued one by one through this binding the
            # ~17 launches were bound by the host (csrc/loss.hip, mss_rcl_loss_device_f32)
            nbytes = _lib.value("mss_rcl_workspace_bytes", B, H, W)
            ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            dscore = torch.empty_like(score) if need_ds else None
            out = torch.empty(8, device=dev, dtype=torch.float32)
            self._step += 1
            call("mss_rcl_loss_device_f32", ra, ptr(ws), nbytes, int(total * self.sample_ratio), self._step & 0xFFFFFFFF, ptr(dlogit),
                 ptr(dscore), ptr(out))
            self.last_terms = out
            return out, dlogit, dscore
        lse = torch.empty(total, device=dev, dtype=torch.float32)
        ce_aug = torch.empty(half, device=dev, dtype=torch.float32)
        kind = torch.empty(total, device=dev, dtype=torch.uint8)
        counters = torch.empty(16, device=dev, dtype=torch.float64)
        sel = torch.zeros(8, device=dev, dtype=torch.int32)
        hist = torch.empty(256, device=dev, dtype=torch.int32)
        call("mss_rcl_pass1_f32", ra, ptr(lse), ptr(ce_aug), ptr(kind), ptr(counters), ptr(dlogit))
        if select:
            call("mss_rcl_select_f32", ptr(ce_aug), half, ptr(counters), float(self.selection_ratio), ptr(hist), ptr(sel))
            call("mss_rcl_pass2_f32", ra, ptr(lse), ptr(ce_aug), ptr(kind), ptr(sel), ptr(counters), 1.0, ptr(dlogit))
        # contrastive part
        nb = _lib.value("mss_rcl_num_compact_blocks", B, H, W)
        idx = torch.empty((3, total), device=dev, dtype=torch.int32)
        block_counts = torch.empty(3 * nb, device=dev, dtype=torch.int32)
        n_out = torch.empty(4, device=dev, dtype=torch.int32)
        call("mss_rcl_compact_f32", ptr(kind), B, H, W, ptr(idx[0]), ptr(idx[1]), ptr(idx[2]), ptr(block_counts), ptr(n_out))
        dscore = torch.empty_like(score) if need_ds else None
        wc = float(self.contras_weight)
        if need_ds:
            call("mss_rcl_cin_bwd_f32", ra, ptr(kind), ptr(counters), wc, ptr(dscore))
        max_samples = int(total * self.sample_ratio)
        # reference pairing / injected permutations (the device-pairing mode returned above)
        n_orig, n_aug, n_ood = (int(v) for v in n_out[:3].tolist())      # host sync, as the reference's .sum()/int()
        n_bad = int(counters[12].item())
        if n_bad:       # F.nll_loss raises here (loss.py:59); the device-pairing mode reports NaN + last_terms[6]
            raise IndexError(f"{n_bad} target value(s) outside [0, {C}) and below in_id=99")
        n = min(max_samples, n_ood, n_orig, n_aug)                         # loss.py:149-156
        if perms is None:
            perms = [torch.randperm(k) for k in (n_orig, n_aug, n_ood)]    # CPU default generator, loss.py:129-131
        p_orig, p_aug, p_ood = (p[:n].to(dev, torch.int64).contiguous() for p in perms)
        for slot, (set_a, pa, margin) in enumerate(((0, p_orig, a.m0), (1, p_aug, a.m1))):
            call("mss_rcl_pairs_f32", ptr(score), ptr(idx[set_a]), ptr(pa), ptr(idx[2]), ptr(p_ood), n, float(margin),
                 ptr(counters), slot, wc, ptr(dscore))
        out = torch.empty(8, device=dev, dtype=torch.float32)
        call("mss_rcl_finalize_f32", ra, ptr(counters), ptr(sel), ptr(out))
        self.last_terms = out
        return out, dlogit, dscore
