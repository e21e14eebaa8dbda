"""Pixel-level OOD metrics on the GPU: host-side mirror of ``lib/utils/metric.py``'s
``eval_ood_measure`` (:170-180) with the same arguments and return convention -- ``(auroc, aupr, fpr)`` or ``None``
when one of the two classes has no pixel -- over device tensors, so the evaluation sweep (test_deeplab.py:84-102,
train_deeplab.py:223-241) never copies a score map to the host. ``OODMeter`` is the streaming form of the
reference's "append every batch, concatenate, evaluate" loop.

All arithmetic is in libmss_hip.so (csrc/metric.hip): exact rank statistics on sorted 32-bit keys, no binning.
"""
import ctypes

import torch

from . import _lib
from ._lib import call, ptr


def _flat(score, label):
    if not (score.is_cuda and label.is_cuda):
        raise RuntimeError("OOD metrics (multishiftseg_amd) run on an MI355X only; there is no CPU path")
    if score.numel() != label.numel():
        raise ValueError(f"score has {score.numel()} elements, label has {label.numel()}")
    return score.detach().reshape(-1).float().contiguous(), label.detach().reshape(-1).long().contiguous()


class OODMeter:
    """update(anomaly_score, target) per batch, compute() at the end of the sweep."""

    def __init__(self, train_id_in=0, train_id_out=1, recall_level=0.95):
        if train_id_in == train_id_out:
            raise ValueError("train_id_in and train_id_out must differ")
        self.id_in, self.id_out, self.recall_level = int(train_id_in), int(train_id_out), float(recall_level)
        self.reset()

    def reset(self):
        self._chunks = []          # (keys [n] u32: inliers packed at the front, OOD at the back; counts [2] i64), on device

    def update(self, score, label):
        """One kernel per batch, no host synchronisation (the reference copies both maps to the host here)."""
        score, label = _flat(score, label)
        n = score.numel()
        if n == 0:
            return
        keys = torch.empty(n, dtype=torch.int32, device=score.device)
        counts = torch.zeros(2, dtype=torch.int64, device=score.device)
        call("mss_oodm_compact_f32", ptr(score), ptr(label), n, self.id_in, self.id_out, ptr(keys), ptr(counts))
        self._chunks.append((keys, counts))

    @staticmethod
    def _sorted(parts):
        keys = parts[0] if len(parts) == 1 else torch.cat(parts)
        n = keys.numel()
        out = torch.empty_like(keys)
        temp = torch.empty(_lib.value("mss_oodm_sort_temp_bytes", n), dtype=torch.uint8, device=keys.device)
        call("mss_oodm_sort_u32", ptr(keys), ptr(out), n, ptr(temp), temp.numel())
        return out

    def compute(self):
        """(auroc, aupr, fpr) as Python floats, or None if there is no in- or no out-of-distribution pixel."""
        if not self._chunks:
            return None
        counts = torch.stack([c for _, c in self._chunks]).tolist()          # the sweep's only D2H before the result
        negs = [k[:c[0]] for (k, _), c in zip(self._chunks, counts) if c[0]]
        poss = [k[k.numel() - c[1]:] for (k, _), c in zip(self._chunks, counts) if c[1]]
        if not negs or not poss:
            return None
        pos, neg = self._sorted(poss), self._sorted(negs)
        P, N = pos.numel(), neg.numel()
        nb = _lib.value("mss_oodm_rank_blocks", P)
        u2 = torch.empty(nb, dtype=torch.int64, device=pos.device)
        ap = torch.empty(nb, dtype=torch.float64, device=pos.device)
        out = torch.empty(3, dtype=torch.float64, device=pos.device)
        call("mss_oodm_measures_f64", ptr(pos), P, ptr(neg), N, ctypes.c_double(self.recall_level), ptr(u2), ptr(ap), ptr(out))
        auroc, aupr, fpr = out.tolist()
        return auroc, aupr, fpr


def eval_ood_measure(conf, seg_label, train_id_in=0, train_id_out=1):
    """Drop-in for lib/utils/metric.py:170-180 on CUDA tensors (any shape, equal element counts)."""
    m = OODMeter(train_id_in, train_id_out)
    m.update(conf, seg_label)
    return m.compute()
