"""TEST INFRASTRUCTURE ONLY -- numpy restatement of RelContrastiveLoss (lib/loss.py:34-156),
forward value, the gradients w.r.t. logits and anomaly_score, and the in-place target mutation.

The three torch.randperm draws (loss.py:129-131) are injected as `perms`; pinned against
tests/golden/rcl_*.npz which record the reference's own permutations, loss, autograd gradients
and mutated targets.
"""
import numpy as np

from .nnops import logsumexp

IN_ID, VOID_ID = 99, 255      # loss.py:31-32


def rel_contrastive_loss(logits, score, targets, params, perms):
    """logits [B,C,H,W], score [B,H,W], targets int64 [B,H,W] (mutated in place like the
    reference), params = the reference's param_dict, perms = (perm_orig, perm_aug, perm_ood) full
    permutations (truncated to n here).  Returns dict(loss, dlogit, dscore, terms)."""
    margins = params.get("inoutaug_contras_margins_tri")
    sample_ratio = params.get("sample_ratio", 1)
    select = params.get("conduct_pixel_selection", False)
    ratio = params.get("selection_ratio", 1.0)
    w0, w1 = params.get("ce_weights", [1, 1])
    wc = params.get("contras_weight", 1.0)
    dt = logits.dtype
    B, C, H, W = logits.shape
    h = B // 2
    ood_mask = (targets > IN_ID) & (targets != VOID_ID)            # :46
    in_mask = targets < IN_ID                                       # :47
    lse = logsumexp(logits, 1)
    tsafe = np.where(in_mask, targets, 0)
    picked = np.take_along_axis(logits, tsafe[:, None], axis=1)[:, 0]
    ce = np.where(in_mask, lse - picked, 0).astype(dt)             # NLL(log_softmax), ignore -> 0   :59-60
    soft = np.exp(logits - lse[:, None])
    onehot = np.zeros_like(logits)
    np.put_along_axis(onehot, tsafe[:, None], 1, axis=1)
    dce = (soft - onehot) * in_mask[:, None]                        # d ce / d logits per pixel
    dlogit = np.zeros_like(logits)
    npix = h * H * W
    ce_orig = ce[:h].sum(dtype=np.float64) / npix                   # .mean() over ALL elements
    dlogit[:h] = w0 * dce[:h] / npix
    if select and 0.0 < ratio < 1.0:                                # :63-65, 90-117
        flat = ce[h:].reshape(-1)
        key = np.where(in_mask[h:].reshape(-1), flat, np.inf)
        total = int(in_mask[h:].sum())
        k = int(np.float32(ratio) * np.float32(total))             # python float * 0-dim int64 tensor -> float32
        if k > 0:
            sel = np.argsort(key, kind="stable")[:k]                # any k smallest; ties by index
            ce_aug = flat[sel].sum(dtype=np.float64) / k
            selmask = np.zeros(flat.shape, dtype=bool)
            selmask[sel] = True
            selmask = selmask.reshape(h, H, W)
            dlogit[h:] = w1 * dce[h:] * selmask[:, None] / k
            targets[h:][~selmask] = 255                             # :110-111
        else:
            ce_aug = 0.0
            targets[h:] = 255                                       # :115
    else:
        ce_aug = ce[h:].sum(dtype=np.float64) / npix                # :67-69
        dlogit[h:] = w1 * dce[h:] / npix
    loss = w0 * ce_orig + w1 * ce_aug
    # contrastive part, :119-147 -- masks are the ones computed BEFORE the mutation above
    first = np.zeros((B, 1, 1), dtype=bool)
    first[:h] = True
    m_orig = in_mask & first
    m_aug = in_mask & ~first
    idx_orig = np.flatnonzero(m_orig.reshape(-1))
    idx_aug = np.flatnonzero(m_aug.reshape(-1))
    idx_ood = np.flatnonzero(ood_mask.reshape(-1))
    n = min(int(B * H * W * sample_ratio), len(idx_ood), len(idx_orig), len(idx_aug))   # :149-156
    sflat = score.reshape(-1)
    dscore = np.zeros_like(sflat)
    po, pa, pd = (np.asarray(p)[:n] for p in perms)
    io, ia, id_ = idx_orig[po], idx_aug[pa], idx_ood[pd]
    with np.errstate(invalid="ignore", divide="ignore"):
        t_orig = sflat[io] + dt.type(margins[0]) - sflat[id_]
        t_aug = sflat[ia] + dt.type(margins[1]) - sflat[id_]
        c_orig = np.maximum(t_orig, 0).sum(dtype=np.float64) / n if n else np.nan      # mean of empty = nan
        c_aug = np.maximum(t_aug, 0).sum(dtype=np.float64) / n if n else np.nan
        if n:
            np.add.at(dscore, io, wc * (t_orig > 0) / n)
            np.add.at(dscore, id_, -wc * (t_orig > 0) / n)
            np.add.at(dscore, ia, wc * (t_aug > 0) / n)
            np.add.at(dscore, id_, -wc * (t_aug > 0) / n)
        same = in_mask[:h] & in_mask[h:]                             # :141
        t_in = score[h:2 * h] - score[:h] - dt.type(margins[2])
        ns = int(same.sum())
        c_in = (np.maximum(t_in, 0) * same).sum(dtype=np.float64) / ns if ns else np.nan
        if ns:
            gin = (wc * ((t_in > 0) & same) / ns).astype(dt)
            d2 = dscore.reshape(B, H, W)
            d2[h:2 * h] += gin
            d2[:h] -= gin
    loss = loss + wc * (c_orig + c_aug + c_in)
    return dict(loss=dt.type(loss), dlogit=dlogit.astype(dt), dscore=dscore.reshape(B, H, W).astype(dt),
                terms=np.array([ce_orig, ce_aug, c_orig, c_aug, c_in], dtype=np.float64))
