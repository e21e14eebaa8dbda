"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's pixel-level OOD metrics
(lib/utils/metric.py:87-127 fpr_and_fdr_at_recall, :130-153 get_measures, :170-180 eval_ood_measure).

sklearn 's roc_auc_score / average_precision_score are third-party arithmetic (reference pin: scikit-learn from
environment.yml; here 1.7.2): restated below from their published algorithm (_binary_clf_curve: stable
descending sort, thresholds at distinct scores, cumulative tps/fps; ROC area by trapezoids; AP = sum_k (R_k -
R_k-1) P_k). Pinned against tests/golden/ood_metrics.npz, written by tools/gen_golden.py from the reference's own
metric.py (loaded by file path) calling the installed sklearn."""
import numpy as np


def _binary_clf_curve(y_true, y_score):
    """tps, fps, thresholds at the distinct score values, scores descending (sklearn.metrics._ranking)."""
    order = np.argsort(y_score, kind="mergesort")[::-1]
    y_score = y_score[order]
    y_true = y_true[order]
    distinct = np.where(np.diff(y_score))[0]
    idx = np.r_[distinct, y_true.size - 1]
    tps = np.cumsum(y_true, dtype=np.float64)[idx]
    fps = 1 + idx - tps
    return tps, fps, y_score[idx]


def roc_auc(y_true, y_score):
    tps, fps, _ = _binary_clf_curve(y_true, y_score)
    tps = np.r_[0, tps]
    fps = np.r_[0, fps]
    fpr, tpr = fps / fps[-1], tps / tps[-1]
    return float(np.trapezoid(tpr, fpr))


def average_precision(y_true, y_score):
    tps, fps, _ = _binary_clf_curve(y_true, y_score)
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def fpr_at_recall(y_true, y_score, recall_level=0.95):
    """metric.py:87-127 with pos_label = 1."""
    tps, fps, _ = _binary_clf_curve(y_true, y_score)
    recall = tps / tps[-1]
    last_ind = tps.searchsorted(tps[-1])
    sl = slice(last_ind, None, -1)
    recall, fps = np.r_[recall[sl], 1], np.r_[fps[sl], 0]
    cutoff = np.argmin(np.abs(recall - recall_level))
    return float(fps[cutoff] / np.sum(np.logical_not(y_true)))


def eval_ood_measure(conf, seg_label, train_id_in=0, train_id_out=1, recall_level=0.95):
    """metric.py:170-180: (auroc, aupr, fpr) over the pixels labelled in/out, None when a class is empty."""
    in_scores = conf[seg_label == train_id_in]
    out_scores = conf[seg_label == train_id_out]
    if len(out_scores) == 0 or len(in_scores) == 0:
        return None
    examples = np.concatenate((out_scores.reshape(-1), in_scores.reshape(-1)))      # metric.py:130-137
    labels = np.zeros(len(examples), dtype=bool)
    labels[:out_scores.size] = True
    return roc_auc(labels, examples), average_precision(labels, examples), fpr_at_recall(labels, examples, recall_level)
