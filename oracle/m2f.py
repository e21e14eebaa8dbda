"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the Mask2Former anomaly score
(train_m2f.py:387-407): 1 - max_c sum_q softmax(cls)[b,q,c<C] * sigmoid(mask)[b,q,h,w].
Pinned against tests/golden/m2f_score.npz (reference arithmetic reproduced with torch ops in
tools/gen_golden.py because train_m2f.py itself needs detectron2 to import)."""
import numpy as np


def anomaly_score(class_logits, mask_logits, size):
    """class_logits [B,Q,C+1], mask_logits [B,Q,Hm,Wm], size (H,W) crop -> [B,H,W]."""
    m = class_logits.max(-1, keepdims=True)
    e = np.exp(class_logits - m)
    probs = (e / e.sum(-1, keepdims=True))[..., :-1]
    sig = 1.0 / (1.0 + np.exp(-mask_logits))
    u = np.einsum("bqc,bqhw->bchw", probs, sig, optimize=True)
    u = u[:, :, :size[0], :size[1]]
    return (1 - u.max(axis=1)).astype(class_logits.dtype)
