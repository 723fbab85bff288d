#!/usr/bin/env python3
"""Write metrics_cases.npz by IMPORTING THE REFERENCE's evaluation helpers (build container only):

    python tests/golden/make_golden_metrics.py

Each case stores fp32 logits, a mask, the reference's binarised predictions, and the metric dictionaries of
trainer/k_fold.py::calculate_metrics (whole array) and test_helper/utils.py::calculate_metrics (per sample).
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from trainer.k_fold import calculate_metrics as fold_metrics        # noqa: E402  (reference)
from test_helper.utils import calculate_metrics as sample_metrics   # noqa: E402  (reference)

FOLD_KEYS = ["dice", "iou", "precision", "recall", "accuracy", "f1"]
SAMPLE_KEYS = ["dice", "iou", "accuracy", "precision", "recall", "specificity", "tp", "fp", "tn", "fn"]


def cases():
    g = torch.Generator().manual_seed(77)
    logits = torch.randn(4, 1, 24, 40, generator=g) * 3
    logits[0, 0, :2] = 0.0                                   # sigmoid(0) = 0.5 is NOT > 0.5
    logits[1, 0, 0, :8] = torch.tensor([1e-3, -1e-3, 1e-5, -1e-5, 30.0, -30.0, 100.0, -100.0])
    mask = (torch.rand(4, 1, 24, 40, generator=g) > 0.6).float()
    yield "random", logits, mask
    yield "all_background", -torch.rand(2, 1, 16, 16, generator=g) - 0.1, torch.zeros(2, 1, 16, 16)
    yield "all_foreground", torch.rand(2, 1, 16, 16, generator=g) + 0.1, torch.ones(2, 1, 16, 16)
    yield "prediction_empty", -torch.rand(1, 1, 16, 16, generator=g) - 0.1, (torch.rand(1, 1, 16, 16, generator=g) > 0.5).float()
    soft = torch.rand(2, 1, 16, 16, generator=g)
    soft = torch.where(soft < 0.3, torch.zeros(()), torch.where(soft > 0.7, torch.ones(()), soft))
    yield "mask_not_binary", torch.randn(2, 1, 16, 16, generator=g), soft
    yield "single_2d", torch.randn(1, 1, 32, 32, generator=g), (torch.rand(1, 1, 32, 32, generator=g) > 0.5).float()


def main():
    out = {}
    for name, logits, mask in cases():
        pred = (torch.sigmoid(logits) > 0.5).float()                      # k_fold.py:289-290
        m_np = mask.permute(0, 2, 3, 1).numpy()                           # masks arrive as [N,H,W,1] arrays
        p_np = pred.numpy()
        y_true = m_np.squeeze() if m_np.ndim > 2 else m_np                # k_fold.py:341-342
        y_pred = p_np.squeeze() if p_np.ndim > 2 else p_np
        fm = fold_metrics(y_true, y_pred)
        out[f"{name}.logits"] = logits.numpy()
        out[f"{name}.mask"] = mask.numpy()
        out[f"{name}.pred"] = pred.numpy()
        out[f"{name}.fold"] = np.array([fm[k] for k in FOLD_KEYS], dtype=np.float64)
        per = []
        for j in range(len(logits)):                                      # test_helper/utils.py:134-137
            sm = sample_metrics(pred[j], mask[j])
            per.append([sm[k] for k in SAMPLE_KEYS])
        out[f"{name}.sample"] = np.array(per, dtype=np.float64)
    out["fold_keys"] = np.array(FOLD_KEYS)
    out["sample_keys"] = np.array(SAMPLE_KEYS)
    np.savez_compressed(os.path.join(HERE, "metrics_cases.npz"), **out)
    print("wrote metrics_cases.npz:", sorted(k for k in out if k.endswith(".fold")))


if __name__ == "__main__":
    main()
