"""CPU oracle for the evaluation path: binarised predictions and segmentation metrics.  TEST INFRASTRUCTURE ONLY.

numpy / CPU-torch restatement of what the reference's evaluation helpers compute; nothing under ``unet_amd/``
imports it.  Parity pin: ``tests/golden/metrics_cases.npz``, written by ``tests/golden/make_golden_metrics.py``
from the reference's own functions in the build container; ``tests/test_oracle_golden.py`` replays it.

What is restated (reference file:line):
  * ``binarise``            trainer/k_fold.py:289-290, test_helper/utils.py:130-131  (sigmoid, then > 0.5)
  * ``confusion_counts``    trainer/k_fold.py:371-380   (exact == 1 / == 0 comparisons, integer counts)
  * ``fold_metrics``        trainer/k_fold.py:386-403   (eps 1e-8, float64 arithmetic on integer counts)
  * ``sample_metrics``      test_helper/utils.py:29-80  (fp32 sums of products, smooth 1e-6)
  * ``plateau_lrs``         trainer/training_utils.py:108,218 (ReduceLROnPlateau('min', patience 5, factor 0.5))
  * ``early_stop_epochs``   trainer/training_utils.py:113-114,227-242
"""
import numpy as np
import torch


def binarise(logits):
    """fp32 logits -> {0,1} fp32 predictions the way both evaluation loops do it."""
    return (torch.sigmoid(torch.as_tensor(logits, dtype=torch.float32)) > 0.5).float().numpy()


def confusion_counts(y_true, y_pred):
    """(tp, tn, fp, fn) as Python ints; values other than exactly 0 or 1 fall into no class."""
    y_true = np.asarray(y_true)
    y_pred = np.asarray(y_pred)
    tp = int(np.sum(np.logical_and(y_pred == 1, y_true == 1)))
    tn = int(np.sum(np.logical_and(y_pred == 0, y_true == 0)))
    fp = int(np.sum(np.logical_and(y_pred == 1, y_true == 0)))
    fn = int(np.sum(np.logical_and(y_pred == 0, y_true == 1)))
    return tp, tn, fp, fn


def fold_metrics(y_true, y_pred):
    tp, tn, fp, fn = (np.int64(v) for v in confusion_counts(y_true, y_pred))
    eps = 1e-8
    precision = tp / (tp + fp + eps)
    recall = tp / (tp + fn + eps)
    return {
        "dice": float((2 * tp) / (2 * tp + fp + fn + eps)),
        "iou": float(tp / (tp + fp + fn + eps)),
        "precision": float(precision),
        "recall": float(recall),
        "accuracy": float((tp + tn) / (tp + tn + fp + fn + eps)),
        "f1": float(2 * precision * recall / (precision + recall + eps)),
    }


def sample_metrics(pred, target, smooth=1e-6):
    """Per-sample metrics of the test report: every quantity is an fp32 tensor expression."""
    pred = torch.as_tensor(pred).float()
    target = torch.as_tensor(target).float()
    tp = (pred * target).sum()
    fp = (pred * (1 - target)).sum()
    tn = ((1 - pred) * (1 - target)).sum()
    fn = ((1 - pred) * target).sum()
    out = {
        "dice": (2 * tp + smooth) / (2 * tp + fp + fn + smooth),
        "iou": (tp + smooth) / (tp + fp + fn + smooth),
        "accuracy": (tp + tn + smooth) / (tp + fp + tn + fn + smooth),
        "precision": (tp + smooth) / (tp + fp + smooth),
        "recall": (tp + smooth) / (tp + fn + smooth),
        "specificity": (tn + smooth) / (tn + fp + smooth),
        "tp": tp, "fp": fp, "tn": tn, "fn": fn,
    }
    return {k: v.item() for k, v in out.items()}


def plateau_lrs(val_losses, lr=1e-3, patience=5, factor=0.5):
    """Learning rate in force after each epoch's scheduler.step(val_loss), from torch's own scheduler."""
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=lr)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", patience=patience, factor=factor)
    out = []
    for v in val_losses:
        sched.step(v)
        out.append(opt.param_groups[0]["lr"])
    return out


def early_stop_epochs(val_losses, patience):
    """(number of epochs run, epochs at which the best model was saved) for a sequence of validation losses."""
    best, counter, saved = float("inf"), 0, []
    for epoch, v in enumerate(val_losses):
        if v < best:
            best, counter = v, 0
            saved.append(epoch)
        else:
            counter += 1
        if counter >= patience:
            return epoch + 1, saved
    return len(val_losses), saved
