"""Training and evaluation drivers with the reference's names and signatures, kept on the device.

    train_model       trainer/training_utils.py:11-251
    evaluate_model    trainer/k_fold.py:237-351
    calculate_metrics trainer/k_fold.py:353-403
    save_results      test_helper/utils.py:82-196

What is different from the reference, and why: the data set is uploaded once and batches are gathered on the
device (the reference's three DataLoader workers collate fp32 NCHW batches on the host, training_utils.py:62-84,
which cannot feed thousands of images per second); losses are accumulated on the device and read back once per
epoch instead of once per step (training_utils.py:170); predictions are binarised and counted by a kernel
(`uh_confusion_counts`) instead of being copied to the host per batch (k_fold.py:290).  The arithmetic of every
reported number is the reference's.
"""
import math
import time

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .model_utils import create_unet, create_lunext
from .cl_dice import soft_dice_cldice
from .data import DeviceDataset


# --------------------------------------------------------------------------------------------------------------
# host-side pieces of the loop (no device work; unit-tested on their own)
# --------------------------------------------------------------------------------------------------------------
class PlateauLR:
    """optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', patience=5, factor=0.5) as
    training_utils.py:108 builds it (relative threshold 1e-4, no cooldown, min_lr 0, eps 1e-8), acting on a
    learning rate reached through a getter and a setter so that it serves torch optimizers and FlatAdam alike."""

    def __init__(self, get_lr, set_lr, patience=5, factor=0.5, threshold=1e-4, min_lr=0.0, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.get_lr, self.set_lr = get_lr, set_lr
        self.patience, self.factor, self.threshold, self.min_lr, self.eps = patience, factor, threshold, min_lr, eps
        self.best = math.inf
        self.num_bad_epochs = 0

    def step(self, metric):
        current = float(metric)
        if current < self.best * (1.0 - self.threshold):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            old = float(self.get_lr())
            new = max(old * self.factor, self.min_lr)
            if old - new > self.eps:
                self.set_lr(new)
            self.num_bad_epochs = 0


def rank0_value(value, group=None):
    """`value` (a float) as rank 0 holds it, on every rank of the process group; the value itself without a group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    box = [value]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return box[0]


def fit_loop(epochs, patience, train_epoch, validate, scheduler, save_best, log=print, model_type="UNet", get_lr=None,
             agree=None):
    """The epoch loop of training_utils.py:125-242: train, validate, scheduler.step(val), keep the best
    checkpoint, stop after `patience` epochs without a new best.  Returns (train_losses, val_losses, best).
    agree(value) -> value: in data-parallel training every decision below (learning-rate schedule, best checkpoint,
    early stop) must fall the same way on every rank, or the ranks stop meeting in their collectives; `agree` replaces
    the local validation loss by one all ranks share (rank 0's)."""
    train_losses, val_losses = [], []
    best_val_loss, patience_counter = float("inf"), 0
    for epoch in range(epochs):
        avg_train_loss = train_epoch(epoch)
        train_losses.append(avg_train_loss)
        avg_val_loss = validate(epoch)
        if agree is not None:
            avg_val_loss = agree(avg_val_loss)
        val_losses.append(avg_val_loss)
        scheduler.step(avg_val_loss)
        log(f"Epoch {epoch + 1}/{epochs}:")
        log(f"  Train Loss: {avg_train_loss:.4f}")
        log(f"  Val Loss: {avg_val_loss:.4f}")
        if get_lr is not None:
            log(f"  Learning Rate: {get_lr():.6f}")
        if avg_val_loss < best_val_loss:
            best_val_loss, patience_counter = avg_val_loss, 0
            save_best()
            log(f"  new best {model_type} model saved (Val Loss: {best_val_loss:.4f})")
        else:
            patience_counter += 1
            log(f"  validation loss did not improve ({patience_counter}/{patience})")
        log("-" * 50)
        if patience_counter >= patience:
            log(f"early stop at epoch {epoch + 1}")
            break
    return train_losses, val_losses, best_val_loss


def metrics_from_counts(tp, tn, fp, fn):
    """k_fold.py:386-403 on integer confusion counts."""
    tp, tn, fp, fn = (np.int64(v) for v in (tp, tn, fp, fn))
    eps = 1e-8
    dice = (2 * tp) / (2 * tp + fp + fn + eps)
    iou = tp / (tp + fp + fn + eps)
    precision = tp / (tp + fp + eps)
    recall = tp / (tp + fn + eps)
    accuracy = (tp + tn) / (tp + tn + fp + fn + eps)
    f1 = 2 * precision * recall / (precision + recall + eps)
    return {"dice": float(dice), "iou": float(iou), "precision": float(precision), "recall": float(recall),
            "accuracy": float(accuracy), "f1": float(f1)}


def calculate_metrics(y_true, y_pred):
    """k_fold.py:353-403 for host arrays (the device path goes through `uh_confusion_counts`)."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    if y_true.ndim > 2:
        y_true = y_true.squeeze()
    if y_pred.ndim > 2:
        y_pred = y_pred.squeeze()
    tp = np.sum(np.logical_and(y_pred == 1, y_true == 1))
    tn = np.sum(np.logical_and(y_pred == 0, y_true == 0))
    fp = np.sum(np.logical_and(y_pred == 1, y_true == 0))
    fn = np.sum(np.logical_and(y_pred == 0, y_true == 1))
    return metrics_from_counts(tp, tn, fp, fn)


def sample_metrics_from_counts(counts, smooth=1e-6):
    """test_helper/utils.py:29-80 for each row (tp, tn, fp, fn) of an integer count table: the reference forms
    every quantity as an fp32 tensor expression, so the same is done here.  Returns a dict of fp64 arrays."""
    c = torch.as_tensor(np.asarray(counts, dtype=np.int64)).to(torch.float32)
    tp, tn, fp, fn = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    out = {
        "dice": (2 * tp + smooth) / (2 * tp + fp + fn + smooth),
        "iou": (tp + smooth) / (tp + fp + fn + smooth),
        "accuracy": (tp + tn + smooth) / (tp + fp + tn + fn + smooth),
        "precision": (tp + smooth) / (tp + fp + smooth),
        "recall": (tp + smooth) / (tp + fn + smooth),
        "specificity": (tn + smooth) / (tn + fp + smooth),
        "tp": tp, "fp": fp, "tn": tn, "fn": fn,
    }
    return {k: v.double().numpy() for k, v in out.items()}


# --------------------------------------------------------------------------------------------------------------
# device-side helpers
# --------------------------------------------------------------------------------------------------------------
def _device(device):
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError("unet_amd runs HIP kernels only: a GPU device is required")
    return dev


def confusion_counts(logits, masks, counts, per_sample=False):
    """counts (int64, device): [4] (or [b,4] with per_sample) += tp, tn, fp, fn of sigmoid(logits) > 0.5 against
    the masks.  logits and masks: same number of elements, fp32, on the device."""
    logits = logits.float().contiguous()
    masks = masks.float().contiguous()
    if logits.numel() != masks.numel():
        raise ValueError(f"predictions {tuple(logits.shape)} and masks {tuple(masks.shape)} differ in size")
    planes = logits.shape[0] if per_sample else 1
    if counts.dtype != torch.int64 or not counts.is_contiguous() or counts.numel() != planes * 4:
        raise ValueError("counts must be a contiguous int64 tensor of 4 values per plane")
    _lib.call("uh_confusion_counts", planes, logits.numel() // planes, logits.data_ptr(), masks.data_ptr(),
              counts.data_ptr(), torch.cuda.current_stream(logits.device).cuda_stream)
    return counts


def _make_loss(custom_loss, model=None):
    """training_utils.py:100-105 and the glue in front of the loss (:142-150): a function (logits, masks) -> loss."""
    if custom_loss == "self_reg":
        if model is None or not hasattr(model, "intermediate_features"):
            raise ValueError("custom_loss='self_reg' needs a model with intermediate_features (the U-Net)")
        from .self_reg import SelfRegLoss
        reg = SelfRegLoss()

        def self_reg_loss(logits, masks):
            return reg(logits, masks, model.intermediate_features)      # the list the last forward pass left
        return self_reg_loss
    if custom_loss == "cl_dice":
        crit = soft_dice_cldice()

        def cldice_loss(logits, masks):
            probs = torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6)
            return crit(masks, probs)                       # argument order (y_true, y_pred)
        return cldice_loss
    return nn.BCEWithLogitsLoss()


def train_model(train_data, validation_data, model_type="UNet", epochs=50, batch_size=16, learning_rate=0.001,
                patience=10, device=None, custom_loss=None, model_save_path=None, shuffle_seed=0, log=print,
                data_parallel=True):
    """trainer/training_utils.py:11.  Same arguments (plus where to keep the best checkpoint, the shuffle seed
    and a log sink); returns the model with the best-validation weights loaded, in eval mode.  Inside an
    initialised process group a U-Net trains data-parallel over its ranks unless data_parallel=False."""
    from .dp import DataParallelTrainer
    dev = _device(device)
    torch.cuda.set_device(dev)
    train_images, train_masks = train_data
    val_images, val_masks = validation_data
    in_channels = train_images.shape[-1] if len(train_images) > 0 else 3       # training_utils.py:87
    # the data set goes to the device once, in the loaders' layout (bytes where exact); batches are gathered there
    ds_tr = DeviceDataset(train_images, train_masks, dev)
    ds_va = DeviceDataset(val_images, val_masks, dev)
    if model_type == "UNet":
        model = create_unet(in_channels=in_channels, out_channels=1)
        save_path = model_save_path or "best_unet_model.pth"
    else:
        model = create_lunext(in_channels=in_channels, out_channels=1)
        save_path = model_save_path or "best_lunext_model.pth"
    model = model.to(dev)
    loss_fn = _make_loss(custom_loss, model)

    if model_type == "UNet":
        stepper = DataParallelTrainer(model, lr=learning_rate, criterion=loss_fn, standalone=not data_parallel)
        opt = stepper.optimizer
        get_lr = lambda: opt.lr                                              # noqa: E731
        set_lr = lambda v: setattr(opt, "lr", v)                             # noqa: E731
        train_step = stepper.train_step
        rank, world = stepper.rank, stepper.world
    else:
        from .dp import LUNeXtTrainer
        if custom_loss == "cl_dice":
            crit = "cl_dice"                       # sigmoid + clamp + clDice, value and gradient from the kernels
        elif custom_loss is None:
            crit = "bce"
        else:
            crit = loss_fn                         # any other criterion: differentiated by autograd at the logits
        stepper = LUNeXtTrainer(model, lr=learning_rate, criterion=crit, standalone=not data_parallel)
        opt = stepper.optimizer
        get_lr = lambda: opt.lr                                              # noqa: E731
        set_lr = lambda v: setattr(opt, "lr", v)                             # noqa: E731
        train_step = stepper.train_step
        rank, world = stepper.rank, stepper.world
    scheduler = PlateauLR(get_lr, set_lr, patience=5, factor=0.5)

    n_train, n_val = len(ds_tr), len(ds_va)
    log(f"training {model_type}: {n_train} training / {n_val} validation samples, batch {batch_size}, lr {learning_rate}")

    def train_epoch(epoch):
        perm = DataParallelTrainer.shard(n_train, epoch, rank, world, seed=shuffle_seed).to(dev)
        total = torch.zeros((), dtype=torch.float64, device=dev)
        n_batches = 0
        for i in range(0, len(perm), batch_size):
            total += train_step(*ds_tr.batch(perm[i:i + batch_size])).double()
            n_batches += 1
        return total.item() / max(n_batches, 1)

    def validate(epoch):
        if world > 1:
            # BatchNorm statistics are per replica (as under DistributedDataParallel): validate with rank 0's on every
            # rank, so that all ranks judge the same model (it is also the one rank 0 saves as the best checkpoint)
            for b in model.buffers():
                torch.distributed.broadcast(b.data, src=0)
        model.eval()
        total = torch.zeros((), dtype=torch.float64, device=dev)
        n_batches = 0
        with torch.no_grad():
            for i in range(0, n_val, batch_size):
                xb, mb = ds_va.slice(i, i + batch_size)
                total += loss_fn(model(xb), mb).double()
                n_batches += 1
        return total.item() / max(n_batches, 1)

    def save_best():
        if rank == 0:
            # parameters may be views into one flat buffer: store independent tensors
            torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, save_path)

    _, _, best = fit_loop(epochs, patience, train_epoch, validate, scheduler, save_best, log=log,
                          model_type=model_type, get_lr=get_lr, agree=rank0_value if world > 1 else None)
    if world > 1:
        torch.distributed.barrier()
    model.load_state_dict(torch.load(save_path, map_location=dev, weights_only=True))
    model.eval()
    log(f"{model_type} training finished; best validation loss {best:.4f}")
    return model


def evaluate_model(model, data, device=None, batch_size=8):
    """trainer/k_fold.py:237.  data = (images [N,H,W,C], masks [N,H,W,1] or [N,H,W]); returns the dict of
    k_fold.py:396-403.  One 32-byte read-back for the whole set."""
    dev = _device(device)
    images, masks = data
    model.eval()
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    total = len(images)
    chunk = min(100, total)                                 # k_fold.py:261
    with torch.no_grad():
        for i in range(0, total, max(chunk, 1)):
            ds = DeviceDataset(images[i:i + chunk], masks[i:i + chunk], dev)
            for j in range(0, len(ds), batch_size):
                x, m = ds.slice(j, j + batch_size)
                if hasattr(model, "predict_counts"):        # the U-Net's head launch binarises and counts as well
                    model.predict_counts(x, m, counts)
                else:
                    confusion_counts(model(x), m, counts)
    tp, tn, fp, fn = counts.cpu().tolist()
    return metrics_from_counts(tp, tn, fp, fn)


SAMPLE_KEYS = ("dice", "iou", "accuracy", "precision", "recall", "specificity", "tp", "fp", "tn", "fn")


def save_results(model, test_data, train_time, filepath, batch_size=1, log=print):
    """test_helper/utils.py:82: per-sample metrics averaged over the test set, BCE-with-logits loss averaged over
    batches (batch 1 as there), written as a text report.  Returns the averaged metrics (the reference returns
    nothing).  Masks are expected in {0, 1}, which is what the reference's loaders produce."""
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("unet_amd runs HIP kernels only: move the model to a GPU first")
    model.eval()
    test_images, test_masks = test_data
    n = len(test_images)
    if n == 0:
        log("warning: no test data")
        return None
    ds = DeviceDataset(test_images, test_masks, dev)
    counts = torch.zeros(n, 4, dtype=torch.int64, device=dev)
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    criterion = nn.BCEWithLogitsLoss()
    num_batches = 0
    with torch.no_grad():
        for i in range(0, n, batch_size):
            xb, mb = ds.slice(i, i + batch_size)
            if hasattr(model, "predict_counts"):
                logits = model.predict_counts(xb, mb, counts[i:i + batch_size], per_sample=True, return_logits=True)
            else:
                logits = model(xb)
                confusion_counts(logits, mb, counts[i:i + batch_size], per_sample=True)
            loss_sum += criterion(logits, mb).double()
            num_batches += 1
    per = sample_metrics_from_counts(counts.cpu().numpy())
    avg = {k: float(per[k].sum() / n) for k in SAMPLE_KEYS}
    avg_loss = loss_sum.item() / num_batches
    lines = ["UNet model test results", "=" * 50, "",
             f"training time: {train_time:.2f} s", f"test samples: {n}", f"device: {dev}",
             f"mean test loss: {avg_loss:.6f}", "", "metrics:", "-" * 30,
             f"Dice Coefficient (DC): {avg['dice']:.6f}", f"Intersection over Union (IoU): {avg['iou']:.6f}",
             f"Accuracy: {avg['accuracy']:.6f}", f"Precision: {avg['precision']:.6f}",
             f"Recall (Sensitivity): {avg['recall']:.6f}", f"Specificity: {avg['specificity']:.6f}", "",
             "confusion matrix (mean per sample):", "-" * 30,
             f"True Positive (TP): {avg['tp']:.2f}", f"False Positive (FP): {avg['fp']:.2f}",
             f"True Negative (TN): {avg['tn']:.2f}", f"False Negative (FN): {avg['fn']:.2f}", "",
             f"F1-Score: {avg['dice']:.6f}", f"Jaccard Index: {avg['iou']:.6f}"]
    with open(filepath, "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    log(f"evaluated {n} samples: loss {avg_loss:.6f}, dice {avg['dice']:.6f}, iou {avg['iou']:.6f}; report: {filepath}")
    avg["loss"] = avg_loss
    return avg


def timed(fn, *args, **kwargs):
    """(result, seconds) with the device drained on both sides: the train_time argument of save_results."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn(*args, **kwargs)
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0
