"""Repeated k-fold cross-validation over the GPUs of one node: independent trainings, no data-path collective.

The reference runs its times x k trainings one after another on one device (trainer/k_fold.py:15-235, folds from
data_helpers/utils.py:151-175).  The folds do not depend on each other, so here job j = (repetition t, fold f)
runs on rank j mod world, each rank training its model alone on its own GPU; only the small result records are
exchanged (as Python objects) at the end.  Launch one process per GPU with torch.distributed.run, or call it in a
single process to run every fold on one GPU.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .trainer import train_model, evaluate_model


def kfold_jobs(n_samples, k_fold=5, times=3):
    """[(t, fold_idx, train_index, val_index)] in the reference's order: KFold(n_splits=k_fold, shuffle=True,
    random_state=42 + t) for t in range(times) (data_helpers/utils.py:157-161)."""
    from sklearn.model_selection import KFold
    jobs = []
    for t in range(times):
        kf = KFold(n_splits=k_fold, shuffle=True, random_state=42 + t)
        for fold_idx, (train_index, val_index) in enumerate(kf.split(np.zeros((n_samples, 1)))):
            jobs.append((t, fold_idx, train_index, val_index))
    return jobs


def summarize(records):
    """Means per repetition and overall mean / sample standard deviation of every metric column
    (k_fold.py:196-215; pandas' std is the n-1 form)."""
    cols = ["Dice", "IoU", "Precision", "Recall", "Accuracy", "F1", "Training Time"]
    per_time = {}
    for t in sorted({r["Time"] for r in records}):
        rows = [r for r in records if r["Time"] == t]
        per_time[t] = {c: float(np.mean([r[c] for r in rows])) for c in cols}
    overall_mean = {c: float(np.mean([r[c] for r in records])) for c in cols}
    overall_std = {c: float(np.std([r[c] for r in records], ddof=1)) if len(records) > 1 else float("nan") for c in cols}
    return per_time, overall_mean, overall_std


def train_kfold(images, masks, test_data=None, model_type="UNet", k_fold=5, times=3, epochs=100, batch_size=8,
                learning_rate=0.001, eval_batch_size=1, custom_loss=None, results_file=None, best_model_path=None,
                train_fn=train_model, eval_fn=evaluate_model, device=None, log=print):
    """trainer/k_fold.py:15 on in-memory arrays (images [N,H,W,C], masks [N,H,W,1]; the reference's file loaders
    are not part of this package).  Returns (records, best_model_path, best_test_metrics): `records` is the table
    the reference writes to `{model_type}_{k}fold_{times}times_results.csv`, identical on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if device is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    best_model_path = best_model_path or f"best_{model_type.lower()}_kfold_model.pth"
    jobs = kfold_jobs(len(images), k_fold, times)
    mine, best_dice, best_state = [], 0.0, None
    for j, (t, fold_idx, tr, va) in enumerate(jobs):
        if j % world != rank:
            continue
        log(f"[rank {rank}] repetition {t + 1}, fold {fold_idx + 1}/{k_fold}")
        train_data, val_data = (images[tr], masks[tr]), (images[va], masks[va])
        t0 = time.time()
        model = train_fn(train_data, val_data, model_type=model_type, epochs=epochs, batch_size=batch_size,
                         learning_rate=learning_rate, device=device, custom_loss=custom_loss,
                         model_save_path=f"{best_model_path}.job{j}", data_parallel=False)
        train_time = time.time() - t0
        if os.path.exists(f"{best_model_path}.job{j}"):
            os.remove(f"{best_model_path}.job{j}")                           # the returned model already holds it
        m = eval_fn(model, val_data, device, batch_size=eval_batch_size)
        mine.append({"Time": t + 1, "Fold": fold_idx + 1, "Model": model_type, "Dice": m["dice"], "IoU": m["iou"],
                     "Precision": m["precision"], "Recall": m["recall"], "Accuracy": m["accuracy"], "F1": m["f1"],
                     "Training Time": train_time, "_job": j})
        if m["dice"] > best_dice:                                            # k_fold.py:137-141
            best_dice = m["dice"]
            best_state = (j, model)
        del model
    # exchange the records; every rank ends with the same table in the reference's order
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, best_dice, best_state[0] if best_state else -1))
    else:
        gathered = [(mine, best_dice, best_state[0] if best_state else -1)]
    records = sorted((r for g in gathered for r in g[0]), key=lambda r: r["_job"])
    # the best fold overall; the earliest job wins a tie, as in a sequential run (strict > at k_fold.py:137)
    best_job, top = -1, 0.0
    for r in records:
        if r["Dice"] > top:
            top, best_job = r["Dice"], r["_job"]
    best_test_metrics = None
    owner = best_job % world if best_job >= 0 else 0
    if best_job >= 0 and rank == owner:
        model = best_state[1]
        torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, best_model_path)
        if test_data is not None:
            best_test_metrics = eval_fn(model, test_data, device, batch_size=max(1, eval_batch_size // 2))
    if world > 1:
        box = [best_test_metrics]
        dist.broadcast_object_list(box, src=owner)
        best_test_metrics = box[0]
    for r in records:
        r.pop("_job")
    if rank == 0:
        import pandas as pd
        results_file = results_file or f"{model_type}_{k_fold}fold_{times}times_results.csv"
        pd.DataFrame(records).to_csv(results_file, index=False)
        per_time, mean, std = summarize(records)
        for t, row in per_time.items():
            log(f"repetition {t}: Dice={row['Dice']:.4f}, IoU={row['IoU']:.4f}")
        for c in ("Dice", "IoU", "Precision", "Recall", "Accuracy", "F1"):
            log(f"{c}: {mean[c]:.4f} +- {std[c]:.4f}")
        if best_test_metrics is not None:
            log("best model on the test data: " + ", ".join(f"{k}={v:.4f}" for k, v in best_test_metrics.items()))
    return records, best_model_path, best_test_metrics
