"""Repeated k-fold over ranks (unet_amd/kfold.py) on CPU, world_size 2, gloo backend.  The trainings themselves need
a GPU; here they are replaced by stubs that return a deterministic score per fold, so what is checked is the part that
runs on every machine: the folds are the reference's (KFold(shuffle=True, random_state=42 + t),
data_helpers/utils.py:157-161), job j runs on rank j mod world and nowhere else, every rank ends with the same
table in the sequential order, and the best model is the one a sequential run would have kept (first strict
maximum of the validation Dice, trainer/k_fold.py:137-141)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Stub(torch.nn.Module):
    def __init__(self, tag):
        super().__init__()
        self.tag = torch.nn.Parameter(torch.tensor(float(tag)))


def _score(val_images):
    """A Dice-like score that depends only on which samples are in the validation fold; two folds tie on purpose."""
    ids = sorted(int(v) for v in val_images[:, 0, 0, 0])
    return 0.9 if ids[0] in (0, 1) else 0.5 + (sum(ids) % 7) / 100.0


def _run(rank, world, out_dir, log_jobs):
    from unet_amd import kfold
    n = 23
    images = np.arange(n, dtype=np.float32).reshape(n, 1, 1, 1) * np.ones((1, 4, 4, 1), np.float32)
    masks = np.zeros((n, 4, 4, 1), np.float32)

    def train_fn(train_data, val_data, **kw):
        assert kw["data_parallel"] is False
        assert len(train_data[0]) + len(val_data[0]) == n
        assert not set(train_data[0][:, 0, 0, 0]) & set(val_data[0][:, 0, 0, 0])
        log_jobs.append(sorted(int(v) for v in val_data[0][:, 0, 0, 0]))
        return _Stub(_score(val_data[0]))

    def eval_fn(model, data, device, batch_size=8):
        s = float(model.tag.item()) if len(data[0]) != 5 else 0.25 + float(model.tag.item()) / 10      # 5 = the test set
        return {"dice": s, "iou": s / 2, "precision": s, "recall": s, "accuracy": s, "f1": s}

    test = (np.zeros((5, 4, 4, 1), np.float32), np.zeros((5, 4, 4, 1), np.float32))
    cwd = os.getcwd()
    os.chdir(out_dir)
    try:
        rec, path, best = kfold.train_kfold(images, masks, test_data=test, k_fold=4, times=2, epochs=1,
                                            train_fn=train_fn, eval_fn=eval_fn, device="cpu", log=lambda *_: None,
                                            best_model_path=os.path.join(out_dir, "best.pth"))
    finally:
        os.chdir(cwd)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"records": [{k: v for k, v in r.items() if k != "Training Time"} for r in rec], "best": best,
                   "val_sets": log_jobs}, f)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _run(rank, world, out_dir, [])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_kfold_jobs_spread_over_two_ranks(tmp_path):
    pytest.importorskip("unet_amd.kfold", reason="libunet_hip.so not built")
    from sklearn.model_selection import KFold
    from unet_amd import kfold
    world = 2
    two = tmp_path / "two"
    one = tmp_path / "one"
    two.mkdir()
    one.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), str(two)), nprocs=world, join=True)
    _run(0, 1, str(one), [])                                                     # the sequential run, one process
    seq = json.load(open(one / "rank0.json"))
    r0, r1 = json.load(open(two / "rank0.json")), json.load(open(two / "rank1.json"))
    assert r0["records"] == r1["records"] == seq["records"]
    assert r0["best"] == r1["best"] == seq["best"]
    assert [(r["Time"], r["Fold"]) for r in seq["records"]] == [(t, f) for t in (1, 2) for f in (1, 2, 3, 4)]
    # the folds are sklearn's with seeds 42, 43; ranks took alternate jobs
    want = []
    for t in range(2):
        for _, va in KFold(n_splits=4, shuffle=True, random_state=42 + t).split(np.zeros((23, 1))):
            want.append(sorted(int(v) for v in va))
    assert seq["val_sets"] == want
    assert r0["val_sets"] == want[0::2] and r1["val_sets"] == want[1::2]
    assert [sorted(int(v) for v in j[3]) for j in kfold.kfold_jobs(23, 4, 2)] == want
    # ties: the earliest job with the maximum Dice is kept, on whichever rank it ran
    dice = [r["Dice"] for r in seq["records"]]
    first_best = dice.index(max(dice))
    assert dice.count(max(dice)) >= 2
    for d in (one, two):
        sd = torch.load(d / "best.pth", weights_only=True)
        assert sd["tag"].item() == pytest.approx(dice[first_best])
    assert seq["best"]["dice"] == pytest.approx(0.25 + dice[first_best] / 10)
    # the table on disk and its summary
    import pandas as pd
    df = pd.read_csv(two / "UNet_4fold_2times_results.csv")
    assert list(df.columns) == ["Time", "Fold", "Model", "Dice", "IoU", "Precision", "Recall", "Accuracy", "F1", "Training Time"]
    per_time, mean, std = kfold.summarize(df.to_dict("records"))
    assert mean["Dice"] == pytest.approx(df["Dice"].mean()) and std["Dice"] == pytest.approx(df["Dice"].std())
    assert per_time[2]["IoU"] == pytest.approx(df[df.Time == 2]["IoU"].mean())
