"""Host-side logic of unet_amd/trainer.py (no device work): the plateau scheduler against torch's own
ReduceLROnPlateau, the early-stopping loop against the oracle's restatement of training_utils.py:227-242, and the
metric formulas against the fixture captured from the reference's helpers.  CPU only."""
import os

import numpy as np
import pytest

from oracle import metrics as om

trainer = pytest.importorskip("unet_amd.trainer", reason="libunet_hip.so not built")


SEQUENCES = {
    "improving": [1.0 / (k + 1) for k in range(20)],
    "flat": [0.5] * 30,
    "plateau_then_drop": [0.9, 0.8] + [0.8] * 8 + [0.4] + [0.41] * 14,
    "tiny_gains": [0.5 * (1 - 5e-5) ** k for k in range(25)],       # below the 1e-4 relative threshold
    "noisy": list(0.6 + 0.3 * np.cos(np.arange(40) * 1.3) * np.exp(-np.arange(40) / 15)),
    "nan": [0.7, 0.6, float("nan")] + [0.65] * 10,
}


@pytest.mark.parametrize("name", sorted(SEQUENCES))
def test_plateau_lr_follows_torch(name):
    seq = SEQUENCES[name]
    state = {"lr": 1e-3}
    sched = trainer.PlateauLR(lambda: state["lr"], lambda v: state.__setitem__("lr", v), patience=5, factor=0.5)
    got = []
    for v in seq:
        sched.step(v)
        got.append(state["lr"])
    assert got == om.plateau_lrs(seq, lr=1e-3, patience=5, factor=0.5)


def test_plateau_lr_stops_at_eps():
    state = {"lr": 3e-8}
    sched = trainer.PlateauLR(lambda: state["lr"], lambda v: state.__setitem__("lr", v), patience=0, factor=0.5)
    seq = [1.0] * 6
    got = []
    for v in seq:
        sched.step(v)
        got.append(state["lr"])
    assert got == om.plateau_lrs(seq, lr=3e-8, patience=0, factor=0.5)


@pytest.mark.parametrize("name", sorted(SEQUENCES))
@pytest.mark.parametrize("patience", [1, 3, 10])
def test_fit_loop_early_stopping(name, patience):
    seq = SEQUENCES[name]
    saved, lrs = [], {"lr": 1e-3}
    epoch_box = {}

    def train_epoch(epoch):
        epoch_box["e"] = epoch
        return 1.0

    sched = trainer.PlateauLR(lambda: lrs["lr"], lambda v: lrs.__setitem__("lr", v))
    tr, va, best = trainer.fit_loop(len(seq), patience, train_epoch, lambda e: seq[e], sched,
                                    lambda: saved.append(epoch_box["e"]), log=lambda *_: None)
    n_run, want_saved = om.early_stop_epochs(seq, patience)
    assert len(va) == n_run and len(tr) == n_run
    assert saved == want_saved
    finite = [v for v in seq[:n_run] if v == v]
    assert best == min(finite)


def test_metric_formulas_match_reference_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "metrics_cases.npz"), allow_pickle=False)
    fold_keys, sample_keys = list(fx["fold_keys"]), list(fx["sample_keys"])
    for c in sorted(k[:-len(".logits")] for k in fx.files if k.endswith(".logits")):
        mask, pred = fx[c + ".mask"], fx[c + ".pred"]
        y_true = np.transpose(mask, (0, 2, 3, 1))
        got = trainer.calculate_metrics(y_true, pred)
        np.testing.assert_array_equal([got[k] for k in fold_keys], fx[c + ".fold"], err_msg=c)
        tp, tn, fp, fn = om.confusion_counts(y_true.squeeze(), pred.squeeze())
        got = trainer.metrics_from_counts(tp, tn, fp, fn)
        np.testing.assert_array_equal([got[k] for k in fold_keys], fx[c + ".fold"], err_msg=c)
        if c == "mask_not_binary":
            continue        # the per-sample report multiplies by the mask instead of comparing with it
        counts = [om.confusion_counts(mask[j], pred[j]) for j in range(len(mask))]
        per = trainer.sample_metrics_from_counts(counts)
        for i, k in enumerate(sample_keys):
            np.testing.assert_array_equal(per[k], fx[c + ".sample"][:, i], err_msg=f"{c}.{k}")


def test_byte_storage_is_chosen_only_when_it_is_exact():
    """the loaders' arrays (uint8 / 255 in fp32, data_helpers/utils.py:58-62; 0/1 masks) are stored as bytes, anything
    else stays fp32: the table lookup must give back every value bit for bit"""
    import numpy as np
    from unet_amd.data import quantize_exact, _LUT
    rng = np.random.default_rng(0)
    k = rng.integers(0, 256, size=(3, 8, 6, 3), dtype=np.uint8)
    img = np.array(k, dtype=np.float32) / 255.0                       # the loaders' expression
    q, ok = quantize_exact(img, chunk=2)
    assert ok and q.dtype == np.uint8 and np.array_equal(q, k)
    assert np.array_equal(_LUT[q].view(np.uint32), img.view(np.uint32))
    mask = (rng.random((3, 8, 6, 1)) > 0.5).astype(np.float32)
    q, ok = quantize_exact(mask)
    assert ok and set(np.unique(q)) <= {0, 255}
    assert quantize_exact(rng.random((2, 4, 4, 3), dtype=np.float32))[1] is False      # arbitrary floats
    bad = img.copy(); bad[1, 2, 3, 0] = np.nextafter(bad[1, 2, 3, 0], np.float32(2))   # one ulp off
    assert quantize_exact(bad)[1] is False
    for v in (np.nan, -0.25, 1.5):
        bad = img.copy(); bad[0, 0, 0, 0] = v
        assert quantize_exact(bad)[1] is False
    assert quantize_exact(k)[1] is False                               # uint8 arrays are not rescaled (see DeviceTensorSet)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` outside a launcher starts the N ranks as a child job and relays rank 0's one JSON
    line and the exit code (here with a stand-in rank program on gloo: the bench itself needs GPUs)."""
    import importlib.util
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = tmp_path / "rank_prog.py"
    prog.write_text(
        "import os, sys, json\n"
        "import torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.ones(1) * (dist.get_rank() + 1)\n"
        "dist.all_reduce(t)\n"
        "print('noise from rank', dist.get_rank(), flush=True)\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'n_gpus': dist.get_world_size(), 'sum': t.item(), 'argv': sys.argv[1:]}), flush=True)\n"
        "dist.destroy_process_group()\n"
        "sys.exit(3 if '--fail' in sys.argv else 0)\n")
    driver = ("import sys; sys.path.insert(0, %r); import bench; "
              "sys.exit(bench.launch_ranks(2, sys.argv[1:], script=%r))" % (root, str(prog)))
    r = subprocess.run([sys.executable, "-c", driver, "--steps", "2"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "sum": 3.0, "argv": ["--steps", "2"]}
    r = subprocess.run([sys.executable, "-c", driver, "--fail"], capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    # the launcher runs before torch is imported by the bench process itself
    src = open(os.path.join(root, "bench.py")).read()
    assert src.index("launch_ranks(args.gpus") < src.index("    import torch\n    import torch.distributed as dist")
