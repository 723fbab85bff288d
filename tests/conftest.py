import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def gpu_present():
    """True on a machine with a GPU, WITHOUT starting the HIP runtime in this process: counting devices does not
    open a context (torch.cuda.is_available() does)."""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


# ---- two data-parallel ranks on the one GPU of the test box ----------------------------------------------------------
# tests/test_dp_two_ranks_gpu.py checks what two ranks of tests/dp_worker.py leave behind.  The ranks run from
# pytest_collection_modifyitems, i.e. only when that test was selected, and this process does not open a GPU context
# before they have finished (test modules look for a GPU with gpu_present(), not torch.cuda.is_available()).
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_two_ranks(out_dir, steps=3, world=2):
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_worker.py"),
           str(out_dir), str(steps)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired as e:
        return dict(out_dir=str(out_dir), steps=steps, world=world, returncode=None,
                    log="timed out after 150 s\n" + str(e.stdout)[-2000:] + str(e.stderr)[-3000:])
    return dict(out_dir=str(out_dir), steps=steps, world=world, returncode=r.returncode,
                log=r.stdout[-3000:] + r.stderr[-4000:])


@pytest.hookimpl(trylast=True)          # after the builtin -m / -k deselection: `items` is what will actually run
def pytest_collection_modifyitems(config, items):
    config._dp_two_ranks = None
    if not any("test_dp_two_ranks_gpu" in it.nodeid for it in items) or not gpu_present():
        return
    import tempfile
    config._dp_tmp = tempfile.mkdtemp(prefix="dp_two_ranks_")
    config._dp_two_ranks = run_two_ranks(config._dp_tmp)


def pytest_unconfigure(config):
    tmp = getattr(config, "_dp_tmp", None)
    if tmp:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.fixture
def two_rank_run(request, tmp_path):
    got = getattr(request.config, "_dp_two_ranks", None)
    return got if got is not None else run_two_ranks(tmp_path)
