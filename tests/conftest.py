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


# ---- two data-parallel ranks on the one GPU of the test box ----------------------------------------------------------
# On this box a rank stalls inside a device copy as soon as a THIRD process holds a context on the same GPU (seen with
# an unrelated idle process as well as with a pytest process that has loaded the kernel library).  The two-rank run of
# tests/test_dp_two_ranks_gpu.py therefore happens here, at session start, before any test module (and with it the
# kernel library) has been imported into the pytest process; the test then checks what the ranks left behind.
def run_two_ranks(out_dir, steps=3, world=2, port=29531):
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"),
           str(out_dir), str(steps)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=100)
    except subprocess.TimeoutExpired:
        return dict(out_dir=str(out_dir), steps=steps, world=world, returncode=None, log="timed out after 100 s")
    return dict(out_dir=str(out_dir), steps=steps, world=world, returncode=r.returncode,
                log=r.stdout[-2000:] + r.stderr[-4000:])


def pytest_sessionstart(session):
    session.config._dp_two_ranks = None
    expr = session.config.getoption("markexpr", "") or ""
    names = [str(a) for a in session.config.args]
    if "not gpu" in expr or (any(n.endswith(".py") or "::" in n for n in names)
                             and not any("test_dp_two_ranks_gpu" in n for n in names)):
        return                                   # the GPU tests, or this one, were not asked for
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    import tempfile
    session.config._dp_two_ranks = run_two_ranks(tempfile.mkdtemp(prefix="dp_two_ranks_"))


@pytest.fixture
def two_rank_run(request, tmp_path):
    got = getattr(request.config, "_dp_two_ranks", None)
    return got if got is not None else run_two_ranks(tmp_path)
