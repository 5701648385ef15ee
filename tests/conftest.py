import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
HERE = os.path.dirname(os.path.abspath(__file__))

_DP = {}   # the two data-parallel worker processes of tests/test_gpu_data_parallel.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_sessionstart(session):
    """The 2-rank data-parallel test needs two extra processes on the GPU.  They are started HERE, before this
    process has touched the GPU (a process that has initialised the GPU must not exec another program on the GPU
    boxes), and only when the GPU tests are selected; the test itself just waits for them."""
    markexpr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:      # counting devices does not initialise the GPU
            return
    except Exception:
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    outdir = tempfile.mkdtemp(prefix="hig_dp_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    logs = [open(os.path.join(outdir, "rank%d.log" % r), "wb") for r in range(2)]
    _DP["outdir"] = outdir
    _DP["procs"] = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), "2", port, outdir],
                                     env=env, stdout=logs[r], stderr=subprocess.STDOUT) for r in range(2)]


def pytest_sessionfinish(session, exitstatus):
    for p in _DP.get("procs", []):
        if p.poll() is None:
            p.kill()


@pytest.fixture(scope="session")
def dp_workers():
    """(outdir, [returncode, ...], [log text, ...]) of the two data-parallel workers, once they have finished."""
    if "procs" not in _DP:
        pytest.skip("data-parallel workers were not started (GPU tests not selected with -m gpu)")
    codes = [p.wait(timeout=900) for p in _DP["procs"]]
    texts = [open(os.path.join(_DP["outdir"], "rank%d.log" % r), errors="replace").read() for r in range(2)]
    return _DP["outdir"], codes, texts


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name))
    return load
