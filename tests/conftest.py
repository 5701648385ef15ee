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

_DP = {}   # worker process groups of the multi-process GPU tests: tag -> {outdir, n, procs}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    # The CPU oracle runs on torch's intra-op thread pool.  The GPU boxes expose 256 host cores: with one thread per core the
    # many SMALL oracle problems of the suite spend their time synchronising (the evaluator's 6-pair oracle: 40 s there, 2 s on
    # 8 cores); 16 threads is also what bench.py's doubling probe finds best for the config-2 forward.  (CPU-only call.)
    try:
        import torch
        if (os.cpu_count() or 1) > 16 and "OMP_NUM_THREADS" not in os.environ:
            torch.set_num_threads(16)
    except Exception:
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    return port


def _spawn(tag, script, ranks, extra_env=None):
    outdir = tempfile.mkdtemp(prefix="hig_%s_" % tag)
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    logs = [open(os.path.join(outdir, "rank%d.log" % r), "wb") for r in range(ranks)]
    argv = (lambda r: [str(r), str(ranks), port, outdir]) if ranks > 1 else (lambda r: [port, outdir])
    _DP[tag] = {"outdir": outdir, "n": ranks,
                "procs": [subprocess.Popen([sys.executable, os.path.join(HERE, script)] + argv(r), env=env,
                                           stdout=logs[r], stderr=subprocess.STDOUT) for r in range(ranks)]}


def pytest_collection_finish(session):
    """The multi-process GPU tests need extra processes on the GPU.  They are started HERE -- after collection, so only
    when their test is really going to run (not on `-k` runs that deselect it), and before any test body of this process
    has touched the GPU (a process that has initialised the GPU must not exec another program on the GPU boxes); the
    tests themselves just wait for them."""
    wanted = {os.path.basename(str(i.fspath)) + "::" + i.name.split("[")[0] for i in session.items}
    try:
        import torch
        if torch.cuda.device_count() < 1:      # counting devices does not initialise the GPU
            return
    except Exception:
        return
    if "test_gpu_data_parallel.py::test_two_rank_fused_training_step_matches_mean_gradient_step" in wanted:
        _spawn("dp", "dp_worker.py", 2)
    if "test_gpu_data_parallel.py::test_two_rank_epoch_loop_shards_checkpoints_and_resumes" in wanted:
        _spawn("dptrain", "dp_train_worker.py", 2)
    if "test_gpu_rccl.py::test_rccl_single_rank_exchange_is_the_identity" in wanted:
        _spawn("rccl", "rccl_worker.py", 1)
    if "test_gpu_data_parallel.py::test_bench_spawns_its_own_ranks_when_started_without_a_launcher" in wanted:
        # plain `python bench.py --gpus 2 ...`: no torchrun, no RANK / WORLD_SIZE in the environment
        outdir = tempfile.mkdtemp(prefix="hig_benchspawn_")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        out, log = open(os.path.join(outdir, "stdout.txt"), "wb"), open(os.path.join(outdir, "rank0.log"), "wb")
        _DP["benchspawn"] = {"outdir": outdir, "n": 1, "procs": [subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"],
            env=env, stdout=out, stderr=log)]}


def pytest_sessionfinish(session, exitstatus):
    for grp in _DP.values():
        for p in grp["procs"]:
            if p.poll() is None:
                p.kill()


def _finished(tag, timeout):
    if tag not in _DP:
        pytest.skip("worker processes '%s' were not started (no GPU, or the test was not collected)" % tag)
    grp = _DP[tag]
    codes = [p.wait(timeout=timeout) for p in grp["procs"]]
    texts = [open(os.path.join(grp["outdir"], "rank%d.log" % r), errors="replace").read() for r in range(grp["n"])]
    return grp["outdir"], codes, texts


@pytest.fixture(scope="session")
def dp_workers():
    """(outdir, [returncode, ...], [log text, ...]) of the two data-parallel step workers, once they have finished."""
    return _finished("dp", 900)


@pytest.fixture(scope="session")
def dp_train_workers():
    return _finished("dptrain", 1200)


@pytest.fixture(scope="session")
def rccl_worker():
    return _finished("rccl", 900)


@pytest.fixture(scope="session")
def bench_spawn():
    return _finished("benchspawn", 1200)


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name))
    return load
