"""Every environment switch the library still reads ("Switches") is flipped ONCE, in a child process (they are read
once per process), over a pass that touches every path they select between: the results must stay at the rounding level of the
path -- a switch chooses between two implementations of the same arithmetic, never between two results.  The diagnostic
switches whose builds are wrong by construction (HIG_BF16_DBG, HIG_BF16_WSP_DBG, HIG_F32_WSP_DBG: timing ablations; HIG_POISON) are not
flipped."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (switch, flipped value): the default of every switch is the other implementation (HIG_BWD_OVERLAP: unset = weight gradients of
# the fp32 backward on the second stream, of the bf16-storage backward on the caller's; 0 / 1 = neither / both)
KNOBS = [
    ("HIG_BWD_OVERLAP", "0"), ("HIG_BWD_OVERLAP", "1"), ("HIG_TEXT_FORK", "0"), ("HIG_FWD_SPLIT", "0"), ("HIG_LNFOLD32", "0"), ("HIG_GEMM_TAIL", "0"),
    ("HIG_GEMM_TILE", "128"), ("HIG_FEW_ROWS_SPLIT", "0"), ("HIG_FULLATTN_WAVES", "4"), ("HIG_FULLATTN_VALU", "1"),
    ("HIG_CTX16", "0"), ("HIG_FWD16_FORK", "0"), ("HIG_JOINT16", "0"), ("HIG_FUSE_APPLY", "0"), ("HIG_FUSE_OUT", "0"),
    ("HIG_EDGE16", "0"), ("HIG_BF16_TILE", "64"), ("HIG_BF16_FEWROW", "0"), ("HIG_BF16_WS", "0"), ("HIG_BF16_WSP", "0"),
    ("HIG_BF16_WS_NWJ", "4"), ("HIG_BF16_WS_ROWS", "100000"), ("HIG_LNFOLD", "0"), ("HIG_LNFOLD1024", "0"), ("HIG_CHIP_CUS", "128"), ("HIG_F32_WSP", "0"),
    ("HIG_TEXT_BATCH", "0"), ("HIG_APPLY_WAVE", "0"),
]


def run(env_extra):
    env = dict(os.environ, OMP_NUM_THREADS="4", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "knob_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return {"error": r.stdout[-2000:] + r.stderr[-2000:]}
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST ")][-1]
    return json.loads(line[len("DIGEST "):])


@pytest.fixture(scope="module")
def digests():
    """Every child once, eight at a time (a child is mostly interpreter start-up and parameter generation on the host; the GPU
    passes are milliseconds): {"": default digest, switch: digest with that switch flipped}."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = [("", {})] + [("%s=%s" % (k, v), {k: v}) for k, v in KNOBS]
    with ThreadPoolExecutor(max_workers=8) as pool:
        return dict(zip([j[0] for j in jobs], pool.map(lambda j: run(j[1]), jobs)))


@pytest.mark.parametrize("knob,value", KNOBS)
def test_every_switch_selects_between_implementations_of_the_same_arithmetic(knob, value, digests):
    default_digest, d = digests[""], digests["%s=%s" % (knob, value)]
    assert "error" not in default_digest, default_digest.get("error")
    assert "error" not in d, (knob, d.get("error"))
    assert set(d) == set(default_digest)
    for k, v in d.items():
        ref = default_digest[k]
        tol = 2e-2 if k.startswith("bf16") else 2e-4       # bf16 storage: another rounding sequence; fp32: another summation order
        assert abs(v - ref) <= tol * abs(ref), (knob, k, v, ref)


def test_fused_and_unfused_attention_front_pass_the_same_gradient_gates_against_the_oracle():
    """the bf16 training forward runs attention output + LayerNorm + modulation + SiLU as ONE kernel that normalises its
    fp32 accumulators while the y it saves is rounded to bf16, so the backward recomputes the LayerNorm statistics from slightly
    different values than the forward used (the unfused sequence, HIG_FUSE_APPLY=0, normalises exactly the bf16 y it saves).  Both
    forms must pass the SAME gates against the fp32 oracle's autograd: the default runs in the suite
    (tests/test_gpu_bf16_training.py), the unfused form here, in a child process (the switch is read once per process)."""
    env = dict(os.environ, OMP_NUM_THREADS="16", HIG_FUSE_APPLY="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_bf16_training.py"),
                        "-k", "every_gradient_against_oracle_autograd and width16"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
