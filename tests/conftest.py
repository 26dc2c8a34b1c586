import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """Committed fixture (data only; generated from the reference by tools/gen_golden.py)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_keys(name):
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        return [(k, tuple(s)) for k, s in json.load(f)[name]]


ERR_LOG = []      # (test id, max-norm relative error, rms-relative error) of every rel_err() call; worst ones printed at session end


def rel_errs(a, b):
    """(max |a-b| / max |b|, ||a-b||_2 / ||b||_2): the global-max "relative fp32" error every tolerance of this suite was first
    written in, and the rms-relative error next to it (the largest elements of a random-init trajectory decide the first one;
    the second weighs every element)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    d = a - b
    return ((d.abs().max() / b.abs().max().clamp_min(1e-30)).item(),
            (d.norm() / b.norm().clamp_min(1e-30)).item())


def rel_err(a, b):
    """The LARGER of the two norms of rel_errs(): ``assert rel_err(got, want) < tol`` therefore holds both the global-max and
    the rms-relative error to ``tol``.  Both figures are logged (ERR_LOG) and the worst are printed at session end."""
    mx, rms = rel_errs(a, b)
    ERR_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], mx, rms))
    return max(mx, rms)


def pytest_terminal_summary(terminalreporter):
    if not ERR_LOG:
        return
    worst = {}
    for tid, mx, rms in ERR_LOG:
        w = worst.get(tid)
        if w is None or max(mx, rms) > max(w):
            worst[tid] = (mx, rms)
    rows = sorted(worst.items(), key=lambda kv: -max(kv[1]))
    terminalreporter.write_line(f"parity error norms (worst rel_err() call per test, {len(rows)} tests): max-norm relative | rms relative")
    nshow = int(os.environ.get("DS_ERR_SUMMARY", "12"))
    for tid, (mx, rms) in rows[:nshow]:
        terminalreporter.write_line(f"  {mx:9.2e} | {rms:9.2e}  {tid}")
    out = os.environ.get("DS_ERR_LOG_FILE")
    if out:
        with open(out, "w") as f:
            for tid, (mx, rms) in rows:
                f.write(f"{mx:.3e}\t{rms:.3e}\t{tid}\n")


@pytest.fixture(scope="session")
def unet_sd():
    from diffusynth_amd.synth import synth_state_dict
    return synth_state_dict(golden_keys("unet_production"))


@pytest.fixture(scope="session")
def vqgan_sd():
    from diffusynth_amd.synth import synth_state_dict
    return synth_state_dict(golden_keys("vqgan_production"))


def gpu_available():
    return torch.cuda.is_available()
