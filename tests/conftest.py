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


def rel_err(a, b):
    """max |a-b| / max |b| — the "relative fp32" error used for every tolerance in this suite."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


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
