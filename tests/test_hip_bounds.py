"""Runs tools/bounds_sweep.py (the suite's shapes through libdiffusynth_hip_bounds.so, every global access of the
convolution / depthwise / attention / GroupNorm kernels checked against its operand's extent) in a child process —
the library is selected by DS_LIB before the first load, so it cannot share the test session's process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_global_access_is_inside_its_operand():
    lib = os.path.join(ROOT, "diffusynth_amd", "libdiffusynth_hip_bounds.so")
    assert os.path.exists(lib), "build it with __graft_entry__.build() / tools/build_variants.py bounds"
    env = dict(os.environ, DS_LIB="libdiffusynth_hip_bounds.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bounds_sweep.py")], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0 and "BOUNDS OK" in r.stdout
