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
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_bounds()                  # no-op when the prebuilt library is up to date (it normally travels with the snapshot)
    assert os.path.exists(g.BOUNDS_LIB)
    env = dict(os.environ, DS_LIB="libdiffusynth_hip_bounds.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bounds_sweep.py")], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-4000:])
    print(r.stderr[-2000:])
    assert r.returncode == 0 and "BOUNDS OK" in r.stdout
