#!/usr/bin/env python3
"""Diagnostic builds of libdiffusynth_hip (never the product library):

    python tools/build_variants.py bounds      -> diffusynth_amd/libdiffusynth_hip_bounds.so   (-DDS_BOUNDS=1)

Select one at run time with DS_LIB=libdiffusynth_hip_bounds.so (diffusynth_amd/_lib.py).  tests/test_hip_bounds.py
drives the suite's shapes through the bounds build and asserts ds_bounds_report() == 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

VARIANTS = {"bounds": ("libdiffusynth_hip_bounds.so", ["-DDS_BOUNDS=1"])}

if __name__ == "__main__":
    # named variants, or ad-hoc ones:  tag=-DFOO=1,-DBAR=2  ->  libdiffusynth_hip_<tag>.so
    from concurrent.futures import ThreadPoolExecutor
    jobs = []
    for name in sys.argv[1:] or ["bounds"]:
        if "=" in name:
            tag, fl = name.split("=", 1)
            jobs.append((f"libdiffusynth_hip_{tag}.so", fl.split(",")))
        else:
            jobs.append(VARIANTS[name])
    with ThreadPoolExecutor(max_workers=3) as ex:
        for lib in ex.map(lambda j: g.build_lib(os.path.join(ROOT, "diffusynth_amd", j[0]), flags=j[1]), jobs):
            print(lib)
