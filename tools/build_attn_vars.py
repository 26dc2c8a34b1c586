#!/usr/bin/env python3
"""Build the attention experiment variants: python tools/build_attn_vars.py 0 1 2 4 6  ->  build/libattn_var_<n>.so"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

for n in sys.argv[1:]:
    g.build_lib(os.path.join(ROOT, "build", "libattn_var_%s.so" % n), flags=["-DDS_ATTN_VAR=%s" % n])
