#!/usr/bin/env python3
"""python tools/run_with_lib.py <libpolr_hip variant .so> <pytest args...> -- the GPU tests against another build of the
device library (an A/B aid: e.g. a build with a fix compiled out, to see the test that pins the fix fail)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import polr_amd.capi as capi  # noqa: E402

capi.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:]))
