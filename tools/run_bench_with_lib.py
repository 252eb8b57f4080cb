#!/usr/bin/env python3
"""python tools/run_bench_with_lib.py <libpolr_hip variant .so> <bench.py args...> -- bench.py against another build of the
device library (an A/B aid, see tools/run_with_lib.py)"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
import polr_amd.capi as capi  # noqa: E402

capi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
