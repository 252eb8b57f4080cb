"""diagnostic: the generic per-wave pipeline's own rate, without routing -- every join order of a JOB shape over the whole
(filtered) source as ONE host-given round through polr_probe_rounds (polr_path_kernel, plain launch); run under
`rocprofv3 --kernel-trace --stats` for the kernel times.
   python3 tools/micro/generic_rate.py job_q18 [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
from polr_amd import capi, workloads  # noqa: E402
from polr_amd import host as phost  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "job_q18"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = {"job_q18": workloads.job_q18, "job_light_01": workloads.job_light_01}[name](scale=1.0, seed=workloads.SEED)
ctx = capi.Context(0)
joins = capi.build_joins(ctx, wl, auto=True)
names = list(wl["probe"]["cols"].keys())
cols = list(wl["probe"]["cols"].values())
n_rows = len(cols[0])
gen = phost.generate_join_orders("each_last_once", len(names), [len(j["payload"]) for j in wl["joins"]],
                                 wl.get("cond_left_index") or [[j["key_src"][0][1]] for j in wl["joins"]],
                                 [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8, routing="adaptive_reinit")
paths = gen[0]
pipe = capi.Pipeline(ctx, cols, n_rows, joins, paths)
flt = wl["probe"].get("filter")
if flt:
    n_tuples, n_chunks = pipe.scan_filter([(names.index(c), op, const) for c, op, const in flt])
else:
    n_tuples = n_rows
print("%s: %d tuples, %d joins, paths %s" % (name, n_tuples, len(joins), paths.tolist()))
for p in range(len(paths)):
    for piece in (n_tuples, 65536, 4096):
        rounds = [(b, min(piece, n_tuples - b), p, 0) for b in range(0, n_tuples, piece)]
        best = None
        for _ in range(reps):
            ctx.sync()
            t0 = time.perf_counter()
            counts = pipe.probe_rounds(rounds)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("  path %d, %5d rounds of %8d: %.3f ms wall (launch + kernel + read-back), stage outputs %s" %
              (p, len(rounds), piece, best * 1e3, counts.sum(axis=0).tolist()))
