"""diagnostic / cross-check: every join order of every JOB-shaped pipeline through the POOL launch (generic pipeline of
polr_gen_device.h, or the flat one) and through the per-round path kernel (polr_probe_device.h) -- two independent
implementations of RunPath; their per-(join order, position) tuple counts must be identical.
   python3 tools/check_job_paths.py [scale] [query ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
from polr_amd import capi, job_family as jf  # noqa: E402
from polr_amd import host as phost  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
only = sys.argv[2:]
shapes = jf.shapes()
tables = jf.Tables(scale=scale)
ctx = capi.Context(0)
bad = 0
for name in sorted(shapes):
    if only and name not in only:
        continue
    wl = jf.workload(name, tables, shapes[name])
    if wl is None:
        continue
    pn = list(wl["probe"]["cols"].keys())
    gen = phost.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                     wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8)
    if gen is None:
        continue
    paths = gen[0]
    joins = capi.build_joins(ctx, wl, auto=True)
    cols = list(wl["probe"]["cols"].values())
    pipe = capi.Pipeline(ctx, cols, len(cols[0]), joins, paths)
    flt = wl["probe"].get("filter")
    if flt:
        n, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt])
    else:
        n, n_chunks = len(cols[0]), (len(cols[0]) + 1023) // 1024
    k, P = len(wl["joins"]), len(paths)
    want = pipe.probe_rounds([(0, n, p, 0) for p in range(P)])
    mpx = capi.DeviceMultiplexer(pipe, "alternate", log_rounds=False)
    if flt:
        mpx.use_scan_chunks()
    capi.run_resident([mpx], [(0, n_chunks)], reset=True, finish=True)
    st = mpx.finish()
    got = np.asarray([[st["stage_out"][p][j] for j in range(k)] for p in range(P)], dtype=np.uint64)
    info = pipe.launch_info(False)
    ok = np.array_equal(got, want)
    print("%s %s: %d tuples, %d joins, %d orders, slots %d, flat %d, intermediates %d" %
          ("ok  " if ok else "DIFF", name, n, k, P, info["tuple_slots"], info["flat"], int(want.sum())), flush=True)
    if not ok:
        bad += 1
        for p in range(P):
            if not np.array_equal(got[p], want[p]):
                print("   order %s: pool %s  path kernel %s" % (paths[p].tolist(), got[p].tolist(), want[p].tolist()))
    mpx.close()
    pipe.close()
    for ht, _ in joins:
        ht.close()
print("%d pipelines differ" % bad)
sys.exit(1 if bad else 0)
