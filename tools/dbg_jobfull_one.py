#!/usr/bin/env python3
"""Runs ONE pipeline of the JOB-shaped family (polr_amd/job_family.py) through the pool launch, repeatedly -- to chase a
run the watchdog gave up:  python tools/dbg_jobfull_one.py 06c --scale 0.2 --executors 4 --repeat 20 [--share-after N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from polr_amd import capi, job_family as jf, host as phost  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("query")
    ap.add_argument("--scale", type=float, default=0.2)
    ap.add_argument("--executors", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--share-after", type=int, default=0)
    ap.add_argument("--watchdog-us", type=int, default=300000)
    ap.add_argument("--routing", default="adaptive_reinit")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ctx = capi.Context(0)
    kn = {"watchdog_us": args.watchdog_us}
    if args.share_after:
        kn["share_after"] = args.share_after
    ctx.set_pool_tuning(**kn)
    shapes = jf.shapes()
    tables = jf.Tables(scale=args.scale)
    wl = jf.workload(args.query, tables, shapes[args.query])
    pn = list(wl["probe"]["cols"].keys())
    paths = phost.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                       wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=8,
                                       routing=args.routing)[0]
    joins = capi.build_joins(ctx, wl, auto=True)
    tens = [torch.from_numpy(np.ascontiguousarray(wl["probe"]["cols"][c])).to(dev) for c in pn]
    n_rows = len(wl["probe"]["cols"][pn[0]])
    cols = [capi.dev_col(t.data_ptr(), t.element_size(), signed=True) for t in tens]
    pipe = capi.Pipeline(ctx, cols, n_rows, joins, paths)
    flt = wl["probe"].get("filter")
    V = 1024
    if flt:
        n_tuples, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt], vector_size=V)
    else:
        n_tuples, n_chunks = n_rows, (n_rows + V - 1) // V
    E = max(1, min(args.executors, n_chunks))
    mpxs = []
    for e in range(E):
        m = capi.DeviceMultiplexer(pipe, args.routing, chunk_size=V, log_rounds=False)
        if flt:
            m.use_scan_chunks()
        mpxs.append(m)
    ranges = [((e * n_chunks) // E, ((e + 1) * n_chunks) // E) for e in range(E)]
    print("%s: %d joins %s, %d tuples, %d chunks, E=%d, paths %s" % (args.query, len(wl["joins"]), [j["name"] for j in wl["joins"]],
                                                                    n_tuples, n_chunks, E, np.asarray(paths).tolist()), flush=True)
    want = None
    bad = 0
    for r in range(args.repeat):
        t0 = time.time()
        capi.run_resident(mpxs, ranges, reset=True, finish=True)
        try:
            sts = capi.finish_many(mpxs)
        except capi.PolrError as e:
            bad += 1
            print("run %d: %s (%.3f s)" % (r, e, time.time() - t0), flush=True)
            continue
        tot = sum(st["num_intermediates"] for st in sts)
        k = len(wl["joins"])
        cnt = sum(st["stage_out"][p][k - 1] for st in sts for p in range(len(paths)))
        if want is None:
            want = (cnt,)
        print("run %d: intermediates %d count %d %s (%.1f ms)" % (r, tot, cnt, "" if (cnt,) == want else "COUNT DIFFERS",
                                                              (time.time() - t0) * 1e3), flush=True)
    print("given up: %d of %d" % (bad, args.repeat))


if __name__ == "__main__":
    main()
