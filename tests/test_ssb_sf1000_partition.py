"""BASELINE configs[4] -- SSB-skew SF1000, broadcast build + partitioned probe over 8 GPUs -- as ONE rank sees it: the
lineorder partition of rank 7 of 8 (750 M rows at row offset 5.25 G of the 6 G-row table; rows a pure function of their
index, polr_amd.ssb_skew) generated on the device, the full SF1000 dimension tables (customer 30 M: a 3.75 MB bit table),
Q4.1 through the pool launch:

  * contiguous 8 M-row samples: COUNT(*) == the REFERENCE's own answer on exactly those rows
    (tests/golden/ssb_sf1000_samples.json, tests/golden/make_golden_sf1000.py);
  * the whole partition: every tuple is probed exactly once and COUNT(*) is the same number for ADAPTIVE_REINIT on 384
    executors, INIT_ONCE and each join order of the bank run statically;
  * a sample of the phase-one partition of rank 3 the same way (its 8 M rows generated on their own).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _count(stats, k, n_paths):
    return int(sum(sum(st["stage_out"][p][k - 1] for p in range(n_paths)) for st in stats))


def test_sf1000_partition_counts_agree_with_the_reference():
    """runs in a process of its own (torch generates the partition and has to initialise the GPU first)"""
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "sf1000 partition ok" in r.stdout


def main():
    import torch
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info()
    assert free > (60 << 30), "needs 60 GB of free HBM"
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import common
    from polr_amd import capi, ssb_skew
    from polr_amd import dist as pdist
    from polr_amd import host as phost
    GOLD = common.load_golden("ssb_sf1000_samples")
    ctx = capi.Context(0)
    q = GOLD["query"]
    z = ssb_skew.sizes(GOLD["scale"])
    wl = ssb_skew.workload(q, sf=GOLD["scale"], n_lo=z["n_lo"], host_probe=False)
    inst = wl["instance"]
    assert {k_: v for k_, v in inst.params().items() if k_ != "year_band_ends"} == GOLD["params"]
    names = list(ssb_skew.PROBE_COLS)
    k = len(wl["joins"])
    dim_rows = {"customer": len(inst.c_custkey), "supplier": inst.n_s, "part": inst.n_p, "date": 2556}
    node_info = [(z["n_lo"], False, False)] + [(dim_rows[j["name"]], j["name"] in ssb_skew.QUERY_WHERE[q], True)
                                              for j in wl["joins"]]
    paths = phost.generate_join_orders("sample", len(names), [0] * k, [[j["key_src"][0][1]] for j in wl["joins"]],
                                       [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=3,
                                       routing="adaptive_reinit", node_info=node_info)[0]
    P = len(paths)
    joins = capi.build_joins(ctx, wl, auto=True)  # (on a multi-GPU run: rank 0 builds, polr_bcast_build ships them)
    V = 1024

    def pipeline_over(lo, hi):
        cols_t = inst.lineorder_torch(lo, hi, dev, cols=names)
        cols = [capi.dev_col(cols_t[c].data_ptr(), cols_t[c].element_size(), signed=False) for c in names]
        return capi.Pipeline(ctx, cols, hi - lo, joins, paths), cols_t, cols

    def run(pipe, routing, n_exec, ranges):
        mpxs = [capi.DeviceMultiplexer(pipe, routing) for _ in range(n_exec)]
        capi.run_resident(mpxs, ranges, reset=True, finish=True)
        stats = capi.finish_many(mpxs)
        for m in mpxs:
            m.close()
        return _count(stats, k, P), int(sum(sum(st["input_tuple_count_per_path"]) for st in stats))

    # ---- rank 7 of 8: the whole partition on the device
    lo, hi = pdist.probe_partition(z["n_lo"], GOLD["world"], 7, V)
    assert [lo, hi] == GOLD["samples"][0]["partition"] and lo >= 5_250_000_000 - V
    pipe, cols_t, cols = pipeline_over(lo, hi)
    n = hi - lo
    n_chunks = (n + V - 1) // V
    rows = GOLD["sample_rows_each"]
    for s in GOLD["samples"]:
        if s["rank"] != 7:
            continue
        c0 = s["start_in_partition"] // V
        got, routed = run(pipe, "adaptive_reinit", 1, [(c0, c0 + rows // V)])
        assert (got, routed) == (s["count_star"], rows), (s, got, routed)
    E = 384
    even = [((e * n_chunks) // E, ((e + 1) * n_chunks) // E) for e in range(E)]
    whole, routed = run(pipe, "adaptive_reinit", E, even)
    assert routed == n
    assert run(pipe, "init_once", E, even) == (whole, n)
    for p in range(P):
        order = [p] + [i for i in range(P) if i != p]
        pp = capi.Pipeline(ctx, cols, n, joins, paths[order])
        assert run(pp, "default_path", 64, [((e * n_chunks) // 64, ((e + 1) * n_chunks) // 64) for e in range(64)]) == (whole, n), p
        pp.close()
    info = pipe.launch_info()
    pipe.close()
    del cols_t, cols
    torch.cuda.empty_cache()
    # ---- a sample of rank 3's partition (skew phase one), generated on its own
    for s in GOLD["samples"]:
        if s["rank"] != 3:
            continue
        p3, c3, _c = pipeline_over(s["start"], s["start"] + rows)
        got, routed = run(p3, "adaptive_reinit", 1, [(0, rows // V)])
        assert (got, routed) == (s["count_star"], rows), (s, got, routed)
        p3.close()
    for ht, _ in joins:
        ht.close()
    ctx.close()
    print("sf1000 partition ok: rank 7 of 8, %d rows, COUNT(*) %d, flat %d, LDS tables %d; samples %s" % (
        n, whole, info["flat"], info["lds_tables"], [s["count_star"] for s in GOLD["samples"]]))


if __name__ == "__main__":
    main()
