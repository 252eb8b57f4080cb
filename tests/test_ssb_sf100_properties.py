"""BASELINE configs[2] at FULL size (SSB-skew Q4.1, SF100: 600 M lineorder rows generated on the device) through
size-independent properties -- the oracle cannot run 600 M rows in seconds, so:

  * two contiguous 8 M-row samples (one per skew phase): COUNT(*) == the reference's own answer on exactly those rows
    (tests/golden/ssb_sf100_samples.json);
  * the whole table: COUNT(*) is the same number for ADAPTIVE_REINIT on 256 executors, INIT_ONCE, every join order of
    the bank run statically, a morsel-driven run and BACKPRESSURE -- whatever the routers decide, a tuple is probed
    exactly once (tuples routed == rows of the table);
  * additivity: COUNT(*) of the two halves of the table == COUNT(*) of the whole.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _count(stats, k, n_paths):
    return int(sum(sum(st["stage_out"][p][k - 1] for p in range(n_paths)) for st in stats))


def test_sf100_counts_agree_across_routing_and_with_the_reference():
    """runs in a process of its own: the table is generated with torch on the device, and torch has to initialise the
    GPU before the HIP library of this repository does (the test session's context has long done that)"""
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "sf100 properties ok" in r.stdout


def main():
    import torch
    dev = torch.device("cuda", 0)
    free, _total = torch.cuda.mem_get_info()
    assert free > (40 << 30), "needs 40 GB of free HBM"
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import common
    from polr_amd import capi, ssb_skew
    from polr_amd import host as phost
    GOLD = common.load_golden("ssb_sf100_samples")
    gpu_ctx = capi.Context(0)
    q = GOLD["query"]
    z = ssb_skew.sizes(GOLD["scale"])
    wl = ssb_skew.workload(q, sf=GOLD["scale"], n_lo=z["n_lo"], host_probe=False)
    inst = wl["instance"]
    names = list(ssb_skew.PROBE_COLS)
    cols_t = inst.lineorder_torch(0, z["n_lo"], dev, cols=names)
    n = z["n_lo"]
    k = len(wl["joins"])
    dim_rows = {"customer": len(inst.c_custkey), "supplier": inst.n_s, "part": inst.n_p, "date": 2556}
    node_info = [(n, False, False)] + [(dim_rows[j["name"]], j["name"] in ssb_skew.QUERY_WHERE[q], True)
                                      for j in wl["joins"]]
    gen = phost.generate_join_orders("sample", len(names), [0] * k, [[j["key_src"][0][1]] for j in wl["joins"]],
                                     [len(j["keys"][0]) for j in wl["joins"]], max_join_orders=3,
                                     routing="adaptive_reinit", node_info=node_info, return_routing=True)
    paths = gen[0]
    assert paths.tolist() == GOLD["join_orders"]  # the bank the reference's SAMPLE enumerator builds for this plan
    P = len(paths)
    joins = capi.build_joins(gpu_ctx, wl, auto=True)
    cols = [capi.dev_col(cols_t[c].data_ptr(), cols_t[c].element_size(), signed=False) for c in names]
    pipe = capi.Pipeline(gpu_ctx, cols, n, joins, paths)
    assert pipe.launch_info()["flat"] == 1
    n_chunks = (n + 1023) // 1024

    def run(routing, n_exec, ranges=None, morsels=0):
        mpxs = [capi.DeviceMultiplexer(pipe, routing) for _ in range(n_exec)]
        if morsels:
            capi.run_resident_morsels(mpxs, 0, n_chunks, morsels, reset=True, finish=True)
        else:
            rg = ranges or [((e * n_chunks) // n_exec, ((e + 1) * n_chunks) // n_exec) for e in range(n_exec)]
            capi.run_resident(mpxs, rg, reset=True, finish=True)
        stats = capi.finish_many(mpxs)
        for m in mpxs:
            m.close()
        return _count(stats, k, P), int(sum(sum(st["input_tuple_count_per_path"]) for st in stats))

    # the reference's answers on the two samples
    V = 1024
    for s0, want in zip(GOLD["sample_starts"], GOLD["sample_count_star"]):
        got, routed = run("adaptive_reinit", 1, ranges=[(s0 // V, (s0 + GOLD["sample_rows_each"]) // V)])
        assert got == want and routed == GOLD["sample_rows_each"]
    # the whole table
    whole, routed = run("adaptive_reinit", 256)
    assert routed == n and whole == GOLD["whole_table_count_star_device"]
    assert run("init_once", 256) == (whole, n)
    assert run("adaptive_reinit", 256, morsels=1024) == (whole, n)
    for p in range(P):
        order = [p] + [i for i in range(P) if i != p]
        pp = capi.Pipeline(gpu_ctx, cols, n, joins, paths[order])
        mp = [capi.DeviceMultiplexer(pp, "default_path") for _ in range(64)]
        capi.run_resident(mp, [((e * n_chunks) // 64, ((e + 1) * n_chunks) // 64) for e in range(64)], reset=True, finish=True)
        st = capi.finish_many(mp)
        assert _count(st, k, P) == whole, "join order %d" % p
        for m in mp:
            m.close()
        pp.close()
    bp = [capi.DeviceMultiplexer(pipe, "backpressure") for _ in range(P)]
    capi.run_backpressure(bp, 0, n_chunks, 2048)
    st = capi.finish_many(bp)
    assert _count(st, k, P) == whole and sum(sum(s["input_tuple_count_per_path"]) for s in st) == n
    for m in bp:
        m.close()
    # additivity over a partition of the table (what the multi-GPU run relies on)
    half = (n_chunks // 2)
    a, ra = run("adaptive_reinit", 128, ranges=[((e * half) // 128, ((e + 1) * half) // 128) for e in range(128)])
    b, rb = run("adaptive_reinit", 128, ranges=[(half + (e * (n_chunks - half)) // 128,
                                                  half + ((e + 1) * (n_chunks - half)) // 128) for e in range(128)])
    assert a + b == whole and ra + rb == n
    pipe.close()
    for ht, _ in joins:
        ht.close()
    gpu_ctx.close()
    print("sf100 properties ok: COUNT(*) %d, samples %s" % (whole, GOLD["sample_count_star"]))


if __name__ == "__main__":
    main()
