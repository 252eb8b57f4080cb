"""SSB-skew generator (duckdb-polr_amd/python/polr_amd/ssb_skew.py): the UPDATEs of benchmark/ssb-skew/init/load.sql:80-253
applied to synthetic base tables.  Checked here: numpy == torch bit for bit, partitions == slices of the whole table,
and the properties the UPDATE rules imply (which are what make the best join order change along the scan)."""
import numpy as np
import pytest

from polr_amd import ssb_skew


@pytest.fixture(scope="module")
def inst():
    return ssb_skew.Instance(n_lo=600_000, n_c=30_000, n_s=20_000, n_p=20_000)


def test_numpy_and_torch_generate_the_same_rows(inst):
    torch = pytest.importorskip("torch")
    cols = ("lo_orderkey", "lo_custkey", "lo_suppkey", "lo_partkey", "lo_orderdate", "lo_quantity", "lo_revenue")
    a = inst.lineorder(1000, 201_000, cols=cols)
    b = inst.lineorder_torch(1000, 201_000, torch.device("cpu"), cols=cols, block=50_000)
    for c in cols:
        bt = b[c].numpy()
        assert np.array_equal(a[c], bt.view(a[c].dtype)), c


def test_partitions_are_slices_of_the_table(inst):
    whole = inst.lineorder(0, inst.n_lo)
    part = inst.lineorder(123_456, 234_567)
    for c, v in part.items():
        assert np.array_equal(whole[c][123_456:234_567], v)


def test_update_rules_hold(inst):
    t = inst.lineorder(0, inst.n_lo, cols=("lo_orderkey", "lo_custkey", "lo_suppkey", "lo_partkey", "lo_orderdate",
                                            "lo_quantity"))
    okey = t["lo_orderkey"].astype(np.int64)
    assert okey.max() == inst.max_orderkey and np.all(np.diff(okey) >= 0)
    creg = inst.c_region[t["lo_custkey"].astype(np.int64) - 1]
    sreg = inst.s_region[t["lo_suppkey"].astype(np.int64) - 1]
    snat = inst.s_nation[t["lo_suppkey"].astype(np.int64) - 1]
    early = okey < inst.t400
    # load.sql:102-110: below the 400 M mark no order points at an ASIA customer any more
    assert not np.any(creg[early] == ssb_skew.R_ASIA)
    # :112-121: above it, AMERICA customers of every third order key were re-pointed to ASIA customers
    late3 = (~early) & (okey % 3 == 0)
    assert np.mean(creg[late3] == ssb_skew.R_ASIA) > 0.5
    # :146-154: below the mark no UNITED STATES supplier is left
    assert not np.any(snat[early] == ssb_skew.N_UNITED_STATES)
    # :156-165: above it the small-quantity lines of ASIA suppliers went to UNITED STATES suppliers
    lateq = (~early) & (t["lo_quantity"] <= 6)
    assert np.mean(snat[lateq] == ssb_skew.N_UNITED_STATES) > 0.8
    # :167-178: below the mark the large-quantity lines of ASIA suppliers went to UNITED KI1 / UNITED KI5 suppliers
    earlyq = early & (t["lo_quantity"] >= 43)
    if len(inst.s3):
        assert np.mean(np.isin(t["lo_suppkey"][earlyq], inst.s3)) > 0.8
    assert np.mean(sreg[early & (t["lo_quantity"] < 43)] == ssb_skew.R_ASIA) > 0.9
    # :198-245: the year is a function of the order-key band
    year = t["lo_orderdate"] // 10000
    ends = inst.bands
    assert np.all(year[okey <= ends[0]] == 1992)
    assert np.all(year[(okey > ends[0]) & (okey <= ends[1])] == 1993)
    assert np.all(year[(okey > ends[4]) & (okey <= inst.t401)] == 1997)
    # 1998 has 364 days in the date table (it ends on 1998-12-30): `lo_orderkey % 365 = 364` finds no row in d98 and
    # load.sql's scalar subquery yields NULL there; the generator writes 0, which equals no d_datekey either
    last = okey > inst.t401
    assert np.all((year[last] == 1998) | ((t["lo_orderdate"][last] == 0) & (okey[last] % 365 == 364)))
    assert set(np.unique(t["lo_orderdate"])) <= set(inst.d_datekey.tolist()) | {0}
    # keys stay inside the dimension tables
    assert t["lo_custkey"].min() >= 1 and t["lo_custkey"].max() <= inst.c_custkey[-1]
    assert t["lo_suppkey"].max() <= inst.n_s and t["lo_partkey"].max() <= inst.n_p


def test_dimension_updates(inst):
    base = np.arange(inst.n_c)
    ck = inst.c_custkey[base]
    # load.sql:81-82: every base customer with c_custkey % 10 <> 0 is AMERICA now (or was before)
    assert np.all(inst.c_region[base][ck % 10 != 0] == ssb_skew.R_AMERICA)
    assert np.all(inst.c_region[inst.n_c:] == ssb_skew.R_OCEANIA) and len(inst.c_custkey) == inst.n_c + 2500
    sk = inst.s_suppkey
    assert np.all(inst.s_region[sk % 10 != 0] == ssb_skew.R_ASIA)
    # :185-187
    assert np.all(inst.p_brand[inst.p_partkey % 3 == 0] == 2239)
    assert set(np.unique(inst.p_category[inst.p_partkey % 2 == 0])) <= {12, 14}
    p = inst.params()
    assert p["view_cardinalities"]["c1"] + p["view_cardinalities"]["c2"] == len(inst.c_custkey)


def test_sf100_view_cardinalities_match_load_sql_constants():
    """at SF100 sizes the views come out at the cardinalities load.sql hard-codes as moduli (within sampling noise):
    the synthetic base tables have the domains the constants were derived from"""
    inst = ssb_skew.Instance(n_lo=4, **{k: v for k, v in ssb_skew.sizes(100).items() if k != "n_lo"})
    got = inst.params()["view_cardinalities"]
    want = inst.params()["load_sql_moduli_at_sf100"]
    for k in ("c1", "c2", "s1"):
        assert abs(got[k] - want[k]) / want[k] < 0.02, (k, got[k], want[k])
    for k in ("c3", "s2"):
        assert abs(got[k] - want[k]) / want[k] < 0.15, (k, got[k], want[k])


def test_workload_shape():
    wl = ssb_skew.workload("q4.1", n_lo=100_000, n_c=30_000, n_s=20_000, n_p=20_000)
    assert [j["name"] for j in wl["joins"]] == ["customer", "supplier", "part", "date"]
    assert set(wl["probe"]["cols"]) == set(ssb_skew.PROBE_COLS)
    assert all(len(c) == 100_000 for c in wl["probe"]["cols"].values())
    wl3 = ssb_skew.workload("q3.1", n_lo=10_000, n_c=30_000, n_s=20_000, n_p=20_000)
    assert len(wl3["joins"]) == 3


def test_row_salt_gives_another_table_of_the_same_shape():
    """weak scaling: rank r probes rows with row_salt = r * n_lo -- order keys, dates (a function of the order key) and
    the skew rules stay, the per-row draws change; salt 0 is the table itself"""
    inst = ssb_skew.workload("q4.1", sf=0.05, host_probe=False)["instance"]
    n = min(inst.n_lo, 200_000)
    cols = ("lo_orderkey", "lo_custkey", "lo_suppkey", "lo_partkey", "lo_orderdate")
    a = inst.lineorder(0, n, cols=cols)
    b = inst.lineorder(0, n, cols=cols, row_salt=0)
    c = inst.lineorder(0, n, cols=cols, row_salt=4 * ((inst.n_lo + 3) // 4))
    for k in cols:
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(a["lo_orderkey"], c["lo_orderkey"]) and np.array_equal(a["lo_orderdate"], c["lo_orderdate"])
    assert (a["lo_partkey"] != c["lo_partkey"]).mean() > 0.9 and (a["lo_suppkey"] != c["lo_suppkey"]).mean() > 0.5
    # the same four lines of an order share their customer in both tables
    assert np.array_equal(c["lo_custkey"][0::4][: n // 4], c["lo_custkey"][1::4][: n // 4])
