"""SelSampleEnumeration -- the reference's DEFAULT join enumerator (client_config.hpp:90,
polar_enumeration_algo.cpp:323-556) -- in the host mirror, against the join orders the reference itself enumerated
(tests/golden/ssb_skew_sample.json, made by tests/golden/make_golden_ssb_skew.py: the reference ran the SSB-skew
queries with `SET join_enumerator TO 'sample'`, its ALTERNATE matrix identifies every order of the bank)."""
import numpy as np
import pytest

import common
from polr_amd import host as phost
from polr_amd import ssb_skew


def _cases():
    gold = common.load_golden("ssb_skew_sample")
    return sorted(gold["cases"].keys())


@pytest.mark.parametrize("case", _cases())
def test_sample_orders_match_reference(case):
    gold = common.load_golden("ssb_skew_sample")
    c = gold["cases"][case]
    k = len(c["node_info"]) - 1
    got = phost.generate_join_orders("sample", 4, [0] * k, [[ssb_skew.PROBE_COLS.index(ssb_skew.DIM_KEY[n][1])]
                                                           for n in ssb_skew.QUERY_JOINS[c["query"]]],
                                     [1] * k, max_join_orders=c["max_join_orders"], node_info=c["node_info"])
    if c["paths"] is None:
        assert got is None
        return
    assert got is not None
    assert got[0].tolist() == c["paths"]


def test_sample_needs_plan_statistics():
    with pytest.raises(RuntimeError):
        phost.generate_join_orders("sample", 4, [0, 0], [[0], [1]], [1, 1], max_join_orders=3)


def test_sample_is_deterministic_and_caps_at_factorial():
    # one relation with a predicate, every build side unique: 1! = 1 distinct order -> SAMPLE finds no alternative;
    # Pipeline::Ready then falls back to BFS_MIN_CARD and pins the routing to DEFAULT_PATH (pipeline.cpp:216-225)
    info = [(1000, False, False), (100, True, True), (50, False, True), (10, False, True)]
    fb = phost.generate_join_orders("sample", 3, [0, 0, 0], [[0], [1], [2]], [30, 20, 10], node_info=info,
                                    return_routing=True)
    assert fb is not None and fb[3] == "default_path" and fb[0][0].tolist() == [0, 1, 2] and len(fb[0]) == 6
    info = [(1000000, False, False), (100, True, False), (5000, True, False), (10, False, False)]
    a = phost.generate_join_orders("sample", 3, [0, 0, 0], [[0], [1], [2]], [1, 1, 1], max_join_orders=8, node_info=info)
    b = phost.generate_join_orders("sample", 3, [0, 0, 0], [[0], [1], [2]], [1, 1, 1], max_join_orders=8, node_info=info)
    assert a is not None and np.array_equal(a[0], b[0])
    assert a[0][0].tolist() == [0, 1, 2] and len(a[0]) <= 6
    rest = [tuple(p) for p in a[0][1:].tolist()]
    assert rest == sorted(set(rest))  # distinct, lexicographic (std::set<vector<idx_t>>)


@pytest.mark.parametrize("case", ["q4.1-nested/3", "q4.1-nested/8"])
def test_sample_over_a_nested_build_side_matches_reference(case):
    """a build side that is itself a join tree (JoinOrderNode::nested_join_order, polar_enumeration_algo.cpp:262-280,
    :401-409): the reference ran Q4.1 with its customer dimension written as (customer JOIN nation WHERE n_region = ...)
    -- tests/golden/make_golden_nested.py; same pipeline and COUNT(*) as Q4.1, a different bank of join orders"""
    gold = common.load_golden("sample_nested")
    flat = common.load_golden("ssb_skew_sample")["cases"]["q4.1/3"]
    c = gold["cases"][case]
    assert c["count_star"] == flat["count_star"] and c["paths"] != flat["paths"]
    k = len(c["node_info"]) - 1
    info = [tuple(x) for x in c["node_info"]]
    assert any(len(x) > 3 and x[3] for x in info)
    got = phost.generate_join_orders("sample", 4, [0] * k, [[ssb_skew.PROBE_COLS.index(ssb_skew.DIM_KEY[n][1])]
                                                           for n in ssb_skew.QUERY_JOINS["q4.1"]],
                                     [1] * k, max_join_orders=c["max_join_orders"], node_info=info)
    assert got is not None and got[0].tolist() == c["paths"]
