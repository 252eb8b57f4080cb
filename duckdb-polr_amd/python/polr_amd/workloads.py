"""Seeded synthetic workloads shaped like the reference's benchmarks.

The reference loads IMDB/SSB data over the network (benchmark/imdb/init/load.sql:1-21,
benchmark/ssb/init/load.sql:74-78); neither is available offline, so every configuration runs on
generated tables with the same schemas, cardinalities and key skew.  All generators are pure
functions of (name, scale, seed) so that tests, the golden-vector maker and bench.py see the same
bytes.  seed 1337 = the reference's default ClientConfig::seed (client_config.hpp:81).

A workload is a dict:
  probe:   {"name", "cols": {col: ndarray}, "key_cols": [...]}
  joins:   list (original join order = path 0) of
           {"name", "keys": [ndarray], "payload": {col: ndarray}, "key_src": [(-1|join, col index)],
            "perfect": (min, max) or None, "sql_type": ...}
  sql:     how the same query reads for the reference (used only by the golden maker / cpu baseline)
"""
import numpy as np

SEED = 1337


def _rng(seed, salt):
    return np.random.default_rng(np.random.SeedSequence([seed, salt]))


def zipf_keys(rng, n, n_keys, s=1.0):
    """n draws from a bounded Zipf(s) over [0, n_keys) (rank 0 most frequent), then the ranks are
    scattered over the key space with a fixed permutation so hot keys are not the small ids."""
    ranks = np.arange(1, n_keys + 1, dtype=np.float64)
    w = ranks ** (-s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    u = rng.random(n)
    idx = np.searchsorted(cdf, u, side="left").astype(np.int64)
    perm = rng.permutation(n_keys)
    return perm[idx]


# -------------------------------------------------------------------------------------------------
# star join with phase skew: the selective dimension changes along the scan order, which is what
# the SSB-skew transforms do (benchmark/ssb-skew/init/load.sql:80-253, Appendix D of SURVEY.md)
# -------------------------------------------------------------------------------------------------
def star_skew(n_fact=200_000, seed=SEED, n_phases=4, dims=((60_000, 37, 0), (2_000, 1, 0), (20_000, 5, 3)),
              with_nulls=False):
    """dims: (n_keys, key_stride, max_dup).  key_stride > 1 spreads the key range past the
    perfect-hash limit when n_keys*stride > 1e6; max_dup > 0 repeats build keys 1..max_dup times."""
    rng = _rng(seed, 1)
    joins = []
    fact = {"id": np.arange(n_fact, dtype=np.int32)}
    phase = (np.arange(n_fact) * n_phases // n_fact).astype(np.int64)
    for d, (n_keys, stride, max_dup) in enumerate(dims):
        keys = (np.arange(n_keys, dtype=np.int64) * stride + 11).astype(np.int32)
        if max_dup:
            reps = rng.integers(1, max_dup + 1, size=n_keys)
            bkeys = np.repeat(keys, reps)
            bkeys = bkeys[rng.permutation(len(bkeys))]
        else:
            bkeys = keys[rng.permutation(n_keys)]
        # a dimension filter keeps a fraction of the build keys; the fraction of FACT rows that hit a
        # kept key depends on the phase: dimension d is very selective in phase d, loose elsewhere
        keep = rng.random(n_keys) < 0.5
        kept_keys = keys[keep]
        dropped_keys = keys[~keep]
        hit_prob = np.where(phase % len(dims) == d, 0.05, 0.9)
        hit = rng.random(n_fact) < hit_prob
        fk = np.where(hit, kept_keys[rng.integers(0, len(kept_keys), n_fact)],
                      dropped_keys[rng.integers(0, len(dropped_keys), n_fact)]).astype(np.int32)
        fact["k%d" % d] = fk
        sel = np.isin(bkeys, kept_keys)
        bkeys = bkeys[sel]
        payload = {"p%d" % d: (bkeys.astype(np.int64) * 7 + d).astype(np.int32),
                   "q%d" % d: (bkeys % 251).astype(np.int16)}
        kmin, kmax = int(keys.min()), int(keys.max())
        perfect = (kmin, kmax) if (kmax - kmin) <= 1_000_000 and not max_dup else None
        joins.append({"name": "d%d" % d, "keys": [bkeys.astype(np.int32)], "key_names": ["k"],
                      "payload": payload, "key_src": [(-1, 1 + d)], "perfect": perfect})
    wl = {"name": "star_skew", "probe": {"name": "fact", "cols": fact}, "joins": joins}
    if with_nulls:
        wl["name"] = "star_skew_nulls"
        pv = {}
        for d in range(len(dims)):
            v = (rng.random(n_fact) > 0.03).astype(np.uint8)
            pv["k%d" % d] = v
        wl["probe"]["valid"] = pv
        for d, j in enumerate(joins):
            n = len(j["keys"][0])
            j["key_valid"] = [(rng.random(n) > 0.02).astype(np.uint8)]
            j["payload_valid"] = {"p%d" % d: (rng.random(n) > 0.1).astype(np.uint8)}
            j["perfect"] = None if j["perfect"] is None else j["perfect"]
    return wl


# -------------------------------------------------------------------------------------------------
# star with non-equality conditions next to the equalities (JoinCondition::comparison other than EQUAL: evaluated by
# RowOperations::Match on every candidate pair, join_hashtable.cpp:455-464).  "preds" = [(op, (src_join, src_col),
# payload column name)]: left OP right, the right side a column of the join's build side
# -------------------------------------------------------------------------------------------------
def star_pred(n_fact=120_000, seed=SEED):
    wl = star_skew(n_fact=n_fact, seed=seed)
    rng = _rng(seed, 7)
    wl["name"] = "star_pred"
    fact = wl["probe"]["cols"]
    fact["v"] = rng.integers(-60, 300, n_fact).astype(np.int16)
    fact["u"] = rng.integers(0, 15_000, n_fact).astype(np.int32)
    names = list(fact.keys())
    j0, j1, j2 = wl["joins"]
    j0["preds"] = [("<", (-1, names.index("v")), "q0")]                                        # fact.v < d0.q0
    j1["preds"] = [(">=", (-1, names.index("u")), "p1")]                                       # fact.u >= d1.p1
    j2["preds"] = [("<>", (-1, names.index("v")), "q2"), ("<=", (-1, names.index("u")), "p2")]  # two on one join
    for j in wl["joins"]:
        j["perfect"] = None  # a join with more than one condition is never a perfect-hash join
    return wl


# -------------------------------------------------------------------------------------------------
# dependent chain: join 1 probes with a column that join 0's build side provides
# (left_expression_bindings, polar_config.cpp:152-229); join 2 is independent
# -------------------------------------------------------------------------------------------------
def chain_dep(n_fact=120_000, seed=SEED):
    rng = _rng(seed, 2)
    n_a, n_b, n_c = 30_000, 8_000, 5_000
    a_keys = (np.arange(n_a, dtype=np.int64) * 41 + 5).astype(np.int32)
    b_keys = (np.arange(n_b, dtype=np.int64) * 3 + 1).astype(np.int32)
    c_keys = (np.arange(n_c, dtype=np.int32) + 100).astype(np.int32)
    a_keep = rng.random(n_a) < 0.6
    a_bref = b_keys[rng.integers(0, n_b, n_a)]          # a.bref -> b.k (FK carried by the build side)
    a_bref = np.where(rng.random(n_a) < 0.7, a_bref, a_bref + 1).astype(np.int32)  # some miss b
    fact = {
        "id": np.arange(n_fact, dtype=np.int32),
        "ak": a_keys[zipf_keys(rng, n_fact, n_a, 0.8)].astype(np.int32),
        "ck": np.where(rng.random(n_fact) < 0.4, c_keys[rng.integers(0, n_c, n_fact)], 7).astype(np.int32),
    }
    perm = rng.permutation(int(a_keep.sum()))
    ja = {"name": "a", "keys": [a_keys[a_keep][perm]], "key_names": ["k"],
          "payload": {"bref": a_bref[a_keep][perm], "av": (a_keys[a_keep][perm] % 1000).astype(np.int32)},
          "key_src": [(-1, 1)], "perfect": None}
    jb = {"name": "b", "keys": [b_keys[rng.permutation(n_b)]], "key_names": ["k"],
          "payload": {"bv": None}, "key_src": [(0, 0)], "perfect": (int(b_keys.min()), int(b_keys.max()))}
    jb["payload"]["bv"] = (jb["keys"][0].astype(np.int64) * 13 % 9973).astype(np.int32)
    cdup = np.repeat(c_keys, rng.integers(1, 4, size=n_c))
    cdup = cdup[rng.permutation(len(cdup))]
    jc = {"name": "c", "keys": [cdup.astype(np.int32)], "key_names": ["k"],
          "payload": {"cv": (np.arange(len(cdup)) % 97).astype(np.int16)}, "key_src": [(-1, 2)], "perfect": None}
    return {"name": "chain_dep", "probe": {"name": "fact", "cols": fact}, "joins": [ja, jb, jc]}


# -------------------------------------------------------------------------------------------------
# key semantics beyond "same type, NULL never matches": a CAST'ed key (INTEGER probe column against a BIGINT build key: the
# binder wraps the left side in CAST, which POLARConfig accepts, polar_config.cpp:75-82) and IS NOT DISTINCT FROM
# (JoinHashTable::null_values_are_equal, join_hashtable.cpp:35-36), next to a plain join
# -------------------------------------------------------------------------------------------------
KEY_BY_VALUE, KEY_NULL_EQUAL = 1, 2  # (polr_hip.h: POLR_KEY_*)


def key_semantics(n_fact=24_000, seed=SEED, cast=True):
    """cast=False: without dim_a -- the reference never multiplexes a pipeline one of whose keys is CAST'ed (it tests
    ExpressionType::CAST where a bound cast is OPERATOR_CAST, polar_config.cpp:78), so only this form has a reference run
    under the multiplexer"""
    rng = _rng(seed, 7)
    n_a, n_b, n_c, n_d = 3_000, 400, 900, 64
    a_keys = (np.arange(n_a, dtype=np.int64) * 5 - 2_000)            # BIGINT, negative values included
    a_keys[::97] += 1 << 33                                          # ... and some no INTEGER can hold
    b_keys = (np.arange(n_b, dtype=np.int32) * 3 + 1).astype(np.int32)
    b_keys = np.concatenate([b_keys, b_keys[:60]])                   # repeated keys
    b_valid = np.ones(len(b_keys), dtype=np.uint8)
    b_valid[rng.choice(len(b_keys), 25, replace=False)] = 0          # 25 build rows whose key is NULL
    c_keys = (np.arange(n_c, dtype=np.int32) + 50).astype(np.int32)
    d_keys = (np.arange(n_d, dtype=np.int32) * 2).astype(np.int32)
    fa = (rng.integers(0, n_a + 400, n_fact) * 5 - 2_000).astype(np.int32)  # (the tail misses dimension a)
    fa_valid = (rng.random(n_fact) < 0.97).astype(np.uint8)
    fb = np.where(rng.random(n_fact) < 0.8, b_keys[rng.integers(0, n_b, n_fact)], 2).astype(np.int32)
    fb_valid = (rng.random(n_fact) < 0.9).astype(np.uint8)           # 10 % NULL: they meet the 25 NULL build rows
    fc = np.where(rng.random(n_fact) < 0.7, c_keys[rng.integers(0, n_c, n_fact)], 7).astype(np.int32)
    phase = (np.arange(n_fact) * 4 // n_fact)                        # d is selective in the second half of the table
    fd = np.where((phase < 2) | (rng.random(n_fact) < 0.2), d_keys[rng.integers(0, n_d, n_fact)], 1).astype(np.int32)
    fact = {"id": np.arange(n_fact, dtype=np.int32), "a": fa, "b": fb, "c": fc, "d": fd}
    perm_b = rng.permutation(len(b_keys))
    ja = {"name": "dim_a", "keys": [a_keys[rng.permutation(n_a)]], "key_names": ["k"],
          "payload": {"pa": None}, "key_src": [(-1, 1)], "perfect": None, "key_flags": [KEY_BY_VALUE]}
    ja["payload"]["pa"] = (ja["keys"][0] % 1000).astype(np.int32)
    jb = {"name": "dim_b", "keys": [b_keys[perm_b]], "key_names": ["k"], "key_valid": [b_valid[perm_b]],
          "payload": {"pb": (np.arange(len(b_keys)) % 13).astype(np.int32)}, "key_src": [(-1, 2)], "perfect": None,
          "key_flags": [KEY_NULL_EQUAL]}
    jc = {"name": "dim_c", "keys": [c_keys[rng.permutation(n_c)]], "key_names": ["k"],
          "payload": {"pc": None}, "key_src": [(-1, 3)], "perfect": (int(c_keys.min()), int(c_keys.max()))}
    jc["payload"]["pc"] = (jc["keys"][0] * 7 % 101).astype(np.int32)
    jd = {"name": "dim_d", "keys": [d_keys[rng.permutation(n_d)]], "key_names": ["k"],
          "payload": {"pd": None}, "key_src": [(-1, 4)], "perfect": (int(d_keys.min()), int(d_keys.max()))}
    jd["payload"]["pd"] = (jd["keys"][0] + 5).astype(np.int32)
    joins = ([ja] if cast else []) + [jb, jc, jd]
    wl = {"name": "key_semantics", "probe": {"name": "fact", "cols": fact, "valid": {"a": fa_valid, "b": fb_valid}},
          "joins": joins, "cond_left_index": ([[1]] if cast else []) + [[2], [3], [4]]}
    # the same for the reference: NULLs travel as a sentinel no key uses and are set by UPDATEs after the load
    NULL_A, NULL_B = -2_000_000_000, -2_000_000_001
    t_fact = {"id": fact["id"], "a": np.where(fa_valid != 0, fa, NULL_A).astype(np.int32),
              "b": np.where(fb_valid != 0, fb, NULL_B).astype(np.int32), "c": fc, "d": fd}
    t_b = {"k": np.where(jb["key_valid"][0] != 0, jb["keys"][0], NULL_B).astype(np.int32), "pb": jb["payload"]["pb"]}
    wl["ref"] = {
        "tables": {"fact": t_fact, "dim_a": {"k": ja["keys"][0], "pa": ja["payload"]["pa"]}, "dim_b": t_b,
                   "dim_c": {"k": jc["keys"][0], "pc": jc["payload"]["pc"]},
                   "dim_d": {"k": jd["keys"][0], "pd": jd["payload"]["pd"]}},
        "pk": {},
        "settings": ["UPDATE fact SET a = NULL WHERE a = %d" % NULL_A, "UPDATE fact SET b = NULL WHERE b = %d" % NULL_B,
                     "UPDATE dim_b SET k = NULL WHERE k = %d" % NULL_B,
                     "SET disabled_optimizers TO 'join_order,statistics_propagation'"],
        "query": "SELECT COUNT(*) FROM fact " + ("JOIN dim_a ON fact.a = dim_a.k " if cast else "") +
                 "JOIN dim_b ON fact.b IS NOT DISTINCT FROM dim_b.k JOIN dim_c ON fact.c = dim_c.k JOIN dim_d ON fact.d = dim_d.k"}
    return wl


# -------------------------------------------------------------------------------------------------
# VARCHAR join keys: the key column of such a join is the 64-bit hash the engine computes for the strings anyway
# (JoinHashTable::Hash), the strings ride along as a verifying condition (POLR_CMP_STR_EQ) -- hash for the bucket, comparison
# for the decision, as RowOperations::Match does.  fact.s = dim_s.k (strings of 3..40 characters, NULLs on both sides,
# repeated build keys), dim_s.t = dim_t.k (a VARCHAR build column of an earlier join as the key), fact.c = dim_c.k.
# hash_bits < 64 truncates the hashes: collisions on purpose.  "codes": the same joins on dictionary codes (what the
# oracle, which has no strings, joins on -- the same equalities).
# -------------------------------------------------------------------------------------------------
def varchar_keys(n_fact=20_000, seed=SEED, hash_bits=12):
    from . import capi as _capi
    rng = _rng(seed, 8)

    def word(i, salt):
        # (short and long strings: inline cells and heap cells; common prefixes among the long ones)
        base = "k%d" % i if i % 3 else "a-rather-long-key-with-a-common-prefix-%06d" % i
        return (base + ("/" + salt if salt else "")).encode()

    n_s, n_t, n_c = 1_500, 300, 700
    t_vals = [word(i, "t") for i in range(n_t)]
    s_vals = [word(i, "") for i in range(n_s)]
    s_rows = list(range(n_s)) + [int(x) for x in rng.integers(0, n_s, 200)]      # 200 repeated build keys
    s_rows = [s_rows[i] for i in rng.permutation(len(s_rows))]
    dim_s_k = [s_vals[i] for i in s_rows]
    dim_s_kvalid = (rng.random(len(s_rows)) < 0.97).astype(np.uint8)
    dim_s_t = [t_vals[int(x)] if x < n_t else word(int(x), "miss") for x in rng.integers(0, n_t + 60, len(s_rows))]
    dim_t_k = [t_vals[i] for i in rng.permutation(n_t)]
    c_keys = (np.arange(n_c, dtype=np.int32) * 2 + 10).astype(np.int32)
    fs = [s_vals[int(x)] if x < n_s else word(int(x), "none") for x in rng.integers(0, n_s + 300, n_fact)]
    fs_valid = (rng.random(n_fact) < 0.95).astype(np.uint8)
    fc = np.where(rng.random(n_fact) < 0.75, c_keys[rng.integers(0, n_c, n_fact)], 3).astype(np.int32)
    h = lambda vals: _capi.string_hashes(vals, hash_bits)  # noqa: E731
    fact = {"id": np.arange(n_fact, dtype=np.int32), "c": fc, "hs": h(fs)}
    js = {"name": "dim_s", "keys": [h(dim_s_k)], "key_names": ["hk"], "key_valid": [dim_s_kvalid],
          "payload": {"ps": (np.arange(len(s_rows)) % 17).astype(np.int32), "ht": h(dim_s_t)},
          "strings": {"k": dim_s_k, "t": dim_s_t}, "key_src": [(-1, 2)], "perfect": None,
          # (left side: the probe table's VARCHAR column "s" = probe column 3, behind the three fixed-width ones)
          "preds": [("str_eq", (-1, 3), "k")]}
    jt = {"name": "dim_t", "keys": [h(dim_t_k)], "key_names": ["hk"],
          "payload": {"pt": (np.arange(n_t) * 3 % 11).astype(np.int32)}, "strings": {"k": dim_t_k},
          "key_src": [(0, 1)], "perfect": None,
          # (left side: dim_s's VARCHAR payload column "t": its payload columns are ps, ht, then the strings k, t)
          "preds": [("str_eq", (0, 3), "k")]}
    jc = {"name": "dim_c", "keys": [c_keys[rng.permutation(n_c)]], "key_names": ["k"], "payload": {"pc": None},
          "key_src": [(-1, 1)], "perfect": (int(c_keys.min()), int(c_keys.max()))}
    jc["payload"]["pc"] = (jc["keys"][0] % 97).astype(np.int32)
    wl = {"name": "varchar_keys", "probe": {"name": "fact", "cols": fact, "strings": {"s": fs},
                                            "valid": {"hs": fs_valid}, "string_valid": {"s": fs_valid}},
          "joins": [js, jt, jc],
          # BoundReference index of every join's probe-side key in the reference's layout: fact(id, c, s), then dim_s's
          # columns (k, ps, t), dim_t's ...
          "cond_left_index": [[2], [5], [1]]}
    # ---- the same joins on dictionary codes (oracle)
    words = sorted(set(fs) | set(dim_s_k) | set(dim_s_t) | set(dim_t_k))
    code = {w: i for i, w in enumerate(words)}
    enc = lambda vals: np.array([code[v] for v in vals], dtype=np.int32)  # noqa: E731
    wl["codes"] = {
        "probe": {"name": "fact", "cols": {"id": fact["id"], "c": fc, "s": enc(fs)}, "valid": {"s": fs_valid}},
        "joins": [{"name": "dim_s", "keys": [enc(dim_s_k)], "key_valid": [dim_s_kvalid],
                   "payload": {"ps": js["payload"]["ps"], "t": enc(dim_s_t)}, "key_src": [(-1, 2)], "perfect": None},
                  {"name": "dim_t", "keys": [enc(dim_t_k)], "payload": {"pt": jt["payload"]["pt"]}, "key_src": [(0, 1)],
                   "perfect": None},
                  {"name": "dim_c", "keys": jc["keys"], "payload": {"pc": jc["payload"]["pc"]}, "key_src": [(-1, 1)],
                   "perfect": jc["perfect"]}]}
    # ---- and for the reference: VARCHAR columns as they are, NULLs as a sentinel set by UPDATEs after the load
    null = b"\x01NULL"
    t_fact = {"id": fact["id"], "c": fc, "s": [v if ok else null for v, ok in zip(fs, fs_valid)]}
    t_s = {"k": [v if ok else null for v, ok in zip(dim_s_k, dim_s_kvalid)], "ps": js["payload"]["ps"], "t": dim_s_t}
    wl["ref"] = {
        "tables": {"fact": t_fact, "dim_s": t_s, "dim_t": {"k": dim_t_k, "pt": jt["payload"]["pt"]},
                   "dim_c": {"k": jc["keys"][0], "pc": jc["payload"]["pc"]}},
        "pk": {},
        "settings": ["UPDATE fact SET s = NULL WHERE s = '\x01NULL'", "UPDATE dim_s SET k = NULL WHERE k = '\x01NULL'",
                     "SET disabled_optimizers TO 'join_order,statistics_propagation'"],
        "query": "SELECT COUNT(*) FROM fact JOIN dim_s ON fact.s = dim_s.k JOIN dim_t ON dim_s.t = dim_t.k "
                 "JOIN dim_c ON fact.c = dim_c.k"}
    return wl


# -------------------------------------------------------------------------------------------------
# heavy fan-out (JOB's cast_info / movie_info style duplicates)
# -------------------------------------------------------------------------------------------------
def fanout(n_fact=40_000, seed=SEED):
    rng = _rng(seed, 3)
    n_m = 6_000
    m_keys = (np.arange(n_m, dtype=np.int64) * 211 + 3).astype(np.int32)
    reps1 = np.minimum(rng.zipf(1.6, size=n_m), 300)
    b1 = np.repeat(m_keys, reps1)
    b1 = b1[rng.permutation(len(b1))]
    reps2 = rng.integers(0, 5, size=n_m)
    b2 = np.repeat(m_keys, reps2)
    b2 = b2[rng.permutation(len(b2))]
    fact = {"id": np.arange(n_fact, dtype=np.int32),
            "mk": np.where(rng.random(n_fact) < 0.7, m_keys[zipf_keys(rng, n_fact, n_m, 1.0)], 1).astype(np.int32)}
    j1 = {"name": "ci", "keys": [b1.astype(np.int32)], "key_names": ["movie_id"],
          "payload": {"role": (np.arange(len(b1)) % 11).astype(np.int32)}, "key_src": [(-1, 1)], "perfect": None}
    j2 = {"name": "mk", "keys": [b2.astype(np.int32)], "key_names": ["movie_id"],
          "payload": {"kw": (np.arange(len(b2)) * 3 % 1009).astype(np.int32)}, "key_src": [(-1, 1)], "perfect": None}
    return {"name": "fanout", "probe": {"name": "fact", "cols": fact}, "joins": [j1, j2]}


# -------------------------------------------------------------------------------------------------
# JOB-light 01 shape (BASELINE.json configs[1]):
#   SELECT COUNT(*) FROM movie_companies mc, title t, movie_info_idx mi_idx
#   WHERE t.id=mc.movie_id AND t.id=mi_idx.movie_id AND mi_idx.info_type_id=112 AND mc.company_type_id=2
# (benchmark/job-light/queries/01.sql).  Probe side = movie_companies (2 609 129 rows) filtered on
# company_type_id; builds = title (2 528 312 rows, dense ids -> key range > 1e6 -> chained table)
# and movie_info_idx filtered on info_type_id (1 380 035 rows before the filter).
# -------------------------------------------------------------------------------------------------
JOB_CARD = {"title": 2_528_312, "movie_companies": 2_609_129, "movie_info_idx": 1_380_035,
            "cast_info": 36_244_344, "movie_info": 14_835_720, "movie_keyword": 4_523_930, "name": 4_167_491}


def job_light_01(scale=1.0, seed=SEED):
    rng = _rng(seed, 4)
    n_t = max(int(JOB_CARD["title"] * scale), 1000)
    n_mc = max(int(JOB_CARD["movie_companies"] * scale), 1000)
    n_mi = max(int(JOB_CARD["movie_info_idx"] * scale), 1000)
    t_id = np.arange(1, n_t + 1, dtype=np.int32)
    t_id = t_id[rng.permutation(n_t)]
    mc_movie = (zipf_keys(rng, n_mc, n_t, 1.0) + 1).astype(np.int32)
    mc_ctype = rng.integers(1, 3, n_mc).astype(np.int32)
    # movie_info_idx: up to 3 info rows per movie for about a fifth of the movies, types 99..113
    mi_movie = (zipf_keys(rng, n_mi, n_t, 0.6) + 1).astype(np.int32)
    mi_type = rng.integers(99, 114, n_mi).astype(np.int32)
    mc_sel = np.nonzero(mc_ctype == 2)[0].astype(np.uint32)
    mi_keep = mi_type == 112
    probe = {"name": "movie_companies", "cols": {"movie_id": mc_movie, "company_type_id": mc_ctype},
             "filter_sel": mc_sel, "filter_sql": "company_type_id=2",
             "filter": [("company_type_id", "=", 2)]}
    jt = {"name": "title", "keys": [t_id], "key_names": ["id"], "payload": {}, "key_src": [(-1, 0)], "perfect": None}
    jmi = {"name": "movie_info_idx", "keys": [mi_movie[mi_keep]], "key_names": ["movie_id"], "payload": {},
           "key_src": [(-1, 0)], "perfect": None,
           "unfiltered": {"movie_id": mi_movie, "info_type_id": mi_type}, "filter_sql": "info_type_id=112"}
    # how the same pipeline reads for the reference: explicit left-deep join order (path 0), both
    # joins probed with mc.movie_id (t.id = mc.movie_id = mi_idx.movie_id is one equivalence class)
    ref = {"tables": {"movie_companies": {"movie_id": mc_movie, "company_type_id": mc_ctype},
                      "title": {"id": t_id},
                      "movie_info_idx": {"movie_id": mi_movie, "info_type_id": mi_type}},
           "settings": ["SET disabled_optimizers TO 'join_order'"],
           "query": "SELECT COUNT(*) FROM movie_companies mc JOIN title t ON mc.movie_id = t.id "
                    "JOIN movie_info_idx mi_idx ON mc.movie_id = mi_idx.movie_id "
                    "WHERE mi_idx.info_type_id = 112 AND mc.company_type_id = 2"}
    return {"name": "job_light_01", "probe": probe, "joins": [jt, jmi], "ref": ref}


def job_q18(scale=1.0, seed=SEED):
    """JOB 18a shape (7 tables; the query BASELINE.json's >= 10x target is quoted on) at IMDB cardinalities:

        cast_info ci (probe, filtered on note)  JOIN name n ON ci.person_id = n.id            (n filtered: ~0.5 %)
                                                JOIN title t ON ci.movie_id = t.id
                                                JOIN movie_info mi ON ci.movie_id = mi.movie_id   (fan-out ~6)
                                                JOIN info_type it1 ON mi.info_type_id = it1.id    (DEPENDENT on mi; 1 row)
                                                JOIN movie_info_idx mi_idx ON ci.movie_id = mi_idx.movie_id
                                                JOIN info_type it2 ON mi_idx.info_type_id = it2.id (DEPENDENT on mi_idx)

    six multiplexed joins, two of them keyed by a build column of an earlier join; COUNT(*) sink.  Strings (ci.note IN
    (...), n.name LIKE, it.info = ...) are integer codes here: the filters are evaluated where the reference pushes
    them (table scans), the joins see the same cardinalities."""
    rng = _rng(seed, 18)
    n_t = max(int(JOB_CARD["title"] * scale), 1000)
    n_ci = max(int(JOB_CARD["cast_info"] * scale), 5000)
    n_n = max(int(JOB_CARD["name"] * scale), 1000)
    n_mi = max(int(JOB_CARD["movie_info"] * scale), 3000)
    n_mx = max(int(JOB_CARD["movie_info_idx"] * scale), 1000)
    ci = {"person_id": (zipf_keys(rng, n_ci, n_n, 0.8) + 1).astype(np.int32),
          "movie_id": (zipf_keys(rng, n_ci, n_t, 0.9) + 1).astype(np.int32),
          "note_id": rng.integers(0, 20, n_ci).astype(np.int32)}
    n_id = np.arange(1, n_n + 1, dtype=np.int32)
    n_flag = (rng.random(n_n) < 0.005).astype(np.int32)  # gender = 'm' AND name LIKE '%Tim%'
    t_id = np.arange(1, n_t + 1, dtype=np.int32)[rng.permutation(n_t)]
    mi_movie = (zipf_keys(rng, n_mi, n_t, 0.6) + 1).astype(np.int32)
    mi_type = rng.integers(1, 21, n_mi).astype(np.int32)
    mx_movie = (zipf_keys(rng, n_mx, n_t, 0.6) + 1).astype(np.int32)
    mx_type = rng.integers(99, 114, n_mx).astype(np.int32)
    it_id = np.arange(1, 114, dtype=np.int32)
    it_code = it_id.copy()  # info_type.info as a code: 'budget' = 5, 'votes' = 112
    jn = {"name": "name", "keys": [n_id[n_flag == 1]], "key_names": ["id"], "payload": {}, "key_src": [(-1, 0)],
          "perfect": None}
    jt = {"name": "title", "keys": [t_id], "key_names": ["id"], "payload": {}, "key_src": [(-1, 1)], "perfect": None}
    jmi = {"name": "movie_info", "keys": [mi_movie], "key_names": ["movie_id"], "payload": {"info_type_id": mi_type},
           "key_src": [(-1, 1)], "perfect": None}
    jit1 = {"name": "info_type1", "keys": [it_id[it_code == 5]], "key_names": ["id"], "payload": {},
            "key_src": [(2, 0)], "perfect": None}
    jmx = {"name": "movie_info_idx", "keys": [mx_movie], "key_names": ["movie_id"], "payload": {"info_type_id": mx_type},
           "key_src": [(-1, 1)], "perfect": None}
    jit2 = {"name": "info_type2", "keys": [it_id[it_code == 112]], "key_names": ["id"], "payload": {},
            "key_src": [(4, 0)], "perfect": None}
    ref = {"tables": {"cast_info": ci, "name": {"id": n_id, "flag": n_flag}, "title": {"id": t_id},
                      "movie_info": {"movie_id": mi_movie, "info_type_id": mi_type},
                      "movie_info_idx": {"movie_id": mx_movie, "info_type_id": mx_type},
                      "info_type": {"id": it_id, "info": it_code}},
           "settings": ["SET disabled_optimizers TO 'join_order'"],
           "query": "SELECT COUNT(*) FROM cast_info ci JOIN name n ON ci.person_id = n.id "
                    "JOIN title t ON ci.movie_id = t.id JOIN movie_info mi ON ci.movie_id = mi.movie_id "
                    "JOIN info_type it1 ON mi.info_type_id = it1.id "
                    "JOIN movie_info_idx mi_idx ON ci.movie_id = mi_idx.movie_id "
                    "JOIN info_type it2 ON mi_idx.info_type_id = it2.id "
                    "WHERE ci.note_id <= 1 AND n.flag = 1 AND it1.info = 5 AND it2.info = 112"}
    return {"name": "job_q18",
            "probe": {"name": "cast_info", "cols": ci, "filter": [("note_id", "<=", 1)],
                      "filter_sel": np.nonzero(ci["note_id"] <= 1)[0].astype(np.uint32)},
            "joins": [jn, jt, jmi, jit1, jmx, jit2], "ref": ref,
            # the BoundReference index of every join's probe-side condition in the original column layout
            # (3 probe columns, then each join's build columns): what POLARConfig::GenerateJoinOrders works on
            "cond_left_index": [[0], [1], [1], [3], [1], [4]]}


# -------------------------------------------------------------------------------------------------
# SSB-skew Q4.1 shape (BASELINE.json configs[2]): lineorder x {customer, supplier, part, date}
# with the selective dimension changing along lo_orderkey (benchmark/ssb-skew/init/load.sql:80-253,
# queries/q4-1.sql).  scale=1.0 is SF100 (600 M lineorder rows); the dimension cardinalities follow
# the SSB specification (customer 30k*SF, supplier 2k*SF, part 200k*(1+log2 SF), date 2556).
# -------------------------------------------------------------------------------------------------
def ssb_skew_q41(sf=1.0, seed=SEED, n_lineorder=None):
    rng = _rng(seed, 5)
    n_lo = int(6_000_000 * sf) if n_lineorder is None else int(n_lineorder)
    n_c = max(int(30_000 * sf), 3000)
    n_s = max(int(2_000 * sf), 200)
    n_p = max(int(200_000 * (1 + np.log2(max(sf, 1)))), 20_000)
    n_d = 2556
    regions = 5
    c_region = rng.integers(0, regions, n_c)
    # skew: 90% of non-AMERICA(=1) customers rewritten to AMERICA (load.sql:81-82)
    c_region = np.where((c_region != 1) & (rng.random(n_c) < 0.9), 1, c_region)
    s_region = rng.integers(0, regions, n_s)
    s_region = np.where(rng.random(n_s) < 0.9, 2, s_region)  # 90% -> ASIA (load.sql:139-140)
    p_mfgr = rng.integers(1, 6, n_p)
    d_keys = (19920101 + np.arange(n_d)).astype(np.uint32)
    d_year = (1992 + np.arange(n_d) * 7 // n_d).astype(np.uint16)
    pos = np.arange(n_lo)
    band = pos * 7 // n_lo  # lo_orderdate is a function of the orderkey band (load.sql:190-245)
    lo_date = d_keys[(band * (n_d // 7) + rng.integers(0, n_d // 7, n_lo)).clip(0, n_d - 1)]
    lo_cust = rng.integers(1, n_c + 1, n_lo).astype(np.uint32)
    lo_supp = rng.integers(1, n_s + 1, n_lo).astype(np.uint32)
    lo_part = rng.integers(1, n_p + 1, n_lo).astype(np.uint32)
    # re-point keys by scan position (load.sql:102-132,146-178): first 2/3 of the scan avoids the
    # AMERICA customers, the last third avoids the AMERICA suppliers
    am_c = np.nonzero(c_region == 1)[0].astype(np.uint32) + 1
    non_am_c = np.nonzero(c_region != 1)[0].astype(np.uint32) + 1
    am_s = np.nonzero(s_region == 1)[0].astype(np.uint32) + 1
    non_am_s = np.nonzero(s_region != 1)[0].astype(np.uint32) + 1
    early = pos < (2 * n_lo) // 3
    if len(non_am_c) and len(am_c):
        lo_cust = np.where(early & (rng.random(n_lo) < 0.85), non_am_c[rng.integers(0, len(non_am_c), n_lo)], lo_cust)
        lo_cust = np.where(~early, am_c[rng.integers(0, len(am_c), n_lo)], lo_cust)
    if len(non_am_s) and len(am_s):
        lo_supp = np.where(early, am_s[rng.integers(0, len(am_s), n_lo)], lo_supp)
        lo_supp = np.where(~early & (rng.random(n_lo) < 0.9), non_am_s[rng.integers(0, len(non_am_s), n_lo)], lo_supp)
    lineorder = {"lo_custkey": lo_cust.astype(np.uint32), "lo_suppkey": lo_supp.astype(np.uint32),
                 "lo_partkey": lo_part.astype(np.uint32), "lo_orderdate": lo_date.astype(np.uint32),
                 "lo_revenue": rng.integers(100, 10_000, n_lo).astype(np.uint32),
                 "lo_supplycost": rng.integers(50, 5_000, n_lo).astype(np.uint32)}
    c_keep = c_region == 1
    s_keep = s_region == 1
    p_keep = (p_mfgr == 1) | (p_mfgr == 2)
    c_keys = (np.nonzero(c_keep)[0] + 1).astype(np.uint32)
    s_keys = (np.nonzero(s_keep)[0] + 1).astype(np.uint32)
    p_keys = (np.nonzero(p_keep)[0] + 1).astype(np.uint32)

    def perfect(n):
        return (1, n) if n - 1 <= 1_000_000 else None

    jc = {"name": "customer", "keys": [c_keys], "key_names": ["c_custkey"],
          "payload": {"c_nation": (c_keys % 25).astype(np.uint16)}, "key_src": [(-1, 0)], "perfect": perfect(n_c)}
    js = {"name": "supplier", "keys": [s_keys], "key_names": ["s_suppkey"], "payload": {}, "key_src": [(-1, 1)],
          "perfect": perfect(n_s)}
    jp = {"name": "part", "keys": [p_keys], "key_names": ["p_partkey"], "payload": {}, "key_src": [(-1, 2)],
          "perfect": perfect(n_p)}
    jd = {"name": "date", "keys": [d_keys], "key_names": ["d_datekey"], "payload": {"d_year": d_year},
          "key_src": [(-1, 3)], "perfect": (int(d_keys.min()), int(d_keys.max()))}
    return {"name": "ssb_skew_q41", "probe": {"name": "lineorder", "cols": lineorder}, "joins": [jc, js, jp, jd]}


def ssb_q11(sf=0.2, seed=SEED):
    """BASELINE.json configs[0]: SSB Q1.1 shape -- lineorder x date, filters on both sides (SURVEY 8(d) cfg 1:
    orderdate uniform over the 2 556 date keys, discount U[0,10], quantity U[1,50], extendedprice; d_year = 1993,
    lo_discount BETWEEN 1 AND 3, lo_quantity < 25).  One join: the plain (non-multiplexed) path."""
    rng = _rng(seed, 11)
    n_lo = int(6_000_000 * sf)
    n_d = 2556
    d_keys = (19920101 + np.arange(n_d)).astype(np.uint32)
    d_year = (1992 + np.arange(n_d) * 7 // n_d).astype(np.uint16)
    lineorder = {"lo_orderdate": d_keys[rng.integers(0, n_d, n_lo)],
                 "lo_discount": rng.integers(0, 11, n_lo).astype(np.uint16),
                 "lo_quantity": rng.integers(1, 51, n_lo).astype(np.uint16),
                 "lo_extendedprice": rng.integers(90_000, 10_000_000, n_lo).astype(np.uint32)}
    keep = d_year == 1993
    jd = {"name": "date", "keys": [d_keys[keep]], "key_names": ["d_datekey"], "payload": {"d_year": d_year[keep]},
          "key_src": [(-1, 0)], "perfect": None}
    return {"name": "ssb_q11", "probe": {"name": "lineorder", "cols": lineorder,
                                         "filter": [("lo_discount", ">=", 1), ("lo_discount", "<=", 3),
                                                    ("lo_quantity", "<", 25)]},
            "joins": [jd], "date_full": {"d_datekey": d_keys, "d_year": d_year},
            "sql_where": "d_year = 1993 AND lo_discount >= 1 AND lo_discount <= 3 AND lo_quantity < 25"}


def default_paths(k, kind="each_last_once"):
    """Join orders a deterministic enumerator of the reference yields when no join depends on
    another (EachLastOnceEnumeration / EachFirstOnceEnumeration, polar_enumeration_algo.cpp:610-667)."""
    base = list(range(k))
    paths = [base]
    if kind == "each_last_once":
        for i in range(k - 1):
            paths.append([j for j in base if j != i] + [i])
    elif kind == "each_first_once":
        for i in range(1, k):
            paths.append([i] + [j for j in base if j != i])
    return np.asarray(paths, dtype=np.int32)
