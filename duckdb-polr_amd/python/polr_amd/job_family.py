"""BASELINE.json configs[3]: "Full JOB (imdb) 113 queries, independent POLR pipelines sharded across 8 GPUs".

The reference runs the 113 queries of the Join Order Benchmark over the IMDB snapshot (benchmark/imdb/*.benchmark,
SQL in benchmark/imdb_plan_cost/queries/*.sql); neither the data (fetched over the network by
benchmark/imdb/init/load.sql:1-21) nor the optimizer that turns each query into pipelines is available / in scope here.
What this module builds is the JOB-SHAPED FAMILY: for every one of the 113 queries ONE multiplexed pipeline that keeps
the query's join graph --

  * job_shapes.json (made from the SQL files by tools/make_job_shapes.py: tables, equi-join predicates, number and kind of
    filter predicates per table; no SQL text) gives the graph;
  * the probe side is the query's largest table (by IMDB cardinality); every other table that an equality chain
    connects to it becomes a hash join, keyed by a probe column where the equivalence class of the join column holds
    one, else by a build column of the join that leads to it (the dependent joins of
    POLARConfig::GenerateJoinOrders, src/parallel/polar_config.cpp:72-95), breadth first, at most 8 joins
    (POLR_MAX_JOINS; the reference multiplexes runs of consecutive joins inside one pipeline, a 17-table query is
    several pipelines there);
  * tables are synthetic at the IMDB cardinalities: dense ids, foreign keys power-law distributed over the parent's
    ids, filters thinned to a selectivity that depends only on the kinds of predicates the query puts on the table.

Every pipeline is independent (own tables, own multiplexers): whole queries are dealt round-robin to the GPUs
(polr_amd.dist.shard_queries), nothing is exchanged.
"""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

CARD = {"title": 2_528_312, "cast_info": 36_244_344, "movie_info": 14_835_720, "movie_keyword": 4_523_930,
        "movie_companies": 2_609_129, "name": 4_167_491, "char_name": 3_140_339, "aka_name": 901_343,
        "aka_title": 361_472, "person_info": 2_963_664, "movie_info_idx": 1_380_035, "movie_link": 29_997,
        "complete_cast": 135_086, "keyword": 134_170, "company_name": 234_997, "info_type": 113, "kind_type": 7,
        "company_type": 4, "role_type": 12, "link_type": 18, "comp_cast_type": 4}
# foreign keys of the IMDB schema: (table, column) -> referenced table (its dense id)
FK = {("cast_info", "movie_id"): "title", ("cast_info", "person_id"): "name", ("cast_info", "person_role_id"): "char_name",
      ("cast_info", "role_id"): "role_type", ("movie_info", "movie_id"): "title",
      ("movie_info", "info_type_id"): "info_type", ("movie_info_idx", "movie_id"): "title",
      ("movie_info_idx", "info_type_id"): "info_type", ("movie_keyword", "movie_id"): "title",
      ("movie_keyword", "keyword_id"): "keyword", ("movie_companies", "movie_id"): "title",
      ("movie_companies", "company_id"): "company_name", ("movie_companies", "company_type_id"): "company_type",
      ("title", "kind_id"): "kind_type", ("aka_name", "person_id"): "name", ("aka_title", "movie_id"): "title",
      ("person_info", "person_id"): "name", ("person_info", "info_type_id"): "info_type",
      ("movie_link", "movie_id"): "title", ("movie_link", "linked_movie_id"): "title",
      ("movie_link", "link_type_id"): "link_type", ("complete_cast", "movie_id"): "title",
      ("complete_cast", "subject_id"): "comp_cast_type", ("complete_cast", "status_id"): "comp_cast_type"}
SMALL = 200  # tables of up to this many rows are enumerations: an equality keeps one of their rows
SEL = {"eq": 0.1, "like": 0.05, "notlike": 0.9, "ne": 0.95, "range": 0.3, "between": 0.2, "null": 0.1, "notnull": 0.9,
       "or": 0.15, "notin": 0.9}
MAX_JOINS = 8
SKEW = 1.4  # exponent of the foreign-key power law (1 = uniform)
_PRIMES = [2654435761, 2246822519, 3266489917, 668265263, 374761393, 2870177467, 1540483507, 2971215073]


def shapes():
    return json.load(open(os.path.join(_HERE, "job_shapes.json")))["queries"]


def _mix32(x, salt):
    """cheap deterministic hash -> [0, 1): which rows a filter keeps"""
    z = (x.astype(np.uint64) + np.uint64(salt)) * np.uint64(0x9E3779B97F4A7C15)
    z ^= z >> np.uint64(31)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(29)
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def _salt(text):
    """process-independent salt of a name (Python's str hash is randomised per process)"""
    v = 7
    for ch in text:
        v = (v * 131 + ord(ch)) % 1_000_003
    return v + 3


def selectivity(table, n_rows, kinds):
    s = 1.0
    for kd in kinds:
        if kd.startswith("in"):
            n = int(kd[2:])
            s *= min(1.0, n / n_rows) if n_rows <= SMALL else min(0.05 * n, 0.5)
        elif kd == "eq" and n_rows <= SMALL:
            s *= 1.0 / n_rows
        else:
            s *= SEL[kd]
    return max(s, 1.0 / max(n_rows, 1))


class Tables:
    """the synthetic IMDB instance of one scale, columns made on demand and kept (every query reuses them)"""

    def __init__(self, scale=1.0, seed=1337):
        self.scale, self.seed = float(scale), int(seed)
        self._cols = {}

    def rows(self, table):
        n = CARD[table]
        return n if n <= SMALL else max(int(n * self.scale), 1000)

    def column(self, table, col):
        key = (table, col)
        if key in self._cols:
            return self._cols[key]
        n = self.rows(table)
        if col == "id":
            v = np.arange(1, n + 1, dtype=np.int32)
        elif key in FK:
            parent = self.rows(FK[key])
            rng = np.random.default_rng(np.random.SeedSequence([self.seed] + [ord(c) for c in table + "." + col]))
            idx = (parent * rng.random(n) ** SKEW).astype(np.int64)        # power law: few parents get most rows
            # scattered over the id space by a multiplier of the column's own (a bijection: the primes exceed every
            # table size), so that the hot parents of two referencing tables are different rows -- a join of two
            # foreign-key columns then fans out by about rows / parents, as in IMDB, not by hot x hot
            mult = _PRIMES[_salt(table + "." + col) % len(_PRIMES)]
            v = (1 + (idx * mult) % parent).astype(np.int32)
        else:
            # a join column that is not a declared foreign key (the queries equate e.g. two movie_id columns through
            # title): same construction over the table's own id space
            rng = np.random.default_rng(np.random.SeedSequence([self.seed] + [ord(c) for c in table + "." + col]))
            v = (1 + (n * rng.random(n) ** 2).astype(np.int64) % n).astype(np.int32)
        self._cols[key] = v
        return v

    def strings(self, table, col, rows=None):
        """a VARCHAR attribute column (title.title, name.name, movie_info.info ...): one string per row, a pure function
        of (table, column, row): lengths 3 .. 27 characters, so that both forms of a string_t cell occur (up to 12
        characters inline, longer ones through the heap); many rows share their first characters, so comparisons have
        to look past the 4-byte prefix.  rows: the row ids wanted (default: all)"""
        n = self.rows(table)
        ids = np.arange(n, dtype=np.int64) if rows is None else np.asarray(rows, dtype=np.int64)
        h = (_mix32(ids, _salt(table + "." + col)) * (1 << 24)).astype(np.int64)
        stem = (table[:2] + col[:2]).encode()
        words = [b"a", b"the", b"of", b"night", b"return", b"story", b"day", b"last"]
        out = []
        for i, x in zip(ids.tolist(), h.tolist()):
            v = stem + b"-" + words[x % 8] + b"-" + str(x % 9973).encode()
            if x & 0x100:
                v += b"-" + words[(x >> 9) % 8] + b"-" + str(i).encode()
            out.append(v)
        return out


def _find(parent, x):
    while parent[x] != x:
        parent[x] = parent[parent[x]]
        x = parent[x]
    return x


def pipeline_shape(q):
    """the multiplexed pipeline of one query shape: probe alias, its columns, and the ordered joins
    [(alias, build key column, key source ('probe', column) | (join index, column))]"""
    tables, joins = q["tables"], q["joins"]
    cols = sorted({(a, c) for j in joins for a, c in (tuple(j[0]), tuple(j[1]))})
    parent = {x: x for x in cols}
    for a, b in joins:
        ra, rb = _find(parent, tuple(a)), _find(parent, tuple(b))
        if ra != rb:
            parent[ra] = rb
    classes = {}
    for x in cols:
        classes.setdefault(_find(parent, x), []).append(x)
    probe = max(sorted(tables), key=lambda a: CARD[tables[a]])
    placed = {probe: -1}
    order = []
    frontier = True
    while frontier and len(order) < MAX_JOINS:
        frontier = False
        cands = []
        for cl in classes.values():
            inside = [x for x in cl if x[0] in placed]
            outside = [x for x in cl if x[0] not in placed]
            if not inside or not outside:
                continue
            # key source: a probe column of the class if there is one, else the earliest placed join's column
            src = min(inside, key=lambda x: (placed[x[0]], x[1]))
            for b in outside:
                cands.append((0 if src[0] == probe else 1, placed[src[0]], b[0], b[1], src))
        seen = set()
        for pri, _p, alias, bcol, src in sorted(cands):
            if alias in seen or alias in placed or len(order) >= MAX_JOINS:
                continue
            seen.add(alias)
            order.append((alias, bcol, src))
            placed[alias] = len(order) - 1
            frontier = True
    return probe, order


def workload(name, tables, q=None):
    """workload dict (the format of polr_amd.workloads) of query `name` over the synthetic instance `tables`"""
    q = q or shapes()[name]
    probe, order = pipeline_shape(q)
    if len(order) < 2:
        return None  # POLAR needs two consecutive joins (polar_config.cpp:44)
    ptable = q["tables"][probe]
    # probe columns: every probe column some join is keyed by, plus the filter column
    pcols = []
    for alias, bcol, src in order:
        if src[0] == probe and src[1] not in pcols:
            pcols.append(src[1])
    probe_cols = {c: tables.column(ptable, c) for c in pcols}
    n_probe = tables.rows(ptable)
    psel = selectivity(ptable, n_probe, q["filters"].get(probe, []))
    flt = None
    if psel < 1.0:
        f = (_mix32(np.arange(n_probe, dtype=np.int64), 17) * 1000).astype(np.int32)
        probe_cols["f"] = f
        cut = max(int(round(psel * 1000)), 1)
        flt = [("f", "<", cut)]
    names = list(probe_cols.keys())
    # payload columns of every join: the columns dependents are keyed by
    payload_cols = {a: [] for a, _b, _s in order}
    for alias, bcol, src in order:
        if src[0] != probe and src[1] not in payload_cols[src[0]]:
            payload_cols[src[0]].append(src[1])
    index_of = {a: i for i, (a, _b, _s) in enumerate(order)}
    joins = []
    for alias, bcol, src in order:
        t = q["tables"][alias]
        n = tables.rows(t)
        sel = selectivity(t, n, q["filters"].get(alias, []))
        keep = np.ones(n, dtype=bool) if sel >= 1.0 else (_mix32(np.arange(n, dtype=np.int64), _salt(alias)) < sel)
        if not keep.any():
            keep[0] = True
        key = tables.column(t, bcol)[keep]
        payload = {c: tables.column(t, c)[keep] for c in payload_cols[alias]}
        if src[0] == probe:
            key_src = [(-1, names.index(src[1]))]
        else:
            key_src = [(index_of[src[0]], payload_cols[src[0]].index(src[1]))]
        joins.append({"name": alias, "table": t, "keys": [key], "key_names": [bcol], "payload": payload,
                      "key_src": key_src, "perfect": None, "kept_rows": np.nonzero(keep)[0]})
    # BoundReference index of every join's probe-side condition in the original column layout
    cond_left = []
    for j in joins:
        sj, sc = j["key_src"][0]
        if sj < 0:
            cond_left.append([sc])
        else:
            cond_left.append([len(names) + sum(len(joins[i]["payload"]) for i in range(sj)) + sc])
    wl = {"name": "job_" + name, "probe": {"name": ptable, "cols": probe_cols}, "joins": joins,
          "cond_left_index": cond_left}
    # the query's select list, MIN(alias.column) ...: the columns of it that this pipeline's tables own, as VARCHAR columns
    # (wl["select"]: (join index or -1 for the probe table, column name); the strings of the rows kept, in "strings")
    select = []
    for alias, col in q.get("select", []):
        t = q["tables"][alias]
        if alias == probe:
            wl["probe"].setdefault("strings", {})[col] = tables.strings(t, col)
            select.append((-1, col))
        elif alias in index_of:
            j = joins[index_of[alias]]
            if col not in j.setdefault("strings", {}):
                j["strings"][col] = tables.strings(t, col, rows=j["kept_rows"])
            select.append((index_of[alias], col))
    wl["select"] = select
    if flt:
        wl["probe"]["filter"] = flt
        wl["probe"]["filter_sel"] = np.nonzero(probe_cols["f"] < flt[0][2])[0].astype(np.uint32)
    wl["ref"] = _reference_form(probe, probe_cols, flt, order, joins, select, wl["probe"].get("strings"))
    return wl


def _reference_form(probe, probe_cols, flt, order, joins, select=(), probe_strings=None):
    """how the same pipeline reads for the reference (as SQL over uploaded tables): one table per alias -- the probe table with its
    filter column, every build side with the rows its filters keep -- and the joins as a left-deep chain in pipeline order,
    pinned with `SET disabled_optimizers TO 'join_order,...'`: the plan the reference executes is then one pipeline whose
    source is the probe table and whose consecutive INNER hash joins are exactly these joins, so POLARConfig
    (src/parallel/polar_config.cpp:19-71) multiplexes all of them."""
    def tname(alias):
        return "t_" + alias

    def cname(alias, col):
        return "%s__%s" % (alias, col)

    tables = {tname(probe): {cname(probe, c): a for c, a in probe_cols.items()}}
    for c, vals in (probe_strings or {}).items():
        tables[tname(probe)][cname(probe, c)] = vals
    sql = "SELECT COUNT(*) FROM %s" % tname(probe)
    for (alias, bcol, src), j in zip(order, joins):
        cols = {cname(alias, bcol): j["keys"][0]}
        for c, a in j["payload"].items():
            if c != bcol:
                cols[cname(alias, c)] = a
        for c, vals in j.get("strings", {}).items():
            cols[cname(alias, c)] = vals
        tables[tname(alias)] = cols
        sql += " JOIN %s ON %s = %s" % (tname(alias), cname(src[0], src[1]), cname(alias, bcol))
    if flt:
        sql += " WHERE " + " AND ".join("%s %s %d" % (cname(probe, c), op, const) for c, op, const in flt)
    # (statistics propagation is switched off with the join order optimizer: on a reduced instance it finds build sides
    # whose key ranges miss the probe side's, and replaces such a join -- and with it the pipeline -- by an empty result)
    out = {"tables": tables, "pk": {}, "query": sql,
           "settings": ["SET disabled_optimizers TO 'join_order,statistics_propagation'"]}
    if select:
        aliases = [probe] + [a for a, _b, _s in order]
        out["query_select"] = sql.replace("SELECT COUNT(*)", "SELECT " + ", ".join(
            "MIN(%s)" % cname(aliases[1 + sj], c) for sj, c in select))
    return out
