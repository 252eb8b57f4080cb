#!/usr/bin/env python3
"""tools/make_job_shapes.py -- BUILD CONTAINER ONLY.  Reads the 113 JOB queries of the reference
(benchmark/imdb_plan_cost/queries/*.sql, the SQL behind benchmark/imdb/*.benchmark) AS DATA and writes the join
*shapes* bench.py's config 4 needs to duckdb-polr_amd/python/polr_amd/job_shapes.json: per query the tables, the
equi-join graph (as column equivalence classes), per table how many filter predicates of which kind it carries, and which
columns the query's MIN(...) select list names.
No SQL text, literal or identifier beyond table / column names is kept."""
import glob
import json
import os
import re
import sys

REF = os.environ.get("POLR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "duckdb-polr_amd", "python", "polr_amd",
                   "job_shapes.json")


def split_top_level_and(where):
    where = re.sub(r"\bBETWEEN\b(.+?)\bAND\b", lambda m: "BETWEEN" + m.group(1) + "&&", where, flags=re.I | re.S)
    parts, depth, cur = [], 0, []
    tokens = re.split(r"(\(|\)|\bAND\b)", where, flags=re.I)
    for t in tokens:
        if t == "(":
            depth += 1
        elif t == ")":
            depth -= 1
        if depth == 0 and t.upper() == "AND":
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(t)
    if "".join(cur).strip():
        parts.append("".join(cur).strip())
    return parts


def classify(pred):
    p = pred.upper()
    if " OR " in p:
        return "or"
    if " NOT LIKE " in p:
        return "notlike"
    if " LIKE " in p:
        return "like"
    if " NOT IN" in p:
        return "notin"
    m = re.search(r"\bIN\s*\((.*)\)", pred, flags=re.I | re.S)
    if m:
        return "in%d" % (m.group(1).count(",") + 1)
    if "BETWEEN" in p:
        return "between"
    if "IS NOT NULL" in p:
        return "notnull"
    if "IS NULL" in p:
        return "null"
    if "!=" in p or "<>" in p:
        return "ne"
    if ">" in p or "<" in p:
        return "range"
    return "eq"


def parse(sql):
    sql = sql.strip().rstrip(";")
    m = re.search(r"\bFROM\b(.*?)\bWHERE\b(.*)$", sql, flags=re.I | re.S)
    from_part, where = m.group(1), m.group(2)
    tables = {}
    for item in from_part.split(","):
        toks = item.split()
        name = toks[0]
        alias = toks[-1] if len(toks) > 1 else toks[0]
        tables[alias] = name
    joins, filters = [], {a: [] for a in tables}
    for pred in split_top_level_and(where):
        pred = pred.strip()
        jm = re.fullmatch(r"(\w+)\.(\w+)\s*=\s*(\w+)\.(\w+)", pred)
        if jm and jm.group(1) in tables and jm.group(3) in tables:
            joins.append([[jm.group(1), jm.group(2)], [jm.group(3), jm.group(4)]])
            continue
        aliases = set(a for a in re.findall(r"\b(\w+)\.\w+", pred) if a in tables)
        kind = classify(pred)
        for a in aliases:
            filters[a].append(kind)
    # the select list: every JOB query returns MIN(alias.column) of a few (mostly VARCHAR) columns
    select = [[a, c] for a, c in re.findall(r"\bMIN\s*\(\s*(\w+)\.(\w+)\s*\)", sql[:m.start()], flags=re.I) if a in tables]
    return {"tables": tables, "joins": joins, "filters": {a: f for a, f in filters.items() if f}, "select": select}


def main():
    files = sorted(glob.glob(os.path.join(REF, "benchmark", "imdb_plan_cost", "queries", "*.sql")))
    shapes = {}
    for f in files:
        name = os.path.basename(f)[:-4]
        shapes[name] = parse(open(f).read())
    assert len(shapes) == 113, len(shapes)
    json.dump({"source": "benchmark/imdb_plan_cost/queries/*.sql (113 JOB queries), shapes only", "queries": shapes},
              open(OUT, "w"), separators=(",", ":"), sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", sum(len(s["joins"]) for s in shapes.values()), "join predicates")


if __name__ == "__main__":
    main()
