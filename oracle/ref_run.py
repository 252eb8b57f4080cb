"""oracle/ref_run.py -- drive the compiled reference (oracle/_ref/ref_driver).  TEST INFRASTRUCTURE:
used by tests/golden/make_golden.py and by bench.py's cpu_baseline leg only."""
import glob
import os
import shutil
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(_HERE, "_ref", "ref_driver")

SQLTYPE = {"int32": "INTEGER", "uint32": "UINTEGER", "int64": "BIGINT", "uint64": "UBIGINT", "int16": "SMALLINT",
           "uint16": "USMALLINT", "int8": "TINYINT", "uint8": "UTINYINT"}


def available():
    return os.path.exists(DRIVER) and os.path.exists(os.path.join(_HERE, "_ref", "libduckdb_ref.so"))


def table_lines(workdir, name, cols, pk=None):
    """pk: name of the column declared PRIMARY KEY (as benchmark/ssb-skew/init/load.sql does for dimension keys)"""
    n = len(next(iter(cols.values())))
    lines = ["table %s %d" % (name, n)]
    for cname, arr in cols.items():
        path = os.path.join(workdir, "%s.%s.bin" % (name, cname))
        if isinstance(arr, (list, tuple)):  # a VARCHAR column: bytes / str per row
            with open(path, "wb") as f:
                for v in arr:
                    v = v.encode() if isinstance(v, str) else bytes(v)
                    f.write(np.uint32(len(v)).tobytes() + v)
            lines.append("col %s VARCHAR %s" % (cname, path))
            continue
        np.ascontiguousarray(arr).tofile(path)
        lines.append("col %s %s %s%s" % (cname, SQLTYPE[str(arr.dtype)], path, " pk" if cname == pk else ""))
    lines.append("endtable")
    return lines


def time_polar_pipeline(tables, query, settings, threads, repeat=5, pk=None):
    """Loads `tables` ({name: {col: ndarray}}), runs `query` `repeat` times with POLAR on and
    PRAGMA enable_measure_pipeline; returns the list of POLAR-pipeline durations in ms
    (Pipeline::Schedule -> Finalize, src/parallel/pipeline.cpp:138,247-263) and the wall ms of each run."""
    workdir = tempfile.mkdtemp(prefix="polr_cpu_baseline_")
    try:
        lines = []
        for name, cols in tables.items():
            lines += table_lines(workdir, name, cols, pk=(pk or {}).get(name))
        lines.append("sql SET threads TO %d" % threads)
        for s in settings:
            lines.append("sql " + s)
        lines.append("sql PRAGMA enable_polr")
        lines.append("sql PRAGMA enable_measure_pipeline")
        lines.append("repeat %d q %s" % (repeat, query))
        script = os.path.join(workdir, "script.txt")
        with open(script, "w") as f:
            f.write("\n".join(lines) + "\n")
        outdir = os.path.join(workdir, "out")
        proc = subprocess.run([DRIVER, script, outdir], capture_output=True, text=True, check=False)
        if proc.returncode != 0:
            raise RuntimeError("reference driver failed:\n" + proc.stdout + proc.stderr)
        files = sorted(glob.glob(os.path.join(outdir, "tmp", "*-*.csv")), key=os.path.getmtime)
        files = [f for f in files if not f.endswith("-enumeration.csv")]
        ms = [float(open(f).read().strip()) for f in files]
        wall = [float(l.split("wall_ms=")[1]) for l in proc.stdout.splitlines() if l.startswith("query q ")]
        result = open(os.path.join(outdir, "q.csv")).read()
        return ms, wall, result
    finally:
        shutil.rmtree(workdir, ignore_errors=True)


def sweep_polar_pipeline(tables, query, settings, thread_counts, repeat=3, pk=None):
    """Loads `tables` ONCE and runs `query` `repeat` times at every thread count of `thread_counts` (SET threads TO n) with
    POLAR on and PRAGMA enable_measure_pipeline.  Returns {threads: (pipeline ms of each run, the answer of the last run:
    COUNT(*) itself for a one-cell result, else the CRC-32 of the result's rows)}."""
    workdir = tempfile.mkdtemp(prefix="polr_cpu_baseline_")
    try:
        lines = []
        for name, cols in tables.items():
            lines += table_lines(workdir, name, cols, pk=(pk or {}).get(name))
        for s_ in settings:
            lines.append("sql " + s_)
        lines.append("sql PRAGMA enable_polr")
        lines.append("sql PRAGMA enable_measure_pipeline")
        for t in thread_counts:
            lines.append("sql SET threads TO %d" % t)
            lines.append("repeat %d q_t%d %s" % (repeat, t, query))
        script = os.path.join(workdir, "script.txt")
        with open(script, "w") as f:
            f.write("\n".join(lines) + "\n")
        outdir = os.path.join(workdir, "out")
        proc = subprocess.run([DRIVER, script, outdir], capture_output=True, text=True, check=False)
        if proc.returncode != 0:
            raise RuntimeError("reference driver failed:\n" + proc.stdout + proc.stderr)
        files = sorted(glob.glob(os.path.join(outdir, "tmp", "*-*.csv")), key=lambda f_: (os.path.getmtime(f_), f_))
        files = [f_ for f_ in files if not f_.endswith("-enumeration.csv")]
        ms = [float(open(f_).read().strip()) for f_ in files]
        out = {}
        for i, t in enumerate(thread_counts):
            runs = ms[i * repeat:(i + 1) * repeat] if len(ms) == repeat * len(thread_counts) else []
            cell = None
            try:
                rows = open(os.path.join(outdir, "q_t%d.csv" % t)).read().strip().splitlines()[1:]
                if len(rows) == 1 and "," not in rows[0]:
                    cell = int(rows[0])  # COUNT(*)
                else:
                    import zlib
                    cell = zlib.crc32("\n".join(rows).encode())  # a whole result: its checksum stands for it
            except Exception:
                pass
            out[t] = (runs, cell)
        return out
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
