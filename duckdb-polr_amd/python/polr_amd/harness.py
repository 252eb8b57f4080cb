"""The reference's experiment harness, restated: the four log artefacts a POLAR run leaves under <cwd>/tmp/ and the
aggregates its plotting scripts compute from them -- same file names, same column names, same text.

  tmp/<prefix><ts>.csv              PhysicalMultiplexer::WriteLogToFile (src/execution/operator/polr/
                                    physical_multiplexer.cpp:194-219): ALTERNATE -> header `path_0,path_1,...,` and one
                                    row per source chunk (trailing comma on every line); other strategies -> header
                                    `intermediates` and one line per routing round
  tmp/<prefix><ts>-intms.txt        total intermediates of the executor (src/parallel/polar_pipeline_executor.cpp:101-105)
  tmp/<prefix><ts>-<hash>.csv       pipeline duration in ms, no newline (src/parallel/pipeline.cpp:247-263; <hash> =
                                    std::hash of the source operator's parameter string there, a stable hash of the
                                    source's description here)
  tmp/<prefix><ts>-enumeration.csv  `num_joins,enumeration_time_ms` (src/parallel/polar_config.cpp:231-246)

<ts> = a steady-clock count, as in the reference (one set per executor and run).  Aggregates of
experiments/scripts/plot_1_1_sel_intms.py:25-32 over an ALTERNATE matrix: default = sum of path_0, exhaustive = sum of
the row-wise minima, best / worst in class = smallest / largest column sum; plot_2_3_routing_dur.py:22-28: median
pipeline duration.
"""
import hashlib
import os
import time

import numpy as np


def format_log(routing, per_round_intermediates, n_paths):
    """text of tmp/<ts>.csv from the multiplexer's log (device: polr_mpx_fetch_log's intermediates)"""
    inter = [int(x) for x in per_round_intermediates]
    if routing == "alternate":
        assert len(inter) % n_paths == 0
        lines = ["".join("path_%d," % p for p in range(n_paths))]
        for c in range(len(inter) // n_paths):
            lines.append("".join("%d," % inter[c * n_paths + p] for p in range(n_paths)))
        return "\n".join(lines) + "\n"
    return "intermediates\n" + "".join("%d\n" % x for x in inter)


def source_hash(description):
    """stands in for std::hash<string>(source->ParamsToString()) (pipeline.cpp:255-256): stable across runs"""
    return int.from_bytes(hashlib.sha256(description.encode()).digest()[:8], "little")


def write_artefacts(directory, prefix, routing, n_paths, per_round_intermediates, total_intermediates, pipeline_ms,
                    enumeration_ms, num_joins, source_description, log_tuples_routed=True, measure_pipeline=True):
    """writes the files a reference run with enable_log_tuples_routed / enable_measure_pipeline leaves; returns
    {kind: path}"""
    tmp = os.path.join(directory, "tmp")
    os.makedirs(tmp, exist_ok=True)
    out = {}
    if log_tuples_routed:
        ts = str(time.monotonic_ns())
        out["log"] = os.path.join(tmp, prefix + ts + ".csv")
        with open(out["log"], "w") as f:
            f.write(format_log(routing, per_round_intermediates, n_paths))
        out["intms"] = os.path.join(tmp, prefix + ts + "-intms.txt")
        with open(out["intms"], "w") as f:
            f.write("%d\n" % int(total_intermediates))
        ts2 = str(time.monotonic_ns())
        out["enumeration"] = os.path.join(tmp, prefix + ts2 + "-enumeration.csv")
        with open(out["enumeration"], "w") as f:
            f.write("num_joins,enumeration_time_ms\n%d,%s\n" % (num_joins, repr(float(enumeration_ms))))
    if measure_pipeline:
        ts3 = str(time.monotonic_ns())
        out["duration"] = os.path.join(tmp, prefix + ts3 + "-" + str(source_hash(source_description)) + ".csv")
        with open(out["duration"], "w") as f:
            f.write(repr(float(pipeline_ms)))
    return out


def read_alternate_csv(path_or_text):
    text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
    lines = text.strip().splitlines()
    cols = [c for c in lines[0].split(",") if c]
    m = np.asarray([[int(x) for x in l.rstrip(",").split(",")] for l in lines[1:]], dtype=np.int64).reshape(-1, len(cols))
    return cols, m


def aggregates(matrices):
    """plot_1_1_sel_intms.py:25-32 over a list of ALTERNATE matrices (one per logged pipeline)"""
    return {"default": [int(m[:, 0].sum()) for m in matrices],
            "exhaustive": [int(m.min(axis=1).sum()) for m in matrices],
            "best_in_class": [int(m.sum(axis=0).min()) for m in matrices],
            "worst_in_class": [int(m.sum(axis=0).max()) for m in matrices]}


def median_duration(paths):
    """plot_2_3_routing_dur.py:22-28: median of the pipeline durations of the runs"""
    return float(np.median([float(open(p).read()) for p in paths]))
