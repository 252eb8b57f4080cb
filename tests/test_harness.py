"""The reference's log artefacts and aggregates (polr_amd/harness.py) against the files the reference itself wrote for the
SSB-skew Q4.1 fixture run (tests/golden/ssb_skew_sample.json `raw_files`: verbatim text of its tmp/<ts>.csv and
tmp/<ts>-intms.txt, the header of its tmp/<ts>-enumeration.csv and the names of the files it left)."""
import os
import re

import numpy as np
import pytest

import common
from common import orc
from polr_amd import harness, ssb_skew

GOLD = common.load_golden("ssb_skew_sample")
RAW = GOLD["raw_files"]
CASE = GOLD["cases"]["q4.1/3"]


def _oracle(routing):
    wl = ssb_skew.workload("q4.1", **GOLD["shape"])
    pcols, pvalid, joins = common.oracle_joins(wl)
    return orc.run_pipeline(pcols, joins, np.asarray(CASE["paths"], dtype=np.int32), routing=routing, caching=False,
                            collect_output=False)


def test_log_text_equals_the_references_files(tmp_path):
    alt = _oracle("alternate")
    text = harness.format_log("alternate", alt["alt_matrix"].reshape(-1), len(CASE["paths"]))
    assert text == RAW["alternate_log"]
    ada = _oracle("adaptive_reinit")
    assert harness.format_log("adaptive_reinit", ada["intermediates_per_round"], len(CASE["paths"])) == RAW["adaptive_log"]
    files = harness.write_artefacts(str(tmp_path), "", "adaptive_reinit", len(CASE["paths"]),
                                    ada["intermediates_per_round"], ada["num_intermediates"], 12.5, 0.25, 4,
                                    "lineorder scan")
    assert open(files["log"]).read() == RAW["adaptive_log"]
    assert open(files["intms"]).read() == RAW["adaptive_intms"]
    assert open(files["enumeration"]).read().splitlines()[0] == RAW["enumeration_header"]
    assert float(open(files["duration"]).read()) == 12.5 and not open(files["duration"]).read().endswith("\n")
    # the same four kinds of names the reference left: <ts>.csv, <ts>-intms.txt, <ts>-enumeration.csv, <ts>-<hash>.csv
    def kinds(names):
        out = set()
        for n in names:
            n = os.path.basename(n)
            out.add("enumeration" if n.endswith("-enumeration.csv") else "intms" if n.endswith("-intms.txt") else
                    "duration" if re.fullmatch(r"\d+-\d+\.csv", n) else "log" if re.fullmatch(r"\d+\.csv", n) else n)
        return out
    assert kinds(files.values()) == kinds(RAW["adaptive_file_names"]) == {"log", "intms", "enumeration", "duration"}


def test_dir_prefix_and_switches(tmp_path):
    files = harness.write_artefacts(str(tmp_path), "sub", "default_path", 2, [5, 6], 11, 1.0, 0.1, 2, "x",
                                    log_tuples_routed=True, measure_pipeline=False)
    assert set(files) == {"log", "intms", "enumeration"}
    assert all(os.path.basename(p).startswith("sub") for p in files.values())
    files = harness.write_artefacts(str(tmp_path), "", "default_path", 2, [5, 6], 11, 1.0, 0.1, 2, "x",
                                    log_tuples_routed=False, measure_pipeline=True)
    assert set(files) == {"duration"}


def test_aggregates_of_the_plotting_scripts():
    cols, m = harness.read_alternate_csv(RAW["alternate_log"])
    assert cols == ["path_%d" % p for p in range(len(CASE["paths"]))]
    assert np.array_equal(m, np.asarray(CASE["alternate"]["matrix"]))
    agg = harness.aggregates([m])
    # plot_1_1_sel_intms.py:25-32 (pandas there): default = df["path_0"].sum(), exhaustive = df.min(axis=1).sum(),
    # best / worst in class = df.sum().min() / .max()
    assert agg["default"] == [int(m[:, 0].sum())]
    assert agg["exhaustive"] == [int(sum(min(r) for r in m.tolist()))]
    assert agg["best_in_class"] == [min(int(m[:, p].sum()) for p in range(m.shape[1]))]
    assert agg["worst_in_class"] == [max(int(m[:, p].sum()) for p in range(m.shape[1]))]
    assert agg["exhaustive"][0] <= agg["best_in_class"][0] <= agg["default"][0] <= agg["worst_in_class"][0]
    # ALTERNATE's total intermediates = the sum of the whole matrix = what the reference wrote to -intms.txt
    assert int(m.sum()) == int(RAW["alternate_intms"])
