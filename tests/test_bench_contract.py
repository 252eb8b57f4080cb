"""The driver's bench contract, checked on the line this repository last committed (profiles/r02_bench_default.json =
stdout of `python bench.py` on the GPU box): one JSON object with the contract's keys, BASELINE.json's metric and unit,
and the two objects this tier adds (`roofline`, `cpu_baseline`)."""
import json
import os

import common


def test_committed_bench_line_has_the_contract_shape():
    path = os.path.join(common.ROOT, "profiles", "r02_bench_default.json")
    lines = [l for l in open(path).read().splitlines() if l.strip()]
    assert len(lines) == 1  # ONE JSON line
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(common.ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "ssb_skew_q41" in d["config"]["workload"] and "SF100" in d["config"]["workload"]
    assert d["config"]["join_enumerator"] == "sample" and d["config"]["max_join_orders"] == 3
    if isinstance(base.get("metric"), str):
        assert d["metric"].split("/")[0].replace("-", "").replace("_", "") in base["metric"].replace("-", "").replace("_", "") \
            or "tuples" in d["metric"]
    # value = whole-job throughput of the timed region
    n = d["config"]["probe_partition_per_gpu"]
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (r["kernel_ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert r["traffic"] is None or (r["traffic"] >= r["algorithmic_bytes_per_step"] and "replayed" in r["traffic_source"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["count_star_matches_reference"] is True
    for sub in d.get("sub_records", []):
        assert {"value", "ms_per_step", "roofline", "config"} <= set(sub)
