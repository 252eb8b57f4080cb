"""The driver's bench contract, checked on the lines this repository committed (profiles/r02_bench_default.json,
profiles/r03_bench_default.json = stdout of `python bench.py` on the GPU box): one JSON object with the contract's keys,
BASELINE.json's metric and unit, and the two objects this tier adds (`roofline`, `cpu_baseline`)."""
import json
import os

import common


import pytest


@pytest.mark.parametrize("line", ["r02_bench_default.json", "r03_bench_default.json"])
def test_committed_bench_line_has_the_contract_shape(line):
    path = os.path.join(common.ROOT, "profiles", line)
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
        assert {"value", "ms_per_step", "config"} <= set(sub)
    if line.startswith("r03"):
        # round 3: the traffic of the headline AND of the JOB sub-records is replayed from committed PMC summaries whose
        # signature is the run's; the sub-record with the shipped GROUP BY carries the reference's figure on the same SQL
        assert r["traffic"] and r["pmc_signature"] == json.load(open(os.path.join(
            common.ROOT, "profiles", r["traffic_source"].split("profiles/")[1])))["workload_signature"]
        subs = {sub["config"]["workload"].split(" ")[0].rstrip(":"): sub for sub in d["sub_records"]}
        for name in ("job_q18", "job_light_01"):
            rs = subs[name]["roofline"]
            assert rs["traffic"] and "replayed" in rs["traffic_source"] and rs["traffic"] >= rs["algorithmic_bytes_per_step"]
            assert subs[name]["cpu_baseline"]["kind"] == "reference"
        shipped = subs["ssb_skew_q41"]
        assert "as shipped" in shipped["config"]["workload"] and shipped["count_star"] == d["count_star"]
        assert shipped["cpu_baseline"]["kind"] == "reference" and shipped["cpu_baseline"]["value"] > 0


def test_committed_pmc_summaries_match_the_commands_bench_runs():
    """the signatures bench.py looks a replayed `roofline.traffic` up by: one committed summary for each of the default
    line's records and for the 113-pipeline pass"""
    import glob
    sigs = [json.load(open(f))["workload_signature"] for f in glob.glob(os.path.join(common.ROOT, "profiles", "r03_*_pmc_summary.json"))]
    base = {"routing": "adaptive_reinit", "n_gpus": 1}
    for want in ({"workload": "ssb_skew_q41", "scale": 100.0, "join_enumerator": "sample", "max_join_orders": 3, "executors_per_gpu": 384},
                 {"workload": "job_q18", "scale": 1.0, "join_enumerator": "each_last_once", "max_join_orders": 8, "executors_per_gpu": 32},
                 {"workload": "job_light_01", "scale": 1.0, "join_enumerator": "each_last_once", "max_join_orders": 8, "executors_per_gpu": 32},
                 {"workload": "job_full", "scale": 1.0, "join_enumerator": "each_last_once", "max_join_orders": 8, "executors_per_gpu": 16}):
        assert dict(base, **want) in sigs, want


_RANK_SCRIPT = r"""
import json, os, sys
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
import torch
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": float(t[0]), "argv": sys.argv[1:],
                      "ranks_started": int(os.environ["POLR_RANKS_STARTED"])}))
else:
    print("noise from rank %d" % rank)
dist.destroy_process_group()
sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == rank)
"""


def test_bench_starts_its_own_ranks(tmp_path, capsys):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes with the launcher's environment, a
    rendezvous on 127.0.0.1, ONE line relayed (rank 0's), non-zero when any rank fails -- two gloo ranks on the CPU"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(common.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rc = bench.spawn_ranks(2, argv=["--gpus", "2", "--steps", "3"], script=str(script))
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1
    d = json.loads(out[0])
    assert d == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2", "--steps", "3"], "ranks_started": 2}
    rc = bench.spawn_ranks(2, argv=[], script=str(script), extra_env={"FAIL_RANK": "1"})
    capsys.readouterr()
    assert rc != 0


def test_bench_refuses_more_ranks_than_gpus():
    """no launcher and fewer visible GPUs than --gpus: a clear error, never a silent one-rank measurement"""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "POLR_SHARE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr and not p.stdout.strip()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "must agree" in p.stderr
