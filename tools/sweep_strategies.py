#!/usr/bin/env python3
"""All routing strategies on the default bench workload (runs ON THE GPU BOX): one line per strategy."""
import json
import subprocess
import sys

rows = []
for routing in ["adaptive_reinit", "init_once", "opportunistic", "dynamic", "exponential_backoff", "default_path"]:
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "40", "--routing", routing] + sys.argv[1:],
                         capture_output=True, text=True)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    rows.append((routing, d["ms_per_step"], d["value"] / 1e9, d["routing_rounds"], d["total_intermediates"],
                 d["roofline"]["kernel_ms_per_step"], d["count_star"]))
    print("| %s | %.4f | %.1f | %d | %d | %.4f | %d |" % rows[-1], flush=True)
