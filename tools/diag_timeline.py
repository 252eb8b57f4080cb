"""diagnostic: reads the per-unit records of one instrumented pass (POLR_DIAG_TIMELINE=<file.npz> python bench.py ...,
library from `make -C duckdb-polr_amd diag`) and prints where the probe waves' time went.

records per wave: [began waiting, got the unit, finished it, exec << 40 | path << 32 | count], 100 MHz wall clock"""
import sys

import numpy as np

f = np.load(sys.argv[1])
tl, used = f["tl"], f["used"]
waves = np.nonzero(used)[0]
print("probe waves that ran units: %d of %d; units: %d" % (len(waves), len(used), int(used.sum())))
rec = np.concatenate([tl[w, :used[w]] for w in waves]).astype(np.int64)
wait0, got, done, meta = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
count = meta & 0xFFFFFFFF
path = (meta >> 32) & 0xFF
probe_q = (meta >> 56) & 0xFF  # quarter microseconds between "got the unit" and "pipeline empty" (saturates at 63.75)
t0 = wait0.min()
t1 = done.max()
span = (t1 - t0) / 100.0
print("span %.1f us (first wave waiting .. last unit done)" % span)
busy = (done - got).sum() / 100.0
waiting = (got - wait0).sum() / 100.0
tot = len(waves) * span
print("wave time: busy %.1f%%, waiting for a unit %.1f%%, before first wait / after last unit %.1f%%" %
      (100 * busy / tot, 100 * waiting / tot, 100 * (tot - busy - waiting) / tot))
print("tuples: %d; per-path tuples: %s" % (count.sum(), {int(p): int(count[path == p].sum()) for p in np.unique(path)}))
# unit size classes
for lo, hi in ((0, 512), (512, 4096), (4096, 16384), (16384, 1 << 30)):
    m = (count > lo) & (count <= hi)
    if m.any():
        d = (done - got)[m] / 100.0
        print("units of (%d, %d] tuples: %d, tuples %d, busy %.1f%% of all busy, mean %.2f us (probing %.2f us, rest = "
              "counters + arrival), ns/tuple %.3f" %
              (lo, hi, m.sum(), count[m].sum(), 100 * d.sum() / busy, d.mean(), probe_q[m].mean() / 4.0,
               1e3 * d.sum() / count[m].sum()))
for p in np.unique(path):
    m = (path == p) & (count >= 4096)
    if m.any():
        d = (done - got)[m] / 100.0
        print("path %d big units: %d, ns/tuple/wave %.3f" % (p, m.sum(), 1e3 * d.sum() / count[m].sum()))
# utilisation over time: 20 bins
nb = 20
edges = np.linspace(t0, t1, nb + 1)
line = []
for b in range(nb):
    a, z = edges[b], edges[b + 1]
    ov = np.clip(np.minimum(done, z) - np.maximum(got, a), 0, None).sum()
    line.append(ov / ((z - a) * len(waves)))
print("busy fraction per %.0f us bin: %s" % (span / nb, " ".join("%.2f" % x for x in line)))
tup = []
for b in range(nb):
    a, z = edges[b], edges[b + 1]
    m = (done > a) & (done <= z)
    tup.append(count[m].sum())
print("M tuples finished per bin:       %s" % " ".join("%.1f" % (x / 1e6) for x in tup))
# the tail: when did each wave finish its last unit
last = np.array([tl[w, used[w] - 1, 2] for w in waves], dtype=np.int64)
q = np.percentile((last - t0) / 100.0, [1, 10, 50, 90, 99, 100])
print("last unit done per wave (us): p1 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(q))
