"""where does the host time of one bench step go? (resident launch, 8 executors)"""
import sys, time
sys.path.insert(0, 'duckdb-polr_amd/python')
import numpy as np
from polr_amd import capi, workloads
wl = workloads.job_light_01()
ctx = capi.Context(0)
joins = capi.build_joins(ctx, wl)
probe = wl['probe']; names = list(probe['cols'].keys()); n_rows = len(probe['cols'][names[0]])
pipe = capi.Pipeline(ctx, list(probe['cols'].values()), n_rows, joins, workloads.default_paths(2))
sel = probe['filter_sel']; pipe.set_selection(sel)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bounds = np.searchsorted(sel, np.arange(0, n_rows + 1024, 1024, dtype=np.int64)).astype(np.uint64)
keep = np.concatenate([[True], bounds[1:] != bounds[:-1]]); offs = bounds[keep]
nc = len(offs) - 1
mpxs = []
for e in range(E):
    m = capi.DeviceMultiplexer(pipe, 'adaptive_reinit', log_rounds=False)
    m.set_chunk_offsets(offs)
    mpxs.append(m)
ranges = [((e * nc) // E, ((e + 1) * nc) // E) for e in range(E)]
N = 300
for it in range(10):
    capi.run_resident(mpxs, ranges, reset=True, finish=True); capi.finish_many(mpxs)
t_run = t_fin = 0.0
t0 = time.perf_counter()
for it in range(N):
    a = time.perf_counter()
    capi.run_resident(mpxs, ranges, reset=True, finish=True)
    b = time.perf_counter()
    capi.finish_many(mpxs)
    c = time.perf_counter()
    t_run += b - a; t_fin += c - b
tot = time.perf_counter() - t0
print("E=%d step %.1f us: run_resident call %.1f us, finish_many (sync + stats) %.1f us" % (E, tot / N * 1e6, t_run / N * 1e6, t_fin / N * 1e6))
for m in mpxs: m.close()
pipe.close()
