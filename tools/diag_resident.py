"""diagnostic (needs a -DPOLR_DIAG_STAMPS build of polr_mpx.o + polr_resident_k2.o): phase stamps of a
resident run -- per routing step: router step begin / published / all arrived, first worker saw / arrived"""
import sys, os, ctypes as C
sys.path.insert(0, 'duckdb-polr_amd/python')
import numpy as np
from polr_amd import capi, workloads
wl = workloads.job_light_01()
ctx = capi.Context(0)
joins = capi.build_joins(ctx, wl)
probe = wl['probe']; names = list(probe['cols'].keys()); n_rows = len(probe['cols'][names[0]])
pipe = capi.Pipeline(ctx, list(probe['cols'].values()), n_rows, joins, workloads.default_paths(2))
sel = probe['filter_sel']; pipe.set_selection(sel)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bounds = np.searchsorted(sel, np.arange(0, n_rows + 1024, 1024, dtype=np.int64)).astype(np.uint64)
keep = np.concatenate([[True], bounds[1:] != bounds[:-1]]); offs = bounds[keep]
nc = len(offs) - 1
mpxs = []
for e in range(E):
    m = capi.DeviceMultiplexer(pipe, 'adaptive_reinit', log_rounds=True)
    m.set_chunk_offsets(offs)
    mpxs.append(m)
ranges = [((e * nc) // E, ((e + 1) * nc) // E) for e in range(E)]
for it in range(3):
    for m in mpxs: m.reset()
    capi.run_resident(mpxs, ranges, reset=True, finish=True)
    capi.finish_many(mpxs)
L = ctx.L
L.polr_mpx_dump_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
n = 40
for e in range(min(E, 2)):
    buf = np.zeros((n, 8), dtype=np.uint64)
    L.polr_mpx_dump_stamps(mpxs[e].h, buf.ctypes.data, n)
    path, tuples, inter = mpxs[e].fetch_log()
    print("executor", e, "rounds:", list(zip(path.tolist(), tuples.tolist())))
    t00 = int(buf[0][0])
    print("router entry %.2f us before its first step, router exit at %.2f us" % ((t00 - int(buf[0][5])) / 100.0, (int(buf[0][6]) - t00) / 100.0))
    for i in range(n):
        r = buf[i].astype(np.int64)
        if r[0] == 0: continue
        f = lambda v: "%.2f" % ((int(v) - t00) / 100.0) if v else "-"
        print(i, "step_begin", f(r[0]), "arrived(prev)", f(r[2]), "published", f(r[1]), "| worker1: seen", f(r[3]), "arrived", f(r[4]))
