import sys, os, ctypes as C
sys.path.insert(0, 'duckdb-polr_amd/python')
import numpy as np
from polr_amd import capi, workloads
import torch
wl = workloads.job_light_01()
ctx = capi.Context(0)
joins = capi.build_joins(ctx, wl)
probe = wl['probe']; names = list(probe['cols'].keys()); n_rows = len(probe['cols'][names[0]])
pipe = capi.Pipeline(ctx, list(probe['cols'].values()), n_rows, joins, workloads.default_paths(2))
sel = probe['filter_sel']; pipe.set_selection(sel)
mpx = capi.DeviceMultiplexer(pipe, 'adaptive_reinit', log_rounds=False)
bounds = np.searchsorted(sel, np.arange(0, n_rows + 1024, 1024, dtype=np.int64)).astype(np.uint64)
keep = np.concatenate([[True], bounds[1:] != bounds[:-1]]); offs = bounds[keep]
mpx.set_chunk_offsets(offs)
for it in range(3):
    mpx.reset(); mpx.run(0, len(offs)-1); st = mpx.finish()
L = ctx.L
L.polr_mpx_dump_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
n = 64
buf = np.zeros((n, 8), dtype=np.uint64)
L.polr_mpx_dump_stamps(mpx.h, buf.ctypes.data, n)
for i in range(32, 48):
    r = buf[i].astype(np.int64)
    if r[0] == 0: continue
    print(i, [int((r[j]-r[0])*10) if r[j] else None for j in range(1,7)], 'ns')
