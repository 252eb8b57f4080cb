import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "duckdb-polr_amd", "python"))
import numpy as np, torch
from polr_amd import capi, ssb_skew
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = capi.Context(0)
for scale, E in ((10, 16), (10, 384), (100, 384), (100, 64)):
    z = ssb_skew.sizes(scale); n = z["n_lo"]
    wl = ssb_skew.workload("q4.1", sf=scale, n_lo=n, host_probe=False)
    inst = wl["instance"]
    names = list(ssb_skew.PROBE_COLS)
    ct = inst.lineorder_torch(0, n, dev, cols=names)
    cols = [capi.dev_col(ct[c].data_ptr(), 4, signed=False) for c in names]
    paths = np.asarray([[0,1,2,3],[1,0,2,3],[1,2,0,3],[1,2,3,0]], dtype=np.int32)
    joins = capi.build_joins(ctx, wl, auto=True)
    pipe = capi.Pipeline(ctx, cols, n, joins, paths)
    n_chunks = (n + 1023)//1024
    for routing in ("adaptive_reinit", "default_path"):
        mp = [capi.DeviceMultiplexer(pipe, routing) for _ in range(E)]
        rg = [((e*n_chunks)//E, ((e+1)*n_chunks)//E) for e in range(E)]
        capi.run_resident(mp, rg, reset=True, finish=True)
        st = capi.finish_many(mp)
        cnt = sum(sum(s["stage_out"][p][3] for p in range(4)) for s in st)
        out = capi.Output(pipe, 1024, 32768)
        capi.run_resident(mp, rg, out=out, reset=True, finish=True)
        st2 = capi.finish_many(mp)
        cnt2 = sum(sum(s["stage_out"][p][3] for p in range(4)) for s in st2)
        rows, chunks, ovf = out.stats()
        print("SF%d E=%d %s: counting %d, emitting stats %d, out rows %d chunks %d overflow %s" % (scale, E, routing, cnt, cnt2, rows, chunks, ovf), flush=True)
        out.close()
        for m in mp: m.close()
    pipe.close()
    for ht,_ in joins: ht.close()
    del ct, cols; torch.cuda.empty_cache()
