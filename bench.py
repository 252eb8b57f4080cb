#!/usr/bin/env python3
"""bench.py -- probe-tuples/s of the POLAR multiplexed hash-join pipeline on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload ssb_skew_q41 --scale 100] [--routing adaptive_reinit]

One *step* = one complete pass of the hot path over the workload's probe side: fresh multiplexers route every source
chunk (device-resident routers), the pool of probe waves sends the routed slices through the bank of join orders and
feeds the per-join counters back into the reward -- inputs (probe columns, selection, build tables) already resident
in HBM when the clock starts.  The sink is COUNT(*) (only counters leave the device).

Default workload = BASELINE.json configs[2], the configuration the metric's roofline is quoted on: SSB-skew Q4.1 at
SF100 -- lineorder 600 M rows x {customer, supplier, part, date}, the skew of benchmark/ssb-skew/init/load.sql:80-253
applied exactly (polr_amd/ssb_skew.py), generated on the device; join orders from the reference's default enumerator
(`sample`, max_join_orders = 3).  At N = 1 the same JSON line carries `sub_records` for the JOB 18a shape (the query
the >= 10x target is quoted on) and the JOB-light 01 shape (configs[1]).

Multi-GPU (torchrun, one rank per GPU): the path shards by probe partition -- rank r owns the r-th contiguous
lo_orderkey range of a lineorder N times as long (weak scaling; `--strong`: the N-th part of the same table) and its own
multiplexers, exactly like one PipelineExecutor per thread in the reference; the build sides are built on rank 0 and
broadcast once over RCCL before the clock starts; no collective on the data path.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = the pool kernel, algorithmic
bytes per SURVEY.md 8(d) over HIP-event kernel time) and `cpu_baseline` (the reference itself, compiled from its
sources by oracle/ref_build.mk, timed on this box's host cores on a stated contiguous sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "duckdb-polr_amd", "python"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
SSB_QUERIES = {"ssb_skew_q41": "q4.1", "ssb_skew_q42": "q4.2", "ssb_skew_q43": "q4.3", "ssb_skew_q31": "q3.1",
               "ssb_skew_q21": "q2.1"}


class _DevBuf:
    """zero-copy view of a library-owned device buffer for torch (RCCL broadcast)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def ref_capacity(n):
    """PointerTableCapacity of the reference's chained table (join_hashtable.hpp:265-267)"""
    want = max(2 * n, (262136 // 8) + 1)
    p = 1
    while p < want:
        p <<= 1
    return p


def find_pmc_summary(sig):
    """The committed PMC summary (profiles/r??_*_pmc_summary.json, newest round first) of exactly this command, if any:
    (summary, file name).  Written by tools/publish_profiles_r03.py from separate rocprofv3 --pmc passes."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        if pmc.get("workload_signature") == sig and pmc.get("traffic_bytes_per_launch_corrected"):
            return pmc, os.path.basename(f)
    return None, None


def algorithmic_bytes(joins_info, paths, tuples_per_path, stage_out):
    """per input tuple of join j:  chained  K + 8 + c(1+K+8) + m(P+4+P),
    perfect  K + 1 + m(4+4);  c = m + n_build/capacity (matching rows + expected bucket collisions at
    the reference's load factor), m = stage outputs / stage inputs, P = payload bytes gathered (0: the
    sink is COUNT(*), nothing is materialised).  Summed over every routed round."""
    total = 0.0
    for p, path in enumerate(paths):
        inp = float(tuples_per_path[p])
        for pos, j in enumerate(path):
            out = float(stage_out[p][pos])
            info = joins_info[j]
            K = info["key_bytes"]
            if info["perfect"]:
                total += inp * (K + 1) + out * 8
            else:
                alpha = info["n_rows"] / float(ref_capacity(info["n_rows"]))
                total += inp * (K + 8 + alpha * (1 + K + 8)) + out * ((1 + K + 8) + 4)
            inp = out
    return total


def chunk_offsets_for(sel, n_rows, V):
    """source chunk boundaries in selection order: the scan emits one (thinned) chunk per V-row
    vector, never an empty one"""
    if sel is None:
        return None
    bounds = np.searchsorted(sel, np.arange(0, n_rows + V, V, dtype=np.int64)).astype(np.uint64)
    keep = np.concatenate([[True], bounds[1:] != bounds[:-1]])
    return bounds[keep]


def thread_ladder():
    """1, 2, 4, ... up to this box's host cores (and the core count itself)"""
    nproc = os.cpu_count() or 1
    ladder, t = [], 1
    while t < nproc:
        ladder.append(t)
        t *= 2
    ladder.append(nproc)
    return ladder


def reference_runs(tables, pk, query, settings, n_tuples, repeat=5, threads=None):
    """the reference's POLAR pipeline on `tables` (loaded once) at every thread count of the ladder 1, 2, 4, ... host
    cores: {threads: (median ms, count)}"""
    from oracle import ref_run
    ladder = threads or thread_ladder()
    try:
        got = ref_run.sweep_polar_pipeline(tables, query, settings, ladder, repeat=repeat, pk=pk)
    except RuntimeError as e:
        # (the reference has been seen to die at high thread counts on some plans: keep what one thread measures)
        print("cpu_baseline: the reference failed somewhere on the thread ladder %s (%s); measuring one thread" %
              (ladder, str(e)[-400:].replace("\n", " | ")), file=sys.stderr, flush=True)
        got = ref_run.sweep_polar_pipeline(tables, query, settings, [1], repeat=repeat, pk=pk)
    return {t: ((float(np.median(ms)) if ms else None), count) for t, (ms, count) in got.items()}


def pick_baseline(runs, n_tuples):
    """best thread count among the runs whose COUNT(*) equals the single-threaded answer (the reference's POLAR pipeline
    has been seen to lose rows with many threads; such a run is reported, never used as the baseline)"""
    detail = {}
    ref_count = runs.get(1, (None, None))[1]
    best = None
    for threads, (ms, count) in sorted(runs.items()):
        detail["threads_%d_pipeline_ms" % threads] = None if ms is None else round(ms, 3)
        detail["count_star_threads_%d" % threads] = count
        if ms is None or count != ref_count:
            continue
        v = n_tuples / (ms / 1e3)
        if best is None or v > best[0]:
            best = (v, threads)
    return best, detail


def cpu_baseline_generic(wl, routing, n_tuples, enumerator):
    """JOB-shaped workloads: the reference on the whole workload (a few tens of ms per run)"""
    from oracle import ref_run
    if not (ref_run.available() and "ref" in wl):
        return None
    ref = wl["ref"]
    settings = list(ref["settings"]) + ["SET multiplexer_routing TO '%s'" % routing,
                                        "SET join_enumerator TO '%s'" % enumerator]
    runs = reference_runs(ref["tables"], ref.get("pk", {}), ref["query"], settings, n_tuples)
    best, detail = pick_baseline(runs, n_tuples)
    if not best:
        return None
    return {"value": best[0], "unit": "probe-tuples/s", "cores": best[1], "kind": "reference",
            "sample": "whole workload, median of 5 runs of the reference's POLAR pipeline (Pipeline::Schedule->Finalize "
                      "incl. table scan + pushed-down filter + count sink) at every thread count of %s; only runs whose "
                      "COUNT(*) equals the single-threaded answer are candidates; `cores` = the best of them" % thread_ladder(),
            "thread_sweep_tuples_per_s": {str(t): (None if ms is None else round(n_tuples / (ms / 1e3), 1))
                                          for t, (ms, _c) in sorted(runs.items())}, **detail}


def cpu_baseline_ssb(inst, query, routing, args, sample_rows, dev=None, shipped=False):
    """SSB-skew: the reference on two CONTIGUOUS samples of lineorder, one from each skew phase (rows just below and
    just above the lo_orderkey 400 M mark would mix the phases; the samples start at 1/3 and at 5/6 of the table), full
    dimension tables with their PRIMARY KEYs, the same query, `sample` enumerator, same max_join_orders.  The table is
    2/3 phase one and 1/3 phase two: probe-tuples/s = 1 / (2/3 t1 + 1/3 t2) with t_i the measured seconds per tuple."""
    from oracle import ref_run
    from polr_amd import ssb_skew
    if not ref_run.available():
        return None
    n = (min(sample_rows, inst.n_lo // 8) // 1024) * 1024  # whole source chunks
    starts = [inst.n_lo // 3 - n // 2, (inst.n_lo * 5) // 6 - n // 2]
    starts = [max(0, (s // 1024) * 1024) for s in starts]
    per_thread = {}
    detail = {"sample_rows_each": n, "sample_starts": starts}
    counts = []
    for ph, s0 in enumerate(starts):
        want_cols = list(ssb_skew.PROBE_COLS) + (["lo_revenue", "lo_supplycost"] if shipped else [])
        if dev is not None:
            # (the same rows, generated on the device and copied back: the generator is one arithmetic in numpy and in
            # torch -- tests/test_ssb_skew.py -- and 64 M rows take seconds there instead of a minute on the host)
            ct = inst.lineorder_torch(s0, s0 + n, dev, cols=want_cols)
            cols = {c: ct[c].cpu().numpy().view(np.uint32) for c in want_cols}
            del ct
        else:
            cols = inst.lineorder(s0, s0 + n, cols=want_cols)
        if shipped:
            for c in ("lo_revenue", "lo_supplycost"):  # (INTEGER columns in SSB: their difference may be negative)
                cols[c] = cols[c].astype(np.int32)
        ref = ssb_skew.reference_form(inst, query, cols)
        if shipped:
            # the query as the reference ships it (benchmark/ssb-skew/queries/q4-1.sql); the count of its rows stands in for
            # COUNT(*) in the checks below (the number of groups: equal across thread counts iff no group is lost)
            ref["query"] = ref["query"].replace("SELECT COUNT(*)", "SELECT d_year, c_nation, SUM(lo_revenue - lo_supplycost) "
                                                "AS profit") + " GROUP BY d_year, c_nation ORDER BY d_year, c_nation"
        settings = list(ref["settings"]) + ["SET multiplexer_routing TO '%s'" % routing,
                                            "SET join_enumerator TO '%s'" % args.enumerator_name,
                                            "SET max_join_orders TO %d" % args.max_join_orders]
        # (the GROUP BY form's ladder stops at 32 threads: beyond, the reference's POLAR pipeline has not returned a right
        # answer on any workload, and here only the NUMBER of groups stands in for COUNT(*) in the validity check)
        ladder = [t for t in thread_ladder() if t <= 32] if shipped else None
        runs = reference_runs(ref["tables"], ref["pk"], ref["query"], settings, n, repeat=3, threads=ladder)
        ref_count = runs.get(1, (None, None))[1]
        counts.append(ref_count)
        for threads, (ms, count) in runs.items():
            detail["phase%d_threads_%d_pipeline_ms" % (ph + 1, threads)] = None if ms is None else round(ms, 3)
            detail["phase%d_count_star_threads_%d" % (ph + 1, threads)] = count
            if ms is not None and count == ref_count:
                per_thread.setdefault(threads, {})[ph] = ms / 1e3 / n
    best = None
    for threads, t in per_thread.items():
        if len(t) == 2:
            v = 1.0 / (2.0 / 3.0 * t[0] + 1.0 / 3.0 * t[1])
            if best is None or v > best[0]:
                best = (v, threads)
    if not best:
        return None
    sweep = {}
    for threads, t in sorted(per_thread.items()):
        if len(t) == 2:
            sweep[str(threads)] = round(1.0 / (2.0 / 3.0 * t[0] + 1.0 / 3.0 * t[1]), 1)
    return {"value": best[0], "unit": "probe-tuples/s", "cores": best[1], "kind": "reference",
            "sample": "two contiguous samples of %d lineorder rows (rows %d.. of phase one, %d.. of phase two), full "
                      "dimension tables; median of 3 runs of the reference's POLAR pipeline each (Pipeline::Schedule->"
                      "Finalize incl. table scan + count sink) at every thread count of %s; value = 1 / (2/3 t1 + 1/3 t2), "
                      "the phases' shares of the table, at the best thread count (`cores`) among those whose COUNT(*) "
                      "equals the single-threaded answer in BOTH phases (the reference's multi-threaded POLAR pipeline "
                      "has been seen to lose rows: such a run is listed, never used)"
                      % (n, starts[0], starts[1], thread_ladder()),
            "thread_sweep_tuples_per_s": sweep,
            "sample_count_star": counts, **detail}


class Case:
    """one measured workload: tables, join orders, pipeline, executors"""
    pass


def run_case(name, scale, args, env, steps, warmup, with_cpu):
    """measure one workload; returns the record (dict).  env: torch, dist, dev, ctx, world, rank"""
    torch, dist, dev, ctx, world, rank = env["torch"], env["dist"], env["dev"], env["ctx"], env["world"], env["rank"]
    from polr_amd import capi, workloads, ssb_skew
    from polr_amd import dist as pdist
    from polr_amd import host as phost
    V = args.chunk_size
    is_ssb = name in SSB_QUERIES
    enumerator = args.enumerator if args.enumerator != "auto" else ("sample" if is_ssb else "each_last_once")
    args.enumerator_name = enumerator
    t_gen0 = time.time()
    inst = None
    if is_ssb:
        query = SSB_QUERIES[name]
        # N > 1 (BASELINE configs[4]): ONE table, contiguous lo_orderkey partitions, one per rank; the
        # dimension tables are built on rank 0 and broadcast.  Default = weak scaling: the table is --scale x N (every rank
        # probes 6 M x --scale rows; N = 8 at the default scale is SF800, the shape of configs[4]); --strong: the table is
        # --scale, the ranks split it; --own-tables: every rank a whole lineorder of --scale of its own (same dimension
        # tables, same order keys / skew phases / load.sql rules, its own per-row draws: row_salt)
        one_table = world > 1 and not args.own_tables
        total_scale = scale * world if (one_table and not args.strong) else scale
        z = ssb_skew.sizes(total_scale)
        n_total = z["n_lo"]
        wl0 = ssb_skew.workload(query, sf=total_scale, n_lo=n_total, host_probe=False)
        inst = wl0["instance"]
        lo, hi = pdist.probe_partition(n_total, world, rank, V) if one_table else (0, n_total)
        names = list(ssb_skew.PROBE_COLS)
        cols_t = inst.lineorder_torch(lo, hi, dev, cols=names,
                                      row_salt=0 if (one_table or world == 1) else rank * (n_total + (-n_total) % 4))
        tens = [cols_t[c] for c in names]
        signed = [False] * len(names)
        n_rows = hi - lo
        sel = None
        flt = None
        dim_rows = {"customer": len(inst.c_custkey), "supplier": inst.n_s, "part": inst.n_p, "date": 2556}
        # what SelSampleEnumeration reads off the plan: base-table cardinality, pushed-down predicate, PRIMARY KEY
        node_info = [(n_total, False, False)] + [(dim_rows[j["name"]], j["name"] in ssb_skew.QUERY_WHERE[query], True)
                                                 for j in wl0["joins"]]
        cond_left = [[j["key_src"][0][1]] for j in wl0["joins"]]
        n_build_cols = [0] * len(wl0["joins"])
        est = [len(j["keys"][0]) for j in wl0["joins"]]
    else:
        builder = {"job_light_01": workloads.job_light_01, "job_q18": workloads.job_q18}.get(name)
        if builder is None:
            raise SystemExit("unknown workload %s" % name)
        wl0 = builder(scale=scale, seed=workloads.SEED)
        wl = wl0 if rank == 0 else builder(scale=scale, seed=pdist.probe_partition_seed(workloads.SEED, rank))
        probe = wl["probe"]
        names = list(probe["cols"].keys())
        n_rows = len(probe["cols"][names[0]])
        tens = [torch.from_numpy(np.ascontiguousarray(probe["cols"][n])).to(dev) for n in names]
        signed = [probe["cols"][n].dtype.kind == "i" for n in names]
        sel = probe.get("filter_sel")
        flt = probe.get("filter")
        node_info = None
        cond_left = wl0.get("cond_left_index") or [[j["key_src"][0][1]] for j in wl0["joins"]]
        n_build_cols = [len(j["payload"]) for j in wl0["joins"]]
        est = [len(j["keys"][0]) for j in wl0["joins"]]
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen0
    k = len(wl0["joins"])

    # ---- the bank of join orders: the host mirror of POLARConfig::GenerateJoinOrders with the session's enumerator
    t_enum0 = time.perf_counter()
    gen = phost.generate_join_orders(enumerator, len(names), n_build_cols, cond_left, est,
                                     max_join_orders=max(1, args.max_join_orders), routing=args.routing,
                                     node_info=node_info, return_routing=True)
    if gen is None:
        raise SystemExit("POLAR does not apply to this pipeline (fewer than two join orders)")
    enumeration_ms = (time.perf_counter() - t_enum0) * 1e3
    paths, routing = gen[0], gen[3]
    if args.pin_path is not None:
        # measurement aid: put join order P of the bank first, so that `--routing default_path` runs the whole table
        # down that one order (the kernel's rate on it, without routing)
        order = [args.pin_path] + [i for i in range(len(paths)) if i != args.pin_path]
        paths = paths[order]

    # ---- build sides: rank 0 builds in HBM, everyone else receives them over RCCL (one broadcast per buffer)
    joins_info = [{"key_bytes": sum(a.dtype.itemsize for a in j["keys"]), "n_rows": len(j["keys"][0]), "perfect": False}
                  for j in wl0["joins"]]
    t_build0 = time.time()
    joins = []
    if rank == 0:
        joins = capi.build_joins(ctx, wl0, auto=not args.reference_tables)
    bcast_bytes = 0
    if world > 1:
        # the path's one exchange step, in the product: polr_bcast_build (librccl, ncclBroadcast over xGMI) -- rank 0's
        # finalized tables to every rank.  torch.distributed only carries the 128-byte communicator id.
        # (POLR_DIST_BACKEND=gloo rehearsals on one GPU cannot form an RCCL communicator: they build locally)
        share_dev = bool(os.environ.get("POLR_SHARE_DEVICE")) and not os.environ.get("POLR_FORCE_COMM")
        if env.get("comm") is None and not share_dev:
            # (POLR_SHARE_DEVICE rehearsals put several ranks on one GPU, which RCCL refuses: they -- and only they --
            # build the deterministic tables on every rank)
            idt = torch.zeros(capi.COMM_ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                idt.copy_(torch.tensor(list(capi.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            ok = torch.ones(1, dtype=torch.int32)
            # ncclCommInitRank blocks until every rank has joined: on a thread of its own with a deadline.  A missed
            # deadline or a failure is FATAL for the multi-GPU run: every rank learns it through the all-reduce below,
            # prints the reason and exits non-zero (os._exit: a thread may still be blocked inside RCCL)
            import threading
            box = {}

            def _make():
                try:
                    box["comm"] = capi.Comm(ctx, bytes(idt.numpy().tobytes()), world, rank)
                except capi.PolrError as e:
                    box["err"] = str(e)

            th = threading.Thread(target=_make, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("POLR_COMM_TIMEOUT_S", "120")))
            if th.is_alive():
                box["err"] = "ncclCommInitRank did not return within the deadline"
            if "err" in box or "comm" not in box:
                ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0:
                print("bench.py rank %d: no RCCL communicator for polr_bcast_build (%s); a multi-GPU run without the "
                      "build broadcast is not measured" % (rank, box.get("err", "another rank failed")),
                      file=sys.stderr, flush=True)
                os._exit(3)
            env["comm"] = box["comm"]
        comm = env.get("comm")
        if comm is not None:
            before = comm.bytes_broadcast()
            got = []
            for x in range(k):
                ht = comm.bcast_build(joins[x][0] if rank == 0 else None, root=0)
                pnames = list(wl0["joins"][x]["payload"].keys())
                ht.preds = [(op, src, pnames.index(col)) for op, src, col in wl0["joins"][x].get("preds", [])]
                got.append((ht, wl0["joins"][x]["key_src"]))
            joins = got
            bcast_bytes = comm.bytes_broadcast() - before
            if bcast_bytes <= 0:
                raise SystemExit("polr_bcast_build moved no bytes")
        elif rank != 0:
            joins = capi.build_joins(ctx, wl0, auto=not args.reference_tables)
        torch.cuda.synchronize()
    t_build = time.time() - t_build0
    for x in range(k):
        joins_info[x]["perfect"] = joins[x][0].info()["kind"] == 1

    cols = [capi.dev_col(t.data_ptr(), t.element_size(), signed=s) for t, s in zip(tens, signed)]
    pipe = capi.Pipeline(ctx, cols, n_rows, joins, paths)
    scan_info = None
    lip_mask = 0
    if args.enable_lip:
        for x, j in enumerate(wl0["joins"]):
            if len(j["key_src"]) == 1 and j["key_src"][0][0] < 0:
                lip_mask |= 1 << x
    device_scan = (bool(flt) or lip_mask != 0) and not args.host_filter
    sel_t = None
    if device_scan:
        # source side on the device: the pushed-down filter of the table scan thins the
        # 1024-row vectors into the chunks the multiplexer sees; selection and chunk boundaries stay in HBM
        f3 = [(names.index(c), op, const) for c, op, const in (flt or [])]
        best = None
        for _ in range(5):
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            n_tuples, n_chunks = pipe.scan_filter(f3, vector_size=V, lip_joins=lip_mask)
            dt_s = time.perf_counter() - t_s
            best = dt_s if best is None else min(best, dt_s)
        scan_bytes = 2 * sum(tens[names.index(c)].element_size() for c, _, _ in (flt or [])) * n_rows + 4 * n_tuples
        # LIP: the probe key column of every pre-filtered join is read once per pass of the scan (count + write)
        scan_bytes += 2 * n_rows * sum(tens[j["key_src"][0][1]].element_size() for x, j in enumerate(wl0["joins"])
                                       if (lip_mask >> x) & 1)
        scan_info = {"rows": n_rows, "selected": int(n_tuples), "chunks": int(n_chunks),
                     "ms": round(best * 1e3, 4), "algorithmic_bytes": int(scan_bytes),
                     "GB/s": round(scan_bytes / best / 1e9, 1),
                     "note": "whole call (five kernels, one synchronisation); outside the timed region"}
        if sel is not None and not lip_mask and n_tuples != len(sel):
            raise SystemExit("device scan selected %d rows, the workload's host filter %d" % (n_tuples, len(sel)))
        offs = None
    else:
        if sel is not None:
            sel_t = torch.from_numpy(np.ascontiguousarray(sel)).to(dev)
            pipe.set_selection(sel_t.data_ptr(), device=True, n=len(sel))
        n_tuples = len(sel) if sel is not None else n_rows
        offs = chunk_offsets_for(sel, n_rows, V)
        n_chunks = len(offs) - 1 if offs is not None else (n_tuples + V - 1) // V
    budget = args.regret_budget
    if routing == "exponential_backoff":
        budget = n_rows / 10240.0 / 10 / 1  # polar_config.cpp:115-120
    # executors: 384 for table-sized sources (SF100 Q4.1: 256 1.50-1.52 ms, 384 1.43-1.45, 448 1.38, 512 1.45; Q4.2 /
    # Q4.3: 256 = 384, 512 slower); 32 for small ones -- except the strategies that decide every chunk
    # (one dependent routing round trip per chunk and executor: they want as many executors as the grid carries)
    per_chunk = routing in ("opportunistic", "dynamic")
    want_e = args.executors if args.executors > 0 else (384 if n_chunks > 65536 else (640 if per_chunk else 32))
    # a per-chunk strategy's rounds are a few hundred tuples each: half the grid probes them just as fast, and every
    # routing round trip (counter exchange, ticket, arrival) is quicker with half the idle waves polling (measured on
    # the JOB-light shape: OPPORTUNISTIC 4.9 -> 6.2 G tuples/s)
    pool_share = 2 if per_chunk and n_chunks <= 65536 else 1
    E = max(1, min(want_e, n_chunks))
    P = max(1, args.streams) if not args.sync_every_step else 1
    sets = []
    for _p in range(P):
        execs = []
        for e in range(E):
            m = capi.DeviceMultiplexer(pipe, routing, chunk_size=V, regret_budget=budget,
                                       init_tuple_count=args.init_tuple_count, atc_multiplier=args.atc_multiplier,
                                       log_rounds=False)
            if device_scan:
                m.use_scan_chunks()
            elif offs is not None:
                m.set_chunk_offsets(offs)
            execs.append((m, (e * n_chunks) // E, ((e + 1) * n_chunks) // E))
        sets.append(execs)
    raw_results = [None]
    ranges = [(x[1], x[2]) for x in sets[0]]
    # --ranges-per-executor R: executor e owns R ranges, one out of every R-th part of the source (part r: chunks
    # [r n / R, (r + 1) n / R), cut E ways) -- every executor sees every stretch of a skewed table
    range_lists = None
    R = max(1, args.ranges_per_executor)
    if R > 1 and not args.morsels:
        range_lists = []
        for e in range(E):
            lst = []
            for r in range(R):
                lo_r, hi_r = (r * n_chunks) // R, ((r + 1) * n_chunks) // R
                lst.append((lo_r + (e * (hi_r - lo_r)) // E, lo_r + ((e + 1) * (hi_r - lo_r)) // E))
            range_lists.append(lst)
    all_mpxs = [x[0] for ex in sets for x in ex]
    step_no = [0]
    pipelined = not args.sync_every_step

    def step(fetch=True):
        # fresh multiplexer states, the whole pass and the closing FinalizePathRun: one launch
        cur = [x[0] for x in sets[step_no[0] % P]]
        step_no[0] += 1
        if args.morsels > 0:
            capi.run_resident_morsels(cur, 0, n_chunks, args.morsels, reset=True, finish=True, share=P)
        else:
            if range_lists is not None:
                capi.run_resident_ranges(cur, range_lists, reset=True, finish=True, share=max(P, pool_share))
            else:
                capi.run_resident(cur, ranges, reset=True, finish=True, share=max(P, pool_share))
        if fetch:
            for ex in sets:  # (settles every stream; the statistics reported are the last pass's)
                ms_ = [x[0] for x in ex]
                got = capi.finish_many_raw(ms_)  # the C structs; read out as dictionaries once the clock has stopped
                if ex[0][0] is cur[0]:
                    raw_results[0] = (got, ms_)
        return raw_results

    def merged(res):
        out = {"num_intermediates": 0, "num_rounds": 0, "input_tuple_count_per_path": [0] * len(paths),
               "stage_out": [[0] * k for _ in range(len(paths))]}
        for r in res:
            out["num_intermediates"] += r["num_intermediates"]
            out["num_rounds"] += r["num_rounds"]
            for p in range(len(paths)):
                out["input_tuple_count_per_path"][p] += r["input_tuple_count_per_path"][p]
                for j in range(k):
                    out["stage_out"][p][j] += r["stage_out"][p][j]
        return out

    for _ in range(warmup):
        step()
    # ---- timed region: K steps, barrier + synchronize on both sides, no profiling hooks -----------
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(fetch=not pipelined or i == steps - 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0  # this rank's K steps; the job's time is the MAX over ranks (below)
    if world > 1:
        dist.barrier()
    results = capi.stats_dicts(*raw_results[0])
    st = merged(results)
    # ---- same K steps again with a HIP event pair around every pool-kernel launch (on the launch stream) to get the
    # dominant kernel's device time for the roofline; kept out of the region `value` is computed from
    kernel_ms, launches, dt_events = 0.0, 0, None
    if not args.no_kernel_events:
        for m in all_mpxs:
            m.kernel_time()
            m.enable_timing(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            step(fetch=not pipelined or i == steps - 1)
        torch.cuda.synchronize()
        dt_events = time.perf_counter() - t1
        for m in all_mpxs:
            ms_e, n_e = m.kernel_time()
            kernel_ms += ms_e
            launches += n_e
            m.enable_timing(False)

    # ---- diagnostic only (POLR_DIAG_TIMELINE=<file.npz>, needs `make -C duckdb-polr_amd diag`): one more pass with the
    # instrumented pool kernels; per probe wave and unit {began waiting, got it, finished, exec/path/count}
    if os.environ.get("POLR_DIAG_TIMELINE"):
        import ctypes as C
        cap_tl = 4096
        n_waves = 256 * 16
        tl = torch.zeros(n_waves * cap_tl * 4, dtype=torch.int64, device=dev)
        kk = 2 if k <= 2 else 4 if k <= 4 else 6 if k <= 6 else 8
        # (the flat pipeline's kernels are built per stage count; the generic pipeline's kernel is one, in two builds)
        sfx = ("k%d" % kk) if pipe.launch_info(False).get("flat") else "g"
        fn = getattr(ctx.L, "polr_diag_timeline_set_%s" % sfx)
        fn.argtypes = [C.c_void_p, C.c_uint32]
        if fn(tl.data_ptr(), cap_tl) != 0:
            raise SystemExit("polr_diag_timeline_set failed")
        ph = getattr(ctx.L, "polr_diag_router_%s" % sfx)
        ph.argtypes = [C.c_void_p, C.c_int]
        pbuf = (C.c_uint64 * 16)()
        ph(pbuf, 1)
        step()
        torch.cuda.synchronize()
        fn(None, 0)
        ph(pbuf, 0)
        if pbuf[5]:
            n_s, n_r = float(pbuf[5]), float(max(pbuf[7], 1))
            print("routers: %d steps; mean us per step: wait %.2f absorb %.2f route %.2f publish %.2f rehearse %.2f; "
                  "entry -> state in LDS %.2f -> initialised %.2f -> first step %.2f -> first publish %.2f us" % (
                      pbuf[5], pbuf[0] / n_s / 100, pbuf[1] / n_s / 100, pbuf[2] / n_s / 100, pbuf[3] / n_s / 100,
                      pbuf[4] / n_s / 100, pbuf[8] / n_r / 100, pbuf[9] / n_r / 100, pbuf[10] / n_r / 100, pbuf[6] / n_r / 100),
                  file=sys.stderr)
        arr = tl.cpu().numpy().reshape(n_waves, cap_tl, 4)
        used = (arr[:, :, 2] != 0).sum(axis=1)
        keep = int(used.max()) if used.size else 0
        np.savez_compressed(os.environ["POLR_DIAG_TIMELINE"], tl=arr[:, :keep, :], used=used)

    # ---- the reference's harness artefacts (benchmark_runner --log_tuples_routed --measure_pipeline --dir_prefix
    # --nruns, benchmark/benchmark_runner.cpp:215-355): tmp/<prefix><ts>.csv, -intms.txt, -enumeration.csv per executor
    # (the reference writes one set per worker thread), tmp/<prefix><ts>-<hash>.csv per run; outside the timed region
    artefacts = None
    if args.log_tuples_routed or args.measure_pipeline:
        from polr_amd import harness
        P_ = len(paths)
        cap = (n_chunks // E + 2) * max(P_, 4) + 1024
        logm = []
        for e in range(E):
            m = capi.DeviceMultiplexer(pipe, routing, chunk_size=V, regret_budget=budget,
                                       init_tuple_count=args.init_tuple_count, atc_multiplier=args.atc_multiplier,
                                       log_rounds=True, max_log_rounds=cap)
            if device_scan:
                m.use_scan_chunks()
            elif offs is not None:
                m.set_chunk_offsets(offs)
            logm.append(m)
        files, mats, durations = [], [], []
        prefix = args.dir_prefix + ("r%d-" % rank if world > 1 else "")
        for _run in range(max(1, args.nruns)):
            torch.cuda.synchronize()
            t_r = time.perf_counter()
            capi.run_resident(logm, ranges, reset=True, finish=True)
            st_r = capi.finish_many(logm)
            ms_r = (time.perf_counter() - t_r) * 1e3
            for e, m in enumerate(logm):
                _lp, _lt, inter = m.fetch_log()
                f = harness.write_artefacts(os.getcwd(), prefix, routing, P_, inter, st_r[e]["num_intermediates"], ms_r,
                                            enumeration_ms, k, "%s rows %d" % (name, n_rows),
                                            log_tuples_routed=args.log_tuples_routed,
                                            measure_pipeline=args.measure_pipeline and e == 0)
                files.append(f)
                if routing == "alternate" and args.log_tuples_routed:
                    mats.append(np.asarray(inter, dtype=np.int64).reshape(-1, P_))
            durations.append(ms_r)
        artefacts = {"directory": os.path.join(os.getcwd(), "tmp"), "files": len(files) and sum(len(f) for f in files),
                     "runs": max(1, args.nruns), "median_pipeline_ms": round(float(np.median(durations)), 4)}
        if mats:
            artefacts["aggregates"] = {k_: int(sum(v)) for k_, v in harness.aggregates(mats).items()}
        for m in logm:
            m.close()
    value, dt_max, total_tuples = pdist.whole_job_throughput(dist, torch, dev, world, n_tuples, dt, steps)
    rec = None
    if rank == 0:
        alg = algorithmic_bytes(joins_info, paths.tolist(), st["input_tuple_count_per_path"], st["stage_out"])
        info = pipe.launch_info(False)
        roof = None
        if launches:
            sec = kernel_ms / 1e3 / steps
            achieved = alg / sec / 1e9
            key_bytes = 0
            for p_, path_ in enumerate(paths.tolist()):
                inp_ = st["input_tuple_count_per_path"][p_]
                for pos_, j_ in enumerate(path_):
                    key_bytes += inp_ * joins_info[j_]["key_bytes"]
                    inp_ = st["stage_out"][p_][pos_]
            working_set = sum(t.numel() * t.element_size() for t in tens) + \
                sum(int(j[0].info()["device_bytes"]) for j in joins)
            roof = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": None,
                    "traffic_note": "not measured inside bench.py (PMC passes are separate rocprofv3 runs: profiles/) and "
                                    "no committed PMC pass matches this command",
                    "kernel": "polr_pool_flat_kernel" if info.get("flat") else "polr_pool_gen_kernel",
                    "algorithmic_bytes_per_step": round(alg),
                    "algorithmic_bytes_per_tuple": round(alg / max(n_tuples, 1), 3),
                    "kernel_ms_per_step": round(kernel_ms / steps, 4),
                    "launches_per_step": launches / steps,
                    "ms_per_step_with_events": round(dt_events / steps * 1e3, 4),
                    "working_set_bytes": int(working_set),
                    # the bytes this design cannot avoid moving: one key per tuple ENTERING each stage (the SURVEY formula
                    # also credits a bitmap byte per lookup and 8 B of selection writes per match, which live in LDS here)
                    "key_bytes_per_step": int(key_bytes),
                    "frac_key_bytes": round(key_bytes / sec / 1e9 / HBM_PEAK_GBS, 5),
                    "note": "one launch = the whole pass: every executor's routing loop and all probe rounds (the kernel "
                            "time includes the device-side waits between dependent routing rounds)"}
            sig = {"workload": name, "scale": float(scale), "routing": routing, "join_enumerator": enumerator,
                   "max_join_orders": int(args.max_join_orders), "executors_per_gpu": int(E), "n_gpus": int(world)}
            roof["pmc_signature"] = sig
            pmc, pmc_file = find_pmc_summary(sig)
            if pmc and not args.morsels:  # (the PMC passes count the pool kernel's dispatches only: a device scan before it does not matter)
                # REPLAYED, not measured in this run: HBM bytes per launch of the same command's separate rocprofv3
                # --pmc passes (FETCH_SIZE, WRITE_SIZE), FETCH_SIZE corrected as calibrated for this kernel's access
                # patterns on gfx950 (profiles/r02_fetch_size_calibration.json)
                roof["traffic"] = int(pmc["traffic_bytes_per_launch_corrected"])
                roof["traffic_source"] = "replayed from profiles/" + pmc_file
                roof["traffic_note"] = pmc["traffic_note"]
        cpu = None
        if with_cpu:
            try:
                if is_ssb:
                    cpu = cpu_baseline_ssb(inst, SSB_QUERIES[name], routing, args, args.cpu_sample_rows, dev=dev)
                else:
                    cpu = cpu_baseline_generic(wl0, routing, n_tuples, enumerator)
                if cpu:
                    cpu["value"] = round(cpu["value"], 1)
            except Exception as e:  # the baseline must never take the measurement down
                cpu = {"value": None, "error": str(e)[-600:]}
        if is_ssb:
            desc = ("%s: SSB-skew %s at SF%g -- lineorder %d rows per GPU (rows %d..%d of %d, contiguous lo_orderkey "
                    "range) x %s; skew = benchmark/ssb-skew/init/load.sql:80-253 applied to synthetic SSB base tables, "
                    "generated on the device" % (name, SSB_QUERIES[name], total_scale, n_rows, lo, hi, n_total,
                                                 " x ".join("%s %d" % (j["name"], len(j["keys"][0])) for j in wl0["joins"])))
        else:
            desc = "%s (IMDB cardinalities x%.3g: %d probe tuples after the pushed-down filter; builds %s)" % (
                name, scale, n_tuples, ", ".join("%s %d" % (j["name"], len(j["keys"][0])) for j in wl0["joins"]))
        rec = {
            "metric": "probe-tuples/s", "value": round(value, 1), "unit": "tuples/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt_max / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if (args.strong and world > 1) else "weak", "vs_baseline": None, "dtype": "u64",
            "dtype_note": "32-bit keys; hash / range arithmetic in 32/64-bit integers; f64 only in the reward",
            "data": "synthetic",
            "config": {"workload": desc, "routing": routing, "join_enumerator": enumerator,
                       "max_join_orders": args.max_join_orders, "join_orders": paths.tolist(), "chunk_size": V,
                       "regret_budget": args.regret_budget, "init_tuple_count": args.init_tuple_count,
                       "sink": "count(*)", "probe_partition_per_gpu": int(n_tuples), "partitioning": (("one table of SF%g, contiguous lo_orderkey partitions, one per rank (%s)" % (
                           total_scale, "strong: the table does not grow with N" if args.strong else
                           "weak: the table is --scale x N")) if one_table else (
                           "every rank probes a lineorder of its own of this shape (same dimension tables, same skew "
                           "phases, per-row draws salted by the rank)" if world > 1 else "single rank: the whole table"))
                       if is_ssb else "per-rank seed",
                       "executors_per_gpu": E, "device_share_of_the_launch": "1/%d" % max(P, pool_share),
                       "lip_joins": [wl0["joins"][x]["name"] for x in range(k) if (lip_mask >> x) & 1],
                       "launch": "pool (one launch per pass: %d router waves + shared probe waves)" % E,
                       "chunks_per_executor": ("morsels of %d" % args.morsels) if args.morsels > 0 else "fixed ranges",
                       "build_tables": ["perfect" if ji["perfect"] else "hash" for ji in joins_info],
                       "passes_in_flight": ("%d streams, 1/%d of the device each" % (P, P)) if P > 1 else
                       ("back to back on one stream" if pipelined else "synchronised per pass")},
            "total_intermediates": int(st["num_intermediates"]),
            "routing_rounds": int(st["num_rounds"]),
            "tuples_per_path": st["input_tuple_count_per_path"],
            "generate_s": round(t_gen, 3), "build_s": round(t_build, 3), "build_broadcast_bytes": int(bcast_bytes),
            "build_broadcast": ("polr_bcast_build over RCCL" if bcast_bytes else ("built on every rank: POLR_SHARE_DEVICE rehearsal, ranks share one device" if world > 1 else "single rank")),
            "ranks_started": int(os.environ.get("POLR_RANKS_STARTED", "0")) or None,
            "roofline": roof, "cpu_baseline": cpu, "scan_filter": scan_info, "launch_info": info,
            "artefacts": artefacts,
            "timed_region": {"gpu": "routing + probing of every source chunk; probe key columns read from HBM inside the "
                                    "pass" + ("; the pushed-down filter (scan_filter.ms) runs before the clock starts"
                                              if scan_info else "; the query has no probe-side filter"),
                             "cpu": "the reference's whole POLAR pipeline: table scan (+ pushed-down filter) + joins + "
                                    "count sink"},
        }
        if is_ssb:
            rec["config"]["ssb_skew_params"] = {k_: v for k_, v in wl0["params"].items() if k_ != "year_band_ends"}
        rec["count_star"] = int(sum(st["stage_out"][p][k - 1] for p in range(len(paths))))
        if scan_info:
            e2e_ms = dt_max / steps * 1e3 + scan_info["ms"]
            rec["value_with_scan"] = round(total_tuples / (e2e_ms / 1e3), 1)
        if cpu and cpu.get("value"):
            rec["gpu_over_cpu"] = round(value / cpu["value"], 2)
            if scan_info:
                rec["gpu_over_cpu_with_scan"] = round(rec["value_with_scan"] / cpu["value"], 2)
            c1 = cpu.get("count_star_threads_1")
            if c1 is not None:
                rec["count_star_matches_reference"] = bool(c1 == rec["count_star"])
            mt = [v for kk, v in cpu.items() if kk.startswith("count_star_threads_") and kk != "count_star_threads_1"]
            if mt and c1 is not None and any(v != c1 for v in mt):
                rec["reference_multithreaded_count_differs"] = True
        if is_ssb and cpu and cpu.get("sample_count_star") and world == 1 and not device_scan:
            # parity at full size: the device's COUNT(*) over the same contiguous samples the reference ran on
            dev_counts = []
            m = capi.DeviceMultiplexer(pipe, routing, chunk_size=V, regret_budget=budget,
                                       init_tuple_count=args.init_tuple_count, log_rounds=False)
            for s0 in cpu["sample_starts"]:
                capi.run_resident([m], [(s0 // V, (s0 + cpu["sample_rows_each"]) // V)], reset=True, finish=True)
                r = capi.finish_many([m])[0]
                dev_counts.append(int(sum(r["stage_out"][p][k - 1] for p in range(len(paths)))))
            m.close()
            rec["sample_count_star_device"] = dev_counts
            rec["count_star_matches_reference"] = bool(dev_counts == cpu["sample_count_star"])
    # release the device objects while the runtime (and a profiler attached to it) is still alive
    for m in all_mpxs:
        m.close()
    pipe.close()
    for ht, _ in joins:
        ht.close()
    del tens
    torch.cuda.empty_cache()
    return rec


def cpu_baseline_job_full(cases, args, budget_s=40.0, cap_tuples=6_000_000):
    """job_full: a BOUNDED sample of the family through the reference itself -- the pipelines of this rank in order, those
    with at most `cap_tuples` probe tuples, until about `budget_s` seconds of wall clock are spent (table load included) --
    each at one thread and at 16 (the best valid count on the other workloads); value = sum of their probe tuples / sum of
    their best valid pipeline times.  Without oracle/_ref: the oracle restatement on the first pipelines (kind "port")."""
    from oracle import ref_run
    if ref_run.available() and all("ref" in c["wl"] for c in cases[:1]):
        t_start = time.time()
        done, secs, used, threads_used, bad = 0, 0.0, [], {}, []
        ladder = sorted({1, min(16, os.cpu_count() or 1)})
        for c in cases:
            if time.time() - t_start > budget_s:
                break
            if c["n_tuples"] > cap_tuples or "ref" not in c["wl"]:
                continue
            ref = c["wl"]["ref"]
            settings = list(ref["settings"]) + ["SET multiplexer_routing TO '%s'" % args.routing,
                                                "SET join_enumerator TO 'each_last_once'"]
            try:
                runs = reference_runs(ref["tables"], ref.get("pk", {}), ref["query"], settings, c["n_tuples"], repeat=3,
                                      threads=ladder)
            except Exception as e:  # (one plan the reference cannot run does not take the sample down)
                bad.append("%s: %s" % (c["name"], str(e)[-120:]))
                continue
            best, _detail = pick_baseline(runs, c["n_tuples"])
            if not best:
                continue
            done += c["n_tuples"]
            secs += c["n_tuples"] / best[0]
            used.append(c["name"])
            threads_used[str(best[1])] = threads_used.get(str(best[1]), 0) + 1
        if done:
            return {"value": round(done / secs, 1), "unit": "probe-tuples/s", "cores": max(int(t) for t in threads_used),
                    "kind": "reference",
                    "sample": "%d of this rank's %d pipelines (in order, those of at most %d probe tuples, until %.0f s of wall "
                              "clock incl. loading their tables): %s -- the reference's POLAR pipeline on the same synthetic "
                              "tables, each at threads %s, median of 3 runs, the faster run with the single-threaded COUNT(*) "
                              "counts; value = their probe tuples / their pipeline times"
                              % (len(used), len(cases), cap_tuples, budget_s, ", ".join(used), ladder),
                    "best_thread_count_histogram": threads_used, "pipelines_the_reference_failed_on": bad or None}
    from oracle import polr_oracle as orc
    done, secs = 0, 0.0
    cap = 300_000
    for c in cases[:4]:
        wl = c["wl"]
        oj = [orc.JoinSpec(orc.HashTable(j["keys"], list(j["payload"].values())), j["key_src"]) for j in wl["joins"]]
        sel = wl["probe"].get("filter_sel")
        pc = list(wl["probe"]["cols"].values())
        if sel is not None:
            sel = sel[:cap]
            n_s = len(sel)
        else:
            pc = [a[:cap] for a in pc]
            n_s = len(pc[0])
        t_c = time.time()
        orc.run_pipeline(pc, oj, c["paths"], routing=args.routing, collect_output=False, sel=sel)
        secs += time.time() - t_c
        done += n_s
    return {"value": round(done / secs, 1), "unit": "probe-tuples/s", "cores": 1, "kind": "port",
            "sample": "the first %d probe tuples of the first %d pipelines of rank 0 through the oracle restatement, single "
                      "thread (oracle/_ref is not built here)" % (cap, min(4, len(cases)))}


def run_job_full(args, env, steps, warmup, with_cpu):
    """BASELINE.json configs[3]: the 113 JOB-shaped pipelines (polr_amd/job_family.py), whole queries dealt round-robin
    to the ranks, nothing exchanged.  One step = one pass of EVERY pipeline of the rank (one pool launch each,
    enqueued back to back)."""
    torch, dist, dev, ctx, world, rank = env["torch"], env["dist"], env["dev"], env["ctx"], env["world"], env["rank"]
    from polr_amd import capi, job_family as jf
    from polr_amd import dist as pdist
    from polr_amd import host as phost
    V = args.chunk_size
    scale = args.scale if args.scale is not None else 1.0
    t_gen0 = time.time()
    shapes = jf.shapes()
    names_all = sorted(shapes)
    mine = [names_all[i] for i in pdist.shard_queries(len(names_all), world, rank)]
    if args.job_limit > 0:
        mine = mine[:args.job_limit]
    tables = jf.Tables(scale=scale)
    dev_cols = {}

    def dev_col(table, cname, arr):
        key = (table, cname)
        if key not in dev_cols:
            dev_cols[key] = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        return dev_cols[key]

    # (executors per pipeline: 4 / 8 / 16 / 32 measured on the x0.2 instance: 61.6 / 49.8 / 44.7 / 43.7 ms of kernel time per
    # pass -- 32 costs more host time per launch than it saves)
    E = args.executors if args.executors > 0 else 16
    cases = []
    for name in mine:
        wl = jf.workload(name, tables, shapes[name])
        if wl is None:
            continue
        k = len(wl["joins"])
        pn = list(wl["probe"]["cols"].keys())
        gen = phost.generate_join_orders("each_last_once", len(pn), [len(j["payload"]) for j in wl["joins"]],
                                         wl["cond_left_index"], [len(j["keys"][0]) for j in wl["joins"]],
                                         max_join_orders=8, routing=args.routing)
        if gen is None:
            continue
        paths = gen[0]
        joins = capi.build_joins(ctx, wl, auto=not args.reference_tables)
        tens = [dev_col(wl["probe"]["name"], c, wl["probe"]["cols"][c]) for c in pn]
        n_rows = len(wl["probe"]["cols"][pn[0]])
        cols = [capi.dev_col(t.data_ptr(), t.element_size(), signed=True) for t in tens]
        pipe = capi.Pipeline(ctx, cols, n_rows, joins, paths)
        flt = wl["probe"].get("filter")
        if flt:
            n_tuples, n_chunks = pipe.scan_filter([(pn.index(c), op, const) for c, op, const in flt], vector_size=V)
        else:
            n_tuples, n_chunks = n_rows, (n_rows + V - 1) // V
        e_n = max(1, min(E, n_chunks))
        mpxs = []
        for e in range(e_n):
            m = capi.DeviceMultiplexer(pipe, args.routing, chunk_size=V, regret_budget=args.regret_budget,
                                       init_tuple_count=args.init_tuple_count, atc_multiplier=args.atc_multiplier,
                                       log_rounds=False)
            if flt:
                m.use_scan_chunks()
            mpxs.append(m)
        ranges = [((e * n_chunks) // e_n, ((e + 1) * n_chunks) // e_n) for e in range(e_n)]
        ji = [{"key_bytes": 4, "n_rows": len(j["keys"][0]), "perfect": joins[x][0].info()["kind"] == 1}
              for x, j in enumerate(wl["joins"])]
        cases.append({"name": name, "wl": wl, "pipe": pipe, "joins": joins, "mpxs": mpxs, "ranges": ranges,
                      "n_tuples": int(n_tuples), "paths": paths, "k": k, "ji": ji})
        if rank == 0 and len(cases) % 10 == 0:
            print("job_full: %d pipelines set up, %.1f s" % (len(cases), time.time() - t_gen0), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen0
    stats = {}

    def step(fetch):
        # (measured: running the pipelines side by side on shares of the device -- POLR_RUN_SHARE 4 / 8 / 16 -- is
        # 15-60 % slower than one after the other; a few fan-out heavy pipelines carry the pass, and they want the
        # whole device)
        # (all on the context's stream: every pipeline's leader has a stream of its own, and full-size launches on
        # different streams overlap at their edges -- the HIP events around a launch would then time the wait for the
        # device, too)
        for c in cases:
            capi.run_resident(c["mpxs"], c["ranges"], reset=True, finish=True,
                              stream=None if args.own_streams else ctx.stream())
        if fetch:
            for c in cases:
                try:
                    stats[c["name"]] = capi.finish_many(c["mpxs"])
                except capi.PolrError as e:
                    raise SystemExit("job_full: pipeline %s (%d joins: %s; %d probe tuples; join orders %s): %s" % (
                        c["name"], c["k"], [j["name"] for j in c["wl"]["joins"]], c["n_tuples"], c["paths"].tolist(), e))

    for _ in range(warmup):
        t_w = time.time()
        step(True)
        if rank == 0:
            print("job_full: warm-up pass %.3f s" % (time.time() - t_w), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i == steps - 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    kernel_ms, launches = 0.0, 0
    if not args.no_kernel_events:
        for c in cases:
            for m in c["mpxs"]:
                m.kernel_time()
                m.enable_timing(True)
        for i in range(steps):
            step(i == steps - 1)
        torch.cuda.synchronize()
        for c in cases:
            c["kernel_ms"] = 0.0
            for m in c["mpxs"]:
                ms_e, n_e = m.kernel_time()
                kernel_ms += ms_e
                launches += n_e
                c["kernel_ms"] += ms_e / steps
                m.enable_timing(False)
    my_tuples = sum(c["n_tuples"] for c in cases)
    value, dt_max, total_tuples = pdist.whole_job_throughput(dist, torch, dev, world, my_tuples, dt, steps)
    alg, inter, rounds, count = 0.0, 0, 0, 0
    for c in cases:
        P, k = len(c["paths"]), c["k"]
        tpp = [sum(st["input_tuple_count_per_path"][p] for st in stats[c["name"]]) for p in range(P)]
        so = [[sum(st["stage_out"][p][j] for st in stats[c["name"]]) for j in range(k)] for p in range(P)]
        alg += algorithmic_bytes(c["ji"], c["paths"].tolist(), tpp, so)
        inter += sum(st["num_intermediates"] for st in stats[c["name"]])
        rounds += sum(st["num_rounds"] for st in stats[c["name"]])
        count += sum(so[p][k - 1] for p in range(P))
    rec = None
    if rank == 0:
        roof = None
        if launches:
            sec = kernel_ms / 1e3 / steps
            roof = {"bound": "hbm", "achieved": round(alg / sec / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg / sec / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                    "kernel": "polr_pool_gen_kernel / polr_pool_flat_kernel (one launch per pipeline and pass)",
                    "algorithmic_bytes_per_step": round(alg), "kernel_ms_per_step": round(kernel_ms / steps, 4),
                    "launches_per_step": launches / steps,
                    "note": "rank 0's pipelines; algorithmic bytes per SURVEY.md 8(d) summed over them"}
            sig = {"workload": "job_full", "scale": float(scale), "routing": args.routing, "join_enumerator": "each_last_once",
                   "max_join_orders": 8, "executors_per_gpu": int(E), "n_gpus": int(world)}
            roof["pmc_signature"] = sig
            pmc, pmc_file = find_pmc_summary(sig)
            if pmc:
                # REPLAYED: the mean HBM bytes per pool-kernel dispatch of the same command's separate --pmc passes x the
                # launches of a pass
                roof["traffic"] = int(pmc["traffic_bytes_per_launch_corrected"] * (launches / steps))
                roof["traffic_source"] = "replayed from profiles/" + pmc_file + " (mean per dispatch x launches per pass)"
                roof["traffic_note"] = pmc["traffic_note"]
        cpu = None
        if with_cpu:
            try:
                cpu = cpu_baseline_job_full(cases, args)
            except Exception as e:
                cpu = {"value": None, "error": str(e)[-600:]}
        # which pipelines carry the pass: the ten longest, with what their joins produced (all join orders together)
        slowest = []
        for c in sorted(cases, key=lambda c_: -c_.get("kernel_ms", 0.0))[:10]:
            P, k = len(c["paths"]), c["k"]
            so = [sum(st["stage_out"][p][j] for st in stats[c["name"]] for p in range(P)) for j in range(k)]
            tpp = [sum(st["input_tuple_count_per_path"][p] for st in stats[c["name"]]) for p in range(P)]
            slowest.append({"query": c["name"], "kernel_ms": round(c.get("kernel_ms", 0.0), 3), "probe_tuples": c["n_tuples"],
                            "joins": [j["name"] for j in c["wl"]["joins"]],
                            "build_rows": [len(j["keys"][0]) for j in c["wl"]["joins"]],
                            "tables": ["perfect" if ji["perfect"] else "hash" for ji in c["ji"]],
                            "produced_by_position": so, "tuples_per_join_order": tpp,
                            "flat": int(c["pipe"].launch_info(False).get("flat", 0))})
        rec = {"metric": "probe-tuples/s", "value": round(value, 1), "unit": "tuples/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": round(dt_max / steps * 1e3, 4), "higher_is_better": True,
               "slowest_pipelines": slowest,
               "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "job_full: the 113 JOB-shaped pipelines (IMDB cardinalities x%.3g), %d of them on "
                                      "this rank, %d probe tuples per pass on this rank; whole queries round-robin over "
                                      "the GPUs, no exchange" % (scale, len(cases), my_tuples),
                          "routing": args.routing, "join_enumerator": "each_last_once", "max_join_orders": 8,
                          "executors_per_pipeline": E, "chunk_size": V, "sink": "count(*)",
                          "joins_per_pipeline": sorted(set(c["k"] for c in cases))},
               "total_intermediates": int(inter), "routing_rounds": int(rounds), "count_star_sum": int(count),
               "pipelines": len(cases), "generate_s": round(t_gen, 3), "roofline": roof, "cpu_baseline": cpu}
        if cpu and cpu.get("value"):
            rec["gpu_over_cpu"] = round(value / cpu["value"], 2)
    for c in cases:
        for m in c["mpxs"]:
            m.close()
        c["pipe"].close()
        for ht, _ in c["joins"]:
            ht.close()
    return rec


def spawn_ranks(n, argv=None, script=None, extra_env=None):
    """`python bench.py --gpus N` without a launcher: the parent -- which has made NO GPU call (torch.cuda.device_count()
    does not initialise the runtime) -- starts N fresh rank processes of this script, one per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rendezvous on 127.0.0.1), relays rank 0's JSON line and returns
    non-zero if any rank does.  Never re-executes a process that touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = script or os.path.abspath(__file__)
    argv = list(sys.argv[1:] if argv is None else argv)
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "POLR_RANKS_STARTED": str(n)})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    for p_ in procs[1:]:
        codes.append(p_.wait())
    # ONE line on stdout: rank 0's JSON object (whatever else a library wrote there goes to stderr)
    for line in out0.decode(errors="replace").splitlines():
        if line.startswith("{"):
            sys.stdout.write(line + "\n")
        elif line.strip():
            sys.stderr.write(line + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print("bench.py: rank(s) %s exited non-zero" % ", ".join("%d (code %d)" % rc for rc in bad), file=sys.stderr)
        return 1
    return 0


def run_q41_shipped(args, env, steps, warmup, with_cpu):
    """sub-record: SSB-skew Q4.1 AS THE REFERENCE SHIPS IT (benchmark/ssb-skew/queries/q4-1.sql) at SF100 -- the multiplexed
    star join with its own sink, `SUM(lo_revenue - lo_supplycost) GROUP BY d_year, c_nation`, all on the device: the flat
    pool launch emits the surviving tuples' row ids, the perfect-hash aggregate (polr_out_aggregate_grouped) folds them into
    the group cells; one step = reset of the output cursor + the whole adaptive pass + the aggregate; only 350 group
    cells leave the GPU."""
    torch, dev, ctx, rank = env["torch"], env["dev"], env["ctx"], env["rank"]
    from polr_amd import capi, ssb_skew
    from polr_amd import host as phost
    V = args.chunk_size
    scale = args.scale if args.scale is not None else 100.0
    query = "q4.1"
    z = ssb_skew.sizes(scale)
    n = z["n_lo"]
    wl0 = ssb_skew.workload(query, sf=scale, n_lo=n, host_probe=False)
    inst = wl0["instance"]
    names = list(ssb_skew.PROBE_COLS) + ["lo_revenue", "lo_supplycost"]
    cols_t = inst.lineorder_torch(0, n, dev, cols=names)
    tens = [cols_t[c] for c in names]
    k = len(wl0["joins"])
    dim_rows = {"customer": len(inst.c_custkey), "supplier": inst.n_s, "part": inst.n_p, "date": 2556}
    node_info = [(n, False, False)] + [(dim_rows[j["name"]], j["name"] in ssb_skew.QUERY_WHERE[query], True) for j in wl0["joins"]]
    gen = phost.generate_join_orders("sample", 4, [0] * k, [[j["key_src"][0][1]] for j in wl0["joins"]],
                                     [len(j["keys"][0]) for j in wl0["joins"]], max_join_orders=3,
                                     routing=args.routing, node_info=node_info, return_routing=True)
    paths, routing = gen[0], gen[3]
    P = len(paths)
    joins = capi.build_joins(ctx, wl0, auto=True)
    cols = [capi.dev_col(t.data_ptr(), t.element_size(), signed=(c in ("lo_revenue", "lo_supplycost")))
            for c, t in zip(names, tens)]
    pipe = capi.Pipeline(ctx, cols, n, joins, paths)
    info = pipe.launch_info(True)
    n_chunks = (n + V - 1) // V
    E = args.executors if args.executors > 0 else 384
    mpxs = [capi.DeviceMultiplexer(pipe, routing, chunk_size=V, regret_budget=args.regret_budget,
                                   init_tuple_count=args.init_tuple_count, log_rounds=False) for _ in range(E)]
    ranges = [((e * n_chunks) // E, ((e + 1) * n_chunks) // E) for e in range(E)]
    fused = not args.two_kernel_sink
    out = capi.Output(pipe, 1024, 64 if fused else 32768)
    cn = wl0["joins"][0]["payload"]["c_nation"]
    dy = wl0["joins"][3]["payload"]["d_year"]
    keys = [(3, 0, int(dy.min()), int(dy.max()) - int(dy.min()) + 1), (0, 0, int(cn.min()), int(cn.max()) - int(cn.min()) + 1)]
    specs = [("count_star", -1, 0), ("sum", -1, names.index("lo_revenue")), ("sum", -1, names.index("lo_supplycost"))]
    result = [None]

    ctx_stream = ctx.stream()  # the run goes on the context's stream: in line with the reset before, the aggregate behind

    if fused:
        out.fuse_grouped(keys, specs)  # the GROUP BY inside the run: the last join folds its survivors into the group cells

    def step():
        out.reset()
        capi.run_resident(mpxs, ranges, out=out, reset=True, finish=True, stream=ctx_stream)
        # (reads the group cells back: synchronises)
        result[0] = out.fused_result(stream=ctx_stream) if fused else out.aggregate_grouped(keys, specs)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = capi.finish_many(mpxs)
    vals, counts, dropped = result[0]
    groups = {}
    nv = keys[1][3]
    for g_, v in enumerate(vals):
        if v[0]:
            groups["%d/%d" % (keys[0][2] + g_ // nv, keys[1][2] + g_ % nv)] = int(v[1] - v[2])
    count_star = int(sum(v[0] for v in vals))
    for m in mpxs:
        m.kernel_time()
        m.enable_timing(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    kernel_ms = sum(m.kernel_time()[0] for m in mpxs) / steps
    inter = sum(st["num_intermediates"] for st in stats)
    cpu = None
    if with_cpu:
        # (the reference runs what the device ran: the SAMPLE enumerator -- run_case leaves the enumerator of ITS workload
        # in args, and each_last_once with max_join_orders below the join count ends the reference in "vector::reserve")
        args.enumerator_name = "sample"
        try:
            cpu = cpu_baseline_ssb(inst, query, routing, args, min(args.cpu_sample_rows, args.shipped_sample_rows), dev=dev,
                                   shipped=True)
            if cpu:
                cpu["value"] = round(cpu["value"], 1)
        except Exception as e:
            # (should the reference fail on the shipped select list: the COUNT(*) form of the same pipeline on the same
            # samples is timed instead and labelled as such)
            err = str(e)[-400:]
            try:
                cpu = cpu_baseline_ssb(inst, query, routing, args, min(args.cpu_sample_rows, args.shipped_sample_rows), dev=dev)
                cpu["value"] = round(cpu["value"], 1)
                cpu["sink"] = "COUNT(*) -- the reference failed on the shipped select list on this host: " + err
            except Exception as e2:
                cpu = {"value": None, "error": err + " / " + str(e2)[-300:]}
    rec = {"metric": "probe-tuples/s", "value": round(n * steps / dt, 1), "unit": "tuples/s", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "ssb_skew_q41 as shipped: SSB-skew q4.1 at SF%g, lineorder %d rows x customer / supplier / "
                                  "part / date, sink SUM(lo_revenue - lo_supplycost) GROUP BY d_year, c_nation "
                                  "(benchmark/ssb-skew/queries/q4-1.sql; c_nation as its code)" % (scale, n),
                      "routing": routing, "join_enumerator": "sample", "max_join_orders": 3, "join_orders": paths.tolist(),
                      "executors_per_gpu": E,
                      "sink": ("perfect-hash GROUP BY FUSED into the run (polr_out_fuse_grouped): the last join folds its "
                               "survivors into the group cells, no row id is written") if fused else
                              ("perfect-hash GROUP BY on the device (polr_out_aggregate_grouped) over the row ids the flat "
                               "pool launch emits"),
                      "launch": "flat pipeline, emitting" if info.get("flat") else "generic pipeline, emitting"},
           "count_star": count_star, "groups": len(groups), "rows_dropped_by_the_group_domain": int(dropped),
           "profit_checksum": int(sum(groups.values())), "total_intermediates": int(inter),
           "pool_kernel_ms_per_step": round(kernel_ms, 4),
           "timed_region": {"gpu": ("reset of the group cells + routing and probing of every source chunk with the aggregate "
                                    "inside the launch; the group cells are read back every step") if fused else
                                   ("output cursor reset + routing and probing of every source chunk + row-id emission of the "
                                    "join result + grouped aggregate; the group cells are read back every step"),
                            "cpu": "the reference's whole pipeline on the same SQL: scan + joins + its perfect-hash aggregate"},
           "cpu_baseline": cpu, "launch_info": info}
    if cpu and cpu.get("value"):
        rec["gpu_over_cpu"] = round(rec["value"] / cpu["value"], 2)
    for m in mpxs:
        m.close()
    out.close()
    pipe.close()
    for ht, _ in joins:
        ht.close()
    del tens, cols_t
    torch.cuda.empty_cache()
    return rec


def main():
    if os.environ.get("POLR_DIAG_TIMELINE"):
        from polr_amd import capi as _capi
        _capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libpolr_hip_diag.so")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ssb_skew_q41",
                    help="ssb_skew_q41 (default; also _q42 _q43 _q31 _q21), job_q18, job_light_01, job_full (the 113 "
                         "JOB-shaped pipelines, BASELINE configs[3])")
    ap.add_argument("--scale", type=float, default=None,
                    help="SSB-skew: scale factor per GPU (default 100; with --strong: of the whole job); JOB shapes: "
                         "fraction of the IMDB cardinalities (default 1)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the ranks split ONE table of --scale")
    ap.add_argument("--own-tables", action="store_true",
                    help="N > 1, SSB-skew: every rank probes a whole lineorder of --scale of its own (per-row draws salted "
                         "by the rank) instead of its partition of one table")
    ap.add_argument("--routing", default="adaptive_reinit")
    ap.add_argument("--enumerator", default="auto",
                    help="SET join_enumerator; auto = sample (the reference's default) for SSB-skew, each_last_once for "
                         "the JOB shapes (their plan statistics are not modelled)")
    ap.add_argument("--regret-budget", type=float, default=0.01)
    ap.add_argument("--init-tuple-count", type=int, default=1024)
    ap.add_argument("--atc-multiplier", type=int, default=1)
    ap.add_argument("--chunk-size", type=int, default=1024, help="STANDARD_VECTOR_SIZE of the host engine")
    ap.add_argument("--max-join-orders", type=int, default=3, help="SET max_join_orders (BASELINE configs[2]: 3)")
    ap.add_argument("--executors", type=int, default=0,
                    help="concurrent pipeline executors per GPU, each with its own multiplexer state and its own "
                         "contiguous share of the source chunks -- the counterpart of the reference's worker threads "
                         "(one PipelineExecutor + MultiplexerState per thread, pipeline.cpp:145-174).  0 (default) = 256 "
                         "for partitions of more than 65 536 chunks, else 32; 1 = the single-executor trace the parity "
                         "tests pin against the single-threaded reference")
    ap.add_argument("--reference-tables", action="store_true",
                    help="index the build sides exactly as the reference's planner would (perfect table only below its "
                         "1 M-value cap); default: polr_ht_finalize_auto")
    ap.add_argument("--host-filter", action="store_true")
    ap.add_argument("--morsels", type=int, default=0)
    ap.add_argument("--ranges-per-executor", type=int, default=1,
                    help="every executor owns this many chunk ranges, one out of each part of the source (1..8)")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--sync-every-step", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-sub-records", action="store_true")
    ap.add_argument("--shipped-sample-rows", type=int, default=16_000_000,
                    help="rows per sample of the CPU leg of the 'Q4.1 as shipped' sub-record")
    ap.add_argument("--two-kernel-sink", action="store_true",
                    help="'Q4.1 as shipped': row ids + polr_out_aggregate_grouped instead of the GROUP BY fused into the run")
    ap.add_argument("--own-streams", action="store_true",
                    help="job_full: every pipeline's launches on its leader's own stream (full-size launches side by side: a "
                         "stress for the pool protocol, not a measurement)")
    ap.add_argument("--job-limit", type=int, default=0, help="job_full: only the first N pipelines")
    ap.add_argument("--enable-lip", action="store_true",
                    help="PRAGMA enable_lip: the filters of the joins keyed by a source column thin the source chunks before "
                         "the multiplexer sees them (polr_pipeline_scan_filter_lip)")
    ap.add_argument("--log-tuples-routed", action="store_true",
                    help="PRAGMA enable_log_tuples_routed: write tmp/<prefix><ts>.csv, -intms.txt, -enumeration.csv per "
                         "executor (extra logged passes after the timed region)")
    ap.add_argument("--measure-pipeline", action="store_true",
                    help="PRAGMA enable_measure_pipeline: write tmp/<prefix><ts>-<hash>.csv with the pipeline duration in ms")
    ap.add_argument("--dir-prefix", default="", help="SET dir_prefix: prefix of the artefact file names")
    ap.add_argument("--nruns", type=int, default=1, help="benchmark_runner --nruns: logged / measured passes")
    ap.add_argument("--pin-path", type=int, default=None, help="measurement aid: make join order P of the bank path 0")
    ap.add_argument("--cpu-sample-rows", type=int, default=64_000_000,
                    help="SSB-skew CPU baseline: rows of each of the two contiguous lineorder samples")
    args = ap.parse_args()

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves (nothing has touched the GPU in this process)
        visible = torch.cuda.device_count()
        if visible < args.gpus and not os.environ.get("POLR_SHARE_DEVICE"):
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (POLR_SHARE_DEVICE=1 rehearses several ranks on "
                             "one device, without RCCL)" % (args.gpus, visible))
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # stdout carries ONE JSON line (rank 0's): libraries that chat on fd 1 (gloo prints its connection summary there)
        # are sent to stderr for the whole run; the line itself goes to the saved descriptor
        sys.stdout.flush()
        real_stdout = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        sys.stdout = real_stdout
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: RCCL's bootstrap over loopback (the container's hostname may not resolve)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        # control plane (rendezvous, the 128-byte communicator id, barriers, the max-over-ranks reduction): gloo.  The
        # data-path exchange -- the build sides -- goes over RCCL inside the library (polr_bcast_build, its own
        # communicator); torch's own RCCL (a second copy, on torch's bundled HIP runtime) is left out of the process
        # unless POLR_DIST_BACKEND=nccl asks for it
        backend = os.environ.get("POLR_DIST_BACKEND", "gloo")
        if os.environ.get("POLR_SHARE_DEVICE"):
            local_rank = 0
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d under a launcher with WORLD_SIZE %d: the two must agree" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from polr_amd import capi
    ctx = capi.Context(local_rank)  # raises without the HIP library / a gfx950 device: no fallback
    # tuning sweeps (tools/sweep_*.sh): the library itself reads nothing from the environment -- the knobs travel through
    # polr_ctx_set_pool_tuning
    knobs = {}
    for env_name, field in (("POLR_POOL_SHARE", "device_share"), ("POLR_POOL_UNITS_X", "units_x"),
                            ("POLR_POOL_HI_UNIT", "hi_unit"), ("POLR_POOL_HI_LOTTERY", "hi_lottery"),
                            ("POLR_POOL_HI_TUPLES", "hi_tuples"), ("POLR_POOL_IDLE_SLEEP", "idle_sleep"),
                            ("POLR_POOL_WATCHDOG_US", "watchdog_us"), ("POLR_POOL_SHARE_AFTER", "share_after")):
        if os.environ.get(env_name):
            knobs[field] = int(os.environ[env_name])
    if knobs:
        ctx.set_pool_tuning(**knobs)
    env = {"torch": torch, "dist": dist, "dev": dev, "ctx": ctx, "world": world, "rank": rank}
    if args.workload == "ssb_skew_q41_shipped":  # (the sub-record of the default line on its own)
        args.enumerator_name = "sample" if args.enumerator == "auto" else args.enumerator
        head = run_q41_shipped(args, env, min(args.steps, 10), min(args.warmup, 2), with_cpu=not args.no_cpu_baseline)
        print(json.dumps(head))
        ctx.close()
        return
    if args.workload == "job_full":
        head = run_job_full(args, env, args.steps, args.warmup, with_cpu=world == 1 and not args.no_cpu_baseline)
        if rank == 0:
            print(json.dumps(head))
        ctx.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    is_ssb = args.workload in SSB_QUERIES
    scale = args.scale if args.scale is not None else (100.0 if is_ssb else 1.0)
    head = run_case(args.workload, scale, args, env, args.steps, args.warmup, with_cpu=world == 1 and not args.no_cpu_baseline)
    subs = []
    if world == 1 and not args.no_sub_records:
        saved = (args.executors, args.max_join_orders)
        for sub, sub_scale in (("job_q18", 1.0), ("job_light_01", 1.0)):
            if sub == args.workload:
                continue
            args.max_join_orders = 8 if saved[1] == 3 else saved[1]  # (the JOB sub-records keep round 1's banks)
            try:
                r = run_case(sub, sub_scale, args, env, min(args.steps, 20), min(args.warmup, 3),
                             with_cpu=not args.no_cpu_baseline)
            except SystemExit as e:
                r = {"workload": sub, "error": str(e)}
            if r is not None:
                subs.append(r)
        args.executors, args.max_join_orders = saved
    if world == 1 and not args.no_sub_records and is_ssb and args.workload == "ssb_skew_q41":
        try:
            r = run_q41_shipped(args, env, min(args.steps, 10), min(args.warmup, 2), with_cpu=not args.no_cpu_baseline)
        except SystemExit as e:
            r = {"workload": "ssb_skew_q41 as shipped", "error": str(e)}
        subs.append(r)
    if rank == 0:
        if subs:
            head["sub_records"] = subs
        print(json.dumps(head))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
