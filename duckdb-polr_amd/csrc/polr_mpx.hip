// duckdb-polr_amd/csrc/polr_mpx.hip -- device-resident multiplexer.
//
// The reference asks its multiplexer for a route once per 1024-tuple chunk on the host
// (POLARPipelineExecutor::Execute, src/parallel/polar_pipeline_executor.cpp:320-366).  Here the
// multiplexer state (PhysicalMultiplexer + RoutingStrategy, polr_routing.h) lives on the device and one
// *routing step* (polr_mpx_device.h) runs between two probe rounds:
//
//     absorb the k counters of the previous round (AddNumIntermediates), FinalizePathRun, Route the next slice,
//     fold the strategy's routing window (num_cache_flushing_skips whole chunks that bypass routing, :322-329)
//     into the same round, publish the round
//
// Two ways to drive it, same decisions, same results (this file is their host side):
//   * polr_mpx_run / _run_many: one launch of the path kernel per round; its last busy workgroup runs the step
//     for the next round; the host pumps launches, throttled by two pinned progress words;
//   * polr_mpx_run_resident: the whole run is one launch; per executor a router wave keeps the state in LDS
//     and probe workgroups wait for its rounds on the device.
// Output row sets and per-round intermediates equal the host classes' exactly: same code (polr_routing.h), same
// IEEE double arithmetic.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "polr_internal.h"
#include "polr_mpx_device.h"
#include "polr_pool_device.h"

struct polr_mpx {
	polr_pipeline *pipe = nullptr;
	polr_ctx *ctx = nullptr; // (kept so that destroying the object never has to go through the pipeline)
	polr_mpx_config cfg;
	DevMpx *dev = nullptr;
	DevRound *round_dev = nullptr;
	uint64_t *prefix_dev = nullptr;
	uint32_t *unit_size_dev = nullptr;
	uint32_t *ticket_dev = nullptr;
	polr_mpx_stats *stats_dev = nullptr;
	unsigned long long *stamps_dev = nullptr; // diagnostic builds only
	uint32_t iter = 0; // launches of the path kernel so far (descriptor slot = iter & 1)
	unsigned long long *counts_dev = nullptr;
	uint64_t *chunk_offsets_dev = nullptr;
	bool chunk_offsets_owned = true; // false: the pipeline's scan result ...
	uint64_t scan_generation = 0;    // ... of this scan
	uint32_t *log_path = nullptr;
	uint64_t *log_tuples = nullptr, *log_inter = nullptr;
	uint32_t *done_host = nullptr;        // pinned, mapped: [0] routing steps completed, [1] done
	volatile uint32_t *progress_dev = nullptr; // the device's view of done_host
	uint32_t steps_base = 0;
	bool pending_sync = false;
	hipStream_t own_stream = nullptr;  // used when the caller passes no stream
	hipStream_t last_stream = nullptr; // where the queued tail of the last run sits
	uint32_t unit_size = 256;
	uint32_t wide0_mask = 0;
	uint64_t n_chunks = 0;
	// resident launches
	ResidentSync *sync_dev = nullptr;  // arrival counters of this executor
	char *execs_dev = nullptr;         // run header + executor descriptors + morsel cursor (owned by the first multiplexer of a run)
	uint32_t execs_cap = 0;
	std::vector<char> execs_host;      // what execs_dev holds (a pass that repeats the last one re-sends nothing)
	PoolSync *pool_dev = nullptr;      // unit rings of the runs this multiplexer leads
	uint32_t *share_dev = nullptr;     // work sharing of the generic pipeline: one record per probe wave + its flag
	size_t share_bytes = 0;
	uint32_t pool_lo_cap = 0, pool_hi_cap = 0;
	bool pool_dirty = false;           // a run was given up: rings and tickets are re-initialised before the next one
	polr_mpx *leader = nullptr;        // the first multiplexer of the last pool run this one took part in (owns the rings)
	uint32_t res_epoch = 0;
	polr_mpx_stats *stats_host = nullptr; // pinned, mapped: closing statistics of a POLR_RUN_FINISH run
	polr_mpx_stats *stats_host_dev = nullptr;
	bool stats_in_host = false;
	// optional per-launch timing (measurement only)
	bool timing = false;
	std::vector<hipEvent_t> ev_start, ev_stop;
	size_t ev_used = 0;
	double timed_ms = 0;
	uint64_t timed_launches = 0;
};

// every multiplexer keeps its work on ONE stream at a time: the caller's, else the stream of its last run
// (a resident run puts all its executors on one stream), else its own
static hipStream_t pick_stream(polr_mpx *m, void *stream) {
	return stream ? (hipStream_t)stream : (m->last_stream ? m->last_stream : m->own_stream);
}

// move a multiplexer onto `st`: whatever it still has queued elsewhere must be finished first
static hipError_t adopt_stream(polr_mpx *m, hipStream_t st) {
	const hipStream_t prev = m->last_stream ? m->last_stream : m->own_stream;
	hipError_t e = hipSuccess;
	if (prev != st) {
		e = hipStreamSynchronize(prev);
		m->pending_sync = false;
	}
	m->last_stream = st;
	return e;
}

static void drain_events(polr_mpx *m) {
	for (size_t i = 0; i < m->ev_used; i++) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, m->ev_start[i], m->ev_stop[i]) == hipSuccess) {
			m->timed_ms += ms;
			m->timed_launches++;
		}
	}
	m->ev_used = 0;
}

__global__ void polr_mpx_init_kernel(DevMpx *m, polr_mpx_config cfg, uint32_t n_paths, uint64_t n_tuples,
                                     uint64_t n_chunks, uint32_t *log_path, uint64_t *log_tuples,
                                     uint64_t *log_inter, uint32_t wide0_mask, volatile uint32_t *progress,
                                     uint32_t steps_done_init) {
	m->wide0_mask = wide0_mask;
	m->cfg = cfg;
	m->n_paths = n_paths;
	m->pad2 = 0;
	m->progress = progress;
	m->steps_done = steps_done_init; // monotonic across resets: the host throttles on differences
	m->pad3 = 0;
	m->core.Init(cfg.routing, n_paths, cfg.regret_budget, cfg.init_tuple_count, cfg.atc_multiplier);
	m->chunk_idx = m->chunk_end = 0;
	m->n_tuples = n_tuples;
	m->n_chunks = n_chunks;
	m->chunk_size = cfg.chunk_size;
	m->done = 1;
	m->chunk_offsets = nullptr;
	m->num_intermediates_total = 0;
	m->num_rounds = 0;
	m->log_enabled = cfg.log_rounds;
	m->max_log = cfg.max_log_rounds;
	m->n_log = 0;
	m->log_path = log_path;
	m->log_tuples = log_tuples;
	m->log_inter = log_inter;
	m->last_path = 0;
	for (uint32_t p = 0; p < POLR_MAX_PATHS; p++) {
		for (uint32_t j = 0; j < POLR_MAX_JOINS; j++) {
			m->stage_out[p][j] = 0;
		}
	}
	// (one thread: runs once per pass, ~300 stores)
}

__global__ void polr_mpx_set_range_kernel(DevMpx *m, uint64_t chunk_begin, uint64_t chunk_end,
                                          const uint64_t *chunk_offsets, uint64_t n_chunks, uint64_t n_tuples) {
	m->chunk_idx = chunk_begin;
	m->chunk_end = chunk_end;
	m->chunk_offsets = chunk_offsets;
	m->n_chunks = n_chunks;
	m->n_tuples = n_tuples;
	m->done = chunk_begin >= chunk_end ? 1 : 0;
}

// stand-alone routing step: primes the first round of a run (later rounds are routed by the last
// workgroup of each path-kernel launch)
__global__ void polr_mpx_router_kernel(DevMpx *m, DevRound *round, uint64_t *unit_prefix, uint32_t *unit_size_out,
                                       unsigned long long *counts, uint32_t k, uint32_t resident_waves) {
	polr_router_step(m, round, unit_prefix, unit_size_out, counts, k, resident_waves, threadIdx.x, false, nullptr);
}

// PushFinalize's closing FinalizePathRun (polar_pipeline_executor.cpp:150-151); one wave
__global__ void polr_mpx_finish_kernel(DevMpx *m, unsigned long long *counts, uint32_t k, polr_mpx_stats *stats) {
	const uint32_t lane = threadIdx.x;
	polr::MultiplexerCore &core = m->core;
	uint64_t s = 0;
	for (uint32_t j = 0; j < k; j++) {
		unsigned long long v = 0;
		if (lane < POLR_NSHARD) {
			v = counts[(uint64_t)lane * k + j];
			counts[(uint64_t)lane * k + j] = 0;
		}
		for (int d = 32; d > 0; d >>= 1) {
			v += __shfl_down(v, d, 64);
		}
		if (lane == 0) {
			s += v;
			m->stage_out[m->last_path][j] += v;
		}
	}
	if (lane == 0) {
		core.AddNumIntermediates(s);
		m->num_intermediates_total += s;
		polr_close_run(m);
	}
	__syncthreads();
	polr_write_stats(m, m, stats, lane);
}

extern "C" {

int polr_mpx_create(polr_pipeline *p, const polr_mpx_config *cfg, polr_mpx **out) {
	POLR_ENTRY();
	if (!p || !cfg || !out) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	*out = nullptr;
	if (cfg->routing > POLR_ROUTE_EXPONENTIAL_BACKOFF) {
		POLR_FAIL(ctx, POLR_E_INVALID, "unknown routing strategy %u", cfg->routing);
	}
	if (cfg->chunk_size < 2 || cfg->chunk_size > 65536) {
		POLR_FAIL(ctx, POLR_E_INVALID, "chunk size %u out of range", cfg->chunk_size);
	}
	if (p->n_paths > polr::kMaxPaths) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "too many join orders");
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	polr_mpx *m = new polr_mpx();
	m->pipe = p;
	m->ctx = polr_ctx_retain(p->ctx);
	m->cfg = *cfg;
	m->n_chunks = (p->n_tuples + cfg->chunk_size - 1) / cfg->chunk_size;
	const uint64_t max_log = cfg->log_rounds ? std::max<uint64_t>(cfg->max_log_rounds, 1) : 1;
	hipError_t e = hipMalloc((void **)&m->dev, sizeof(DevMpx));
	e = e == hipSuccess ? hipMalloc((void **)&m->round_dev, 2 * sizeof(DevRound)) : e;
	e = e == hipSuccess ? hipMalloc((void **)&m->prefix_dev, 4 * 8) : e;
	e = e == hipSuccess ? hipMalloc((void **)&m->unit_size_dev, 2 * 4) : e;
	e = e == hipSuccess ? hipMalloc((void **)&m->ticket_dev, 64) : e;
	e = e == hipSuccess ? hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) : e;
	e = e == hipSuccess ? hipMemset(m->ticket_dev, 0, 64) : e;
	// two counter banks (a resident run keeps up to two rounds in flight; everything else uses the first)
	e = e == hipSuccess ? hipMalloc((void **)&m->counts_dev, POLR_SLOTS * POLR_NSHARD * POLR_KMAX * 8) : e;
	e = e == hipSuccess ? hipMalloc((void **)&m->log_path, max_log * 4) : e;
	e = e == hipSuccess ? hipMalloc((void **)&m->log_tuples, max_log * 8) : e;
	e = e == hipSuccess ? hipMalloc((void **)&m->log_inter, max_log * 8) : e;
	e = e == hipSuccess ? hipHostMalloc((void **)&m->done_host, 64, hipHostMallocMapped) : e;
	if (e == hipSuccess) {
		memset(m->done_host, 0, 64);
		e = hipHostGetDevicePointer((void **)&m->progress_dev, m->done_host, 0);
	}
	e = e == hipSuccess ? hipMemset(m->counts_dev, 0, POLR_SLOTS * POLR_NSHARD * POLR_KMAX * 8) : e;
	e = e == hipSuccess ? hipMalloc((void **)&m->sync_dev, sizeof(ResidentSync)) : e;
	e = e == hipSuccess ? hipHostMalloc((void **)&m->stats_host, sizeof(polr_mpx_stats), hipHostMallocMapped) : e;
	e = e == hipSuccess ? hipHostGetDevicePointer((void **)&m->stats_host_dev, m->stats_host, 0) : e;
	e = e == hipSuccess ? hipMemset(m->sync_dev, 0, sizeof(ResidentSync)) : e;
	if (e != hipSuccess) {
		polr_mpx_destroy(m);
		POLR_FAIL(ctx, POLR_E_HIP, "multiplexer allocation failed: %s", hipGetErrorString(e));
	}
	m->cfg.max_log_rounds = (uint32_t)max_log;
	// every stage 0 takes wide (256-tuple) steps, so units are multiples of 256 tuples
	m->wide0_mask = p->n_paths >= 32 ? 0xFFFFFFFFu : ((1u << p->n_paths) - 1u);
	// (zero first: the bookkeeping of resident runs, res_valid / res_target, is not touched by the init kernel --
	// a host-side reset keeps it)
	e = hipMemsetAsync(m->dev, 0, sizeof(DevMpx), ctx->stream);
	if (e != hipSuccess) {
		polr_mpx_destroy(m);
		POLR_FAIL(ctx, POLR_E_HIP, "multiplexer init failed: %s", hipGetErrorString(e));
	}
	hipLaunchKernelGGL(polr_mpx_init_kernel, dim3(1), dim3(1), 0, ctx->stream, m->dev, m->cfg, p->n_paths, p->n_tuples,
	                   m->n_chunks, m->log_path, m->log_tuples, m->log_inter, m->wide0_mask, m->progress_dev, 0u);
	e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) {
		polr_mpx_destroy(m);
		POLR_FAIL(ctx, POLR_E_HIP, "multiplexer init failed: %s", hipGetErrorString(e));
	}
	*out = m;
	return POLR_OK;
}

int polr_mpx_set_chunk_offsets(polr_mpx *m, const uint64_t *offsets, uint64_t n_chunks) {
	POLR_ENTRY();
	if (!m || (!offsets && n_chunks)) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = m->pipe->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (m->chunk_offsets_dev && m->chunk_offsets_owned) {
		hipFree(m->chunk_offsets_dev);
	}
	m->chunk_offsets_dev = nullptr;
	m->chunk_offsets_owned = true;
	if (!offsets) {
		m->n_chunks = (m->pipe->n_tuples + m->cfg.chunk_size - 1) / m->cfg.chunk_size;
		return POLR_OK;
	}
	for (uint64_t c = 0; c < n_chunks; c++) {
		if (offsets[c + 1] < offsets[c] || offsets[c + 1] > m->pipe->n_tuples) {
			POLR_FAIL(ctx, POLR_E_INVALID, "chunk offsets must be non-decreasing and within the %llu source tuples",
			          (unsigned long long)m->pipe->n_tuples);
		}
	}
	HIPCHK(ctx, hipMalloc((void **)&m->chunk_offsets_dev, (n_chunks + 1) * 8));
	HIPCHK(ctx, hipMemcpy(m->chunk_offsets_dev, offsets, (n_chunks + 1) * 8, hipMemcpyHostToDevice));
	m->n_chunks = n_chunks;
	return POLR_OK;
}

// A pool launch about to be enqueued on `st`, sized for 1 / share of the device: make `st` wait for the pool launches of
// OTHER streams it cannot run beside -- all of them if this one takes the whole device, the full-size ones otherwise
// (launches that each declared a share, POLR_RUN_SHARE, are the caller's to add up) -- and open the entry whose event the
// caller records behind the launch.  Same-stream launches are ordered by the stream.
static int order_pool_launch(polr_ctx *ctx, hipStream_t st, uint32_t share) {
	// (the lists change only when every call below has succeeded)
	std::vector<polr_ctx::PoolLaunch> keep;
	std::vector<hipEvent_t> retired;
	for (auto &f : ctx->pool_launches) {
		bool retire = f.stream == st; // (superseded: this stream's next launch carries the newer event)
		if (!retire && (share <= 1 || f.share <= 1)) {
			HIPCHK(ctx, hipStreamWaitEvent(st, f.done, 0));
			retire = share <= 1; // (everything enqueued later waits for THIS launch, which waits for f)
		}
		if (retire) {
			retired.push_back(f.done);
		} else {
			keep.push_back(f);
		}
	}
	hipEvent_t done = nullptr;
	if (!retired.empty()) {
		done = retired.back();
		retired.pop_back();
	} else if (!ctx->pool_events_free.empty()) {
		done = ctx->pool_events_free.back();
		ctx->pool_events_free.pop_back();
	} else {
		HIPCHK(ctx, hipEventCreateWithFlags(&done, hipEventDisableTiming));
	}
	ctx->pool_events_free.insert(ctx->pool_events_free.end(), retired.begin(), retired.end());
	keep.push_back({st, done, share});
	ctx->pool_launches.swap(keep);
	return POLR_OK;
}

int polr_pipeline_launch_info(polr_pipeline *p, int materialize, polr_launch_info *info) {
	POLR_ENTRY();
	if (!p || !info) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const bool mat = materialize != 0;
	const bool flat = p->host_count.flat != 0 && (!mat || p->flat_emit);
	const DevPipeline &dp = (mat && !flat) ? p->host_mat : p->host_count;
	const uint32_t wq = dp.W + (dp.mult ? 1u : 0u); // slots per queued tuple in the pool launch: ids (+ multiplicity)
	const uint32_t wpb = flat ? p->flat_wpb : polr_pool_waves_per_block(dp.k, wq);
	memset(info, 0, sizeof(*info));
	info->waves_per_workgroup = wpb;
	const int occ = wpb == 0 ? 0 : (flat ? polr_pool_flat_occupancy(dp.k, wpb, dp.lds_table_dwords, mat) : polr_pool_occupancy(dp.k, wq, dp.ext != 0));
	if (occ < 1) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "per-wave LDS queues exceed 160 KB (too many joins x carried ids)");
	}
	info->workgroups_per_cu = (uint32_t)std::max(0, std::min(occ, 8));
	info->lds_bytes_per_workgroup = (uint32_t)(flat ? polr_pool_flat_lds_bytes(dp.k, wpb, dp.lds_table_dwords)
	                                                : polr_pool_lds_bytes(dp.k, wq));
	info->compiled_stages = dp.k <= 2 ? 2 : (dp.k <= 4 ? 4 : (dp.k <= 6 ? 6 : 8));
	info->tuple_slots = flat ? dp.W : wq;
	info->n_cus = (uint32_t)ctx->n_cus;
	info->flat = flat ? 1u : 0u;
	info->lds_tables = flat ? dp.n_lds_tables : 0u;
	info->lds_table_bytes = flat ? dp.lds_table_dwords * 4u : 0u;
	return POLR_OK;
}

// the source chunks are the ones polr_pipeline_scan_filter produced (boundaries stay on the device)
int polr_mpx_use_scan_chunks(polr_mpx *m) {
	POLR_ENTRY();
	if (!m) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = m->pipe;
	polr_ctx *ctx = p->ctx;
	if (!p->scan_valid) {
		POLR_FAIL(ctx, POLR_E_INVALID, "no scan result: call polr_pipeline_scan_filter first");
	}
	if (m->chunk_offsets_dev && m->chunk_offsets_owned) {
		HIPCHK(ctx, hipSetDevice(ctx->device));
		hipFree(m->chunk_offsets_dev);
	}
	m->chunk_offsets_dev = p->scan_offsets_dev;
	m->chunk_offsets_owned = false;
	m->scan_generation = p->scan_generation;
	m->n_chunks = p->scan_n_chunks;
	return POLR_OK;
}

// ---- a run = begin (set range, prime the first round) + a non-blocking pump that keeps a few
// self-routing launches queued ahead of the progress the device publishes -------------------------
struct RunState {
	polr_mpx *m = nullptr;
	hipStream_t st = nullptr;
	SelfRoute sr;
	DevOut dout;
	const DevPipeline *dpd = nullptr;
	uint32_t W = 0, k = 0, max_blocks = 0, wpb = 0;
	uint32_t launched = 0, base_steps = 0;
	bool finished = false;
};

static int run_begin(RunState &rs, polr_mpx *m, void *stream, uint64_t chunk_begin, uint64_t chunk_end,
                     polr_out *out, uint32_t share) {
	polr_pipeline *p = m->pipe;
	polr_ctx *ctx = p->ctx;
	if (!m->chunk_offsets_owned && (!p->scan_valid || m->scan_generation != p->scan_generation)) {
		POLR_FAIL(ctx, POLR_E_INVALID, "the pipeline was scanned again: call polr_mpx_use_scan_chunks");
	}
	if (chunk_begin > chunk_end || chunk_end > m->n_chunks) {
		POLR_FAIL(ctx, POLR_E_INVALID, "chunks [%llu, %llu) outside the %llu source chunks",
		          (unsigned long long)chunk_begin, (unsigned long long)chunk_end, (unsigned long long)m->n_chunks);
	}
	if (out && out->pipe != p) {
		POLR_FAIL(ctx, POLR_E_INVALID, "output object belongs to another pipeline");
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = stream ? (hipStream_t)stream : m->own_stream;
	HIPCHK(ctx, adopt_stream(m, st));
	const bool materialize = out != nullptr;
	uint32_t unit_unused, max_blocks;
	int rc = polr_plan_launch(p, materialize, p->n_tuples, &unit_unused, &max_blocks);
	if (rc) {
		return rc;
	}
#ifdef POLR_DIAG_STAMPS
	if (!m->stamps_dev) {
		HIPCHK(ctx, hipMalloc((void **)&m->stamps_dev, 4096 * 8 * 8));
		HIPCHK(ctx, hipMemset(m->stamps_dev, 0, 4096 * 8 * 8));
	}
#endif
	// executors that run concurrently share the device: each sizes its units and grid for its share
	uint32_t resident_waves = std::max<uint32_t>(polr_resident_waves(p, materialize) / std::max<uint32_t>(share, 1), 256);
	const uint32_t wpb = polr_waves_per_block(p, materialize);
	max_blocks = std::max<uint32_t>(max_blocks / std::max<uint32_t>(share, 1), 64);
	rs.m = m;
	rs.st = st;
	m->stats_in_host = false;
	memset(&rs.dout, 0, sizeof(rs.dout));
	if (out) {
		rs.dout = out->dev;
		out->stats_valid = false;
	}
	const DevPipeline &dp = materialize ? p->host_mat : p->host_count;
	rs.dpd = materialize ? p->dev_mat : p->dev_count;
	rs.W = dp.W;
	rs.k = dp.k;
	rs.max_blocks = max_blocks;
	rs.wpb = wpb;
	if (m->pending_sync) { // a previous run left launches queued: settle before reading progress
		HIPCHK(ctx, hipStreamSynchronize(st));
		m->pending_sync = false;
	}
	hipLaunchKernelGGL(polr_mpx_set_range_kernel, dim3(1), dim3(1), 0, st, m->dev, chunk_begin, chunk_end,
	                   (const uint64_t *)m->chunk_offsets_dev, m->n_chunks, p->n_tuples);
	m->steps_base = ((volatile uint32_t *)m->done_host)[0];
	((volatile uint32_t *)m->done_host)[1] = 0;
	// prime: route the first round of this run into the descriptor slot the next launch reads
	const uint32_t slot0 = m->iter & 1u;
	hipLaunchKernelGGL(polr_mpx_router_kernel, dim3(1), dim3(64), 0, st, m->dev, m->round_dev + slot0,
	                   m->prefix_dev + 2 * slot0, m->unit_size_dev + slot0, m->counts_dev, p->k, resident_waves);
	rs.sr.mpx = m->dev;
	rs.sr.rounds_base = m->round_dev;
	rs.sr.prefix_base = m->prefix_dev;
	rs.sr.unit_base = m->unit_size_dev;
	rs.sr.ticket = m->ticket_dev;
	rs.sr.resident_waves = resident_waves;
	rs.sr.stamps = m->stamps_dev;
	rs.sr.iter = 0;
	rs.launched = 0;
	rs.base_steps = m->steps_base;
	rs.finished = false;
	m->last_stream = st;
	return POLR_OK;
}

// Every launch probes the round in its slot and its last workgroup routes the next one.  The host never
// synchronises inside a run: it keeps a few launches queued ahead of the progress the device publishes
// (pinned host words written by the router) and stops when the router reports the end of the range.
// Launches queued past the end find an empty round and exit.  Non-blocking: returns after at most one launch.
static int run_pump(RunState &rs) {
	polr_mpx *m = rs.m;
	polr_ctx *ctx = m->pipe->ctx;
	volatile uint32_t *prog = (volatile uint32_t *)m->done_host;
	const uint32_t look_ahead = 3;
	const uint32_t steps = prog[0] - rs.base_steps; // 1 after the prime step, +1 per routed launch
	if (prog[1] && steps >= 1) {
		rs.finished = true; // the router has seen the end of the range
		m->steps_base = prog[0];
		m->pending_sync = true; // the queued tail is drained by whoever synchronises next
		return POLR_OK;
	}
	if (rs.launched + 1 > steps + look_ahead) {
		return POLR_OK; // enough launches in flight
	}
	size_t ev = 0;
	if (m->timing) {
		ev = m->ev_used++;
		if (ev >= m->ev_start.size()) {
			hipEvent_t a, b;
			HIPCHK(ctx, hipEventCreate(&a));
			HIPCHK(ctx, hipEventCreate(&b));
			m->ev_start.push_back(a);
			m->ev_stop.push_back(b);
		}
		HIPCHK(ctx, hipEventRecord(m->ev_start[ev], rs.st));
	}
	rs.sr.iter = m->iter++;
	hipError_t e = polr_launch_path_kernel(rs.W, rs.k, rs.max_blocks, rs.wpb, rs.st, rs.dpd, m->round_dev,
	                                       m->prefix_dev, 1, m->unit_size_dev, rs.dout, m->counts_dev, rs.sr);
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "path kernel launch failed: %s", hipGetErrorString(e));
	}
	if (m->timing) {
		HIPCHK(ctx, hipEventRecord(m->ev_stop[ev], rs.st));
	}
	rs.launched++;
	return POLR_OK;
}

int polr_mpx_run(polr_mpx *m, void *stream, uint64_t chunk_begin, uint64_t chunk_end, polr_out *out) {
	POLR_ENTRY();
	if (!m) {
		return POLR_E_INVALID;
	}
	RunState rs;
	int rc = run_begin(rs, m, stream, chunk_begin, chunk_end, out, 1);
	while (!rc && !rs.finished) {
		rc = run_pump(rs);
	}
	return rc;
}

// Several executors at once (the reference runs one PipelineExecutor + MultiplexerState per worker
// thread over morsels of one pipeline, pipeline.cpp:145-174): every multiplexer routes its own chunk
// range on its own stream; one host thread pumps them round-robin, so their routing rounds overlap on
// the device instead of queueing behind each other.
int polr_mpx_run_many(polr_mpx **ms, void **streams, const uint64_t *chunk_begin, const uint64_t *chunk_end,
                      uint32_t n, polr_out *out) {
	POLR_ENTRY();
	if (!ms || !chunk_begin || !chunk_end || n == 0) {
		return POLR_E_INVALID;
	}
	std::vector<RunState> rs(n);
	int rc = POLR_OK;
	for (uint32_t i = 0; i < n && !rc; i++) {
		if (!ms[i] || ms[i]->pipe != ms[0]->pipe) {
			return POLR_E_INVALID;
		}
		rc = run_begin(rs[i], ms[i], streams ? streams[i] : nullptr, chunk_begin[i], chunk_end[i], out, n);
	}
	bool all_done = false;
	while (!rc && !all_done) {
		all_done = true;
		for (uint32_t i = 0; i < n && !rc; i++) {
			if (!rs[i].finished) {
				rc = run_pump(rs[i]);
				all_done = all_done && rs[i].finished;
			}
		}
	}
	return rc;
}

// The whole run in ONE launch (polr_pool.hip): one router wave per executor + a pool of probe waves that serves
// the rounds of all executors; routing decisions never leave the device, the host only enqueues.  Asynchronous:
// polr_mpx_finish / _finish_many synchronise.
static uint32_t next_pow2_u32(uint64_t v) {
	uint32_t p = 1;
	while (p < v) {
		p <<= 1;
	}
	return p;
}

#define POOL_HEADER_BYTES 512
static_assert(sizeof(PoolRun) <= POOL_HEADER_BYTES, "run header");

// ranges_per_exec > 1: chunk_begin / chunk_end are [n][ranges_per_exec] (executor i routes its ranges in order)
static int run_resident_impl(polr_mpx **ms, void *stream, const uint64_t *chunk_begin, const uint64_t *chunk_end,
                             uint32_t n, polr_out *out, uint32_t flags, uint64_t morsel_begin, uint64_t morsel_end,
                             uint32_t morsel_chunks, bool backpressure = false, uint32_t ranges_per_exec = 1) {
	if (!ms || n == 0 || !ms[0] || (morsel_chunks == 0 && (!chunk_begin || !chunk_end))) {
		return POLR_E_INVALID;
	}
	polr_mpx *m0 = ms[0];
	polr_pipeline *p = m0->pipe;
	polr_ctx *ctx = p->ctx;
	if (n > 4096) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "at most 4096 executors per run");
	}
	if (p->n_tuples >= 0xFFFFFFF0ull) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "source partition too large for 32-bit tuple positions");
	}
	if (out && out->pipe != p) {
		POLR_FAIL(ctx, POLR_E_INVALID, "output object belongs to another pipeline");
	}
	for (uint32_t i = 0; i < n; i++) {
		if (!ms[i] || ms[i]->pipe != p) {
			return POLR_E_INVALID;
		}
		if (!ms[i]->chunk_offsets_owned && (!p->scan_valid || ms[i]->scan_generation != p->scan_generation)) {
			POLR_FAIL(ctx, POLR_E_INVALID, "the pipeline was scanned again: call polr_mpx_use_scan_chunks");
		}
		for (uint32_t r = 0; r < (morsel_chunks ? 1u : ranges_per_exec); r++) {
			const uint64_t cb = morsel_chunks ? morsel_begin : chunk_begin[(size_t)i * ranges_per_exec + r];
			const uint64_t ce = morsel_chunks ? morsel_end : chunk_end[(size_t)i * ranges_per_exec + r];
			if (cb > ce || ce > ms[i]->n_chunks) {
				POLR_FAIL(ctx, POLR_E_INVALID, "chunks [%llu, %llu) outside the %llu source chunks", (unsigned long long)cb,
				          (unsigned long long)ce, (unsigned long long)ms[i]->n_chunks);
			}
		}
	}
	{
		std::vector<polr_mpx *> sorted(ms, ms + n);
		std::sort(sorted.begin(), sorted.end());
		if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
			POLR_FAIL(ctx, POLR_E_INVALID, "the same multiplexer twice in one run");
		}
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = stream ? (hipStream_t)stream : m0->own_stream;
	const bool materialize = out != nullptr;
	// a bank of single-key unique-match joins takes the flat pipeline -- counting runs always, emitting runs when every
	// join is a perfect table (polr_flat_device.h); it runs on the counting variant's descriptors either way
	const bool flat = p->host_count.flat != 0 && (!materialize || p->flat_emit);
	const DevPipeline &dp = (materialize && !flat) ? p->host_mat : p->host_count;
	const uint32_t wq = dp.W + (dp.mult ? 1u : 0u); // slots per queued tuple: ids (+ multiplicity)
	const uint32_t wpb = flat ? p->flat_wpb : polr_pool_waves_per_block(dp.k, wq);
	int occ = wpb == 0 ? 0 : (flat ? polr_pool_flat_occupancy(dp.k, wpb, dp.lds_table_dwords, materialize) : polr_pool_occupancy(dp.k, wq, dp.ext != 0));
	if (occ < 1) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "pool kernel does not fit on a CU (per-wave LDS queues: too many joins x carried ids)");
	}
	occ = std::min(occ, 8);
	uint32_t share = std::max<uint32_t>((flags >> 8) & 0xFFu, 1u);
	if (ctx->tuning.device_share) { // (polr_ctx_set_pool_tuning)
		share = ctx->tuning.device_share;
	}
	if (share > 16) {
		POLR_FAIL(ctx, POLR_E_INVALID, "device share 1/%u: at most 16 runs side by side", share);
	}
	// grid: router workgroups first (one wave per executor), then the pool; never more than is co-resident
	const uint32_t capacity = std::max<uint32_t>((uint32_t)ctx->n_cus * (uint32_t)occ / share, 2u);
	const uint32_t n_router_blocks = (n + wpb - 1) / wpb;
	if (n_router_blocks + 1 > capacity) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "%u executors do not fit on the device at once", n);
	}
	const uint32_t n_blocks = capacity;
	const uint32_t n_workers = n_blocks - n_router_blocks;
	for (uint32_t i = 0; i < n; i++) {
		// (work still queued on `st` needs no host synchronisation: everything a run touches is ordered by the
		// stream -- consecutive passes can be enqueued back to back)
		HIPCHK(ctx, adopt_stream(ms[i], st));
	}
	const size_t execs_bytes = POOL_HEADER_BYTES + (size_t)n * sizeof(ResidentExec) + 64;
	if (m0->execs_cap < n) {
		if (m0->execs_dev) {
			HIPCHK(ctx, hipStreamSynchronize(st));
			hipFree(m0->execs_dev);
			m0->execs_dev = nullptr;
		}
		const uint32_t cap = std::max<uint32_t>(n, 8);
		HIPCHK(ctx, hipMalloc((void **)&m0->execs_dev, POOL_HEADER_BYTES + (size_t)cap * sizeof(ResidentExec) + 64));
		m0->execs_cap = cap;
		m0->execs_host.clear();
	}
	// unit rings: sized for everything the executors of this run can have in flight (two slots each) plus the EXIT
	// entries, with a factor of two to spare
	const uint32_t pool_waves = n_workers * wpb;
	// Rings in use: every ring must have probe waves that serve it.  Ring capacity: a round of U units leaves at most
	// U / R + 1 entries on a ring; the executors that have rounds in flight (a of them, at most POLR_SLOTS rounds each) published them
	// when at least a executors were still routing, so all their lo units together are at most POLR_SLOTS x (4 x pool_waves + 17 a);
	// a hi round has at most POLR_POOL_HI_TUPLES / 64 units.  Twice that, plus the EXIT entries.
	uint32_t n_rings = 1;
	while (n_rings * 2 <= std::min<uint32_t>(POLR_POOL_RINGS, pool_waves)) {
		n_rings *= 2;
	}
	const uint64_t R = n_rings;
	// (+ a round larger than target x 65 536 tuples has tuples / 65 536 units)
	const uint32_t lo_cap = next_pow2_u32(2ull * ((4ull * POLR_SLOTS * pool_waves + 17ull * POLR_SLOTS * n +
	                                               (uint64_t)POLR_SLOTS * (p->n_tuples >> 16)) / R +
	                                              (uint64_t)POLR_SLOTS * n + pool_waves / R + 16) + 64);
	// (+ work sharing: a probe wave has at most one shared piece outstanding, published on the ring after its own)
	const uint32_t hi_cap = next_pow2_u32(
	    2ull * ((uint64_t)POLR_SLOTS * n * (POLR_POOL_HI_TUPLES / POLR_POOL_HI_UNIT / R + 1) + pool_waves / R + 1) + 64);
	if (((volatile uint32_t *)m0->done_host)[2]) {
		// an earlier run on these rings was given up (whoever finished it): probe waves left holding tickets
		m0->pool_dirty = true;
		if (!m0->pending_sync) {
			((volatile uint32_t *)m0->done_host)[2] = 0; // (nothing in flight that could still report it)
		}
	}
	// work sharing (generic pipeline only): records of 8 + 64 x (words per queued tuple) dwords, one per probe wave, and
	// their flags -- all flags are 0 between runs (a record is released by the wave that took it) unless a run was given up
	const uint32_t share_after = flat ? 0xFFFFFFFFu : (ctx->tuning.share_after ? ctx->tuning.share_after : 32u);
	const uint32_t share_stride = 8u + 64u * wq;
	uint32_t *share_recs = nullptr, *share_flags = nullptr;
	if (share_after != 0xFFFFFFFFu) {
		const size_t need = ((size_t)pool_waves * share_stride + pool_waves) * sizeof(uint32_t);
		if (m0->share_bytes < need) {
			if (m0->share_dev) {
				HIPCHK(ctx, hipStreamSynchronize(st));
				hipFree(m0->share_dev);
				m0->share_dev = nullptr;
				m0->share_bytes = 0;
			}
			HIPCHK(ctx, hipMalloc((void **)&m0->share_dev, need));
			m0->share_bytes = need;
			HIPCHK(ctx, hipMemsetAsync(m0->share_dev, 0, need, st));
		} else if (m0->pool_dirty) {
			HIPCHK(ctx, hipMemsetAsync(m0->share_dev, 0, m0->share_bytes, st));
		}
		share_recs = m0->share_dev;
		share_flags = m0->share_dev + (size_t)pool_waves * share_stride;
	}
	if (!m0->pool_dev || m0->pool_lo_cap < lo_cap || m0->pool_hi_cap < hi_cap || m0->pool_dirty) {
		if (m0->pool_dev && (m0->pool_lo_cap < lo_cap || m0->pool_hi_cap < hi_cap)) {
			HIPCHK(ctx, hipStreamSynchronize(st));
			hipFree(m0->pool_dev);
			m0->pool_dev = nullptr;
		}
		const uint32_t lc = std::max(lo_cap, m0->pool_lo_cap), hc = std::max(hi_cap, m0->pool_hi_cap);
		const size_t bytes = sizeof(PoolSync) + (size_t)POLR_POOL_RINGS * (2 * (size_t)lc + hc) * sizeof(PoolEntry);
		if (!m0->pool_dev) {
			HIPCHK(ctx, hipMalloc((void **)&m0->pool_dev, bytes));
		}
		HIPCHK(ctx, hipMemsetAsync(m0->pool_dev, 0, bytes, st));
		m0->pool_lo_cap = lc;
		m0->pool_hi_cap = hc;
		m0->pool_dirty = false;
	}
	std::vector<char> host(execs_bytes, 0);
	PoolRun *hr = (PoolRun *)host.data();
	ResidentExec *ex = (ResidentExec *)(host.data() + POOL_HEADER_BYTES);
	unsigned long long *cursor_dev =
	    (unsigned long long *)(m0->execs_dev + POOL_HEADER_BYTES + (size_t)m0->execs_cap * sizeof(ResidentExec));
	hr->sync = m0->pool_dev;
	hr->n_exec = n;
	hr->n_router_blocks = n_router_blocks;
	hr->n_rings = n_rings;
	{
		const polr_pool_tuning &tn = ctx->tuning; // (polr_ctx_set_pool_tuning; 0 = default)
		hr->units_x = tn.units_x ? tn.units_x : 4u; // (ring capacities are sized for 4)
		// tuples per unit of a small round.  Default: flat: two steps of the pipeline's stage 0 (1 024 tuples = one
		// exploration slice of init_tuple_count in one
		// unit: measured 1.52 ms against 1.55-1.56 with 512 on the SF100 run), generic: a wide step of 256.  64-tuple units
		// finish a lone small round soonest, but a unit costs its wave the same chain of dependent round trips whatever
		// its size, and with hundreds of executors exploring that wave time is what the pool runs out of (measured on
		// the SF100 run: 2.29 ms with 64-tuple units, 1.77 ms with 512)
		{
			// the fewest probe waves any ring has; the lottery divides them into at most 8 classes
			const uint32_t min_waves = pool_waves / n_rings;
			uint32_t lot = 1;
			while (lot * 2 <= std::min<uint32_t>(8, min_waves)) {
				lot *= 2;
			}
			hr->hi_lottery = (tn.hi_lottery >= 1 && tn.hi_lottery <= lot) ? tn.hi_lottery : lot;
		}
		// (generic pipelines with few executors: 64 -- an exploration slice spread over 16 waves; measured on the 113
		// JOB-shaped pipelines, 8 executors each: 46.9 ms per pass against 47.4 with 128 and 49.8 with 256)
		hr->hi_unit = tn.hi_unit ? tn.hi_unit : (flat ? 1024u : (n <= 64u ? 64u : 256u));
	}
	for (uint32_t r = 0; r < POLR_POOL_RINGS; r++) {
		hr->worker_waves[r] = r < n_rings ? (pool_waves + n_rings - 1 - r) / n_rings : 0u; // (wave g serves ring g % n_rings)
	}
	hr->pool_waves = pool_waves;
	hr->lo_cap = m0->pool_lo_cap;
	hr->hi_cap = m0->pool_hi_cap;
	// (the size up to which a round is latency-critical)
	hr->hi_tuples = ctx->tuning.hi_tuples_p1 ? std::min<uint32_t>(ctx->tuning.hi_tuples_p1 - 1u, POLR_POOL_HI_TUPLES)
	                                         : POLR_POOL_HI_TUPLES;
	hr->idle_sleep = ctx->tuning.idle_sleep == 16 ? 16u : 64u; // (an idle probe wave's longest back-off)
	// watchdog: ticks of the 100 MHz wall clock (default 4 s: a wait this long is a lost run)
	hr->timeout_ticks = ctx->tuning.watchdog_us ? (unsigned long long)ctx->tuning.watchdog_us * 100ull : POLR_RES_TIMEOUT_TICKS;
	hr->share_recs = share_recs;
	hr->share_flags = share_flags;
	hr->share_stride = share_stride;
	hr->share_after = share_after;
	hr->routers_done = 0;
	hr->abort = 0;
	// the rings belong to the multiplexer that leads the run: a run that is given up says so in ITS host words too,
	// whichever router saw the watchdog fire (the leader's own router may have finished long before)
	hr->host_words = m0->progress_dev;
	for (uint32_t i = 0; i < n; i++) {
		polr_mpx *m = ms[i];
		m->leader = m0;
		ex[i].mpx = m->dev;
		ex[i].sync = m->sync_dev;
		ex[i].counts = m->counts_dev;
		ex[i].chunk_begin = morsel_chunks ? 0 : chunk_begin[(size_t)i * ranges_per_exec];
		ex[i].chunk_end = morsel_chunks ? 0 : chunk_end[(size_t)i * ranges_per_exec];
		ex[i].n_more = morsel_chunks ? 0 : ranges_per_exec - 1;
		for (uint32_t r = 1; r < ranges_per_exec && !morsel_chunks; r++) {
			ex[i].more_begin[r - 1] = chunk_begin[(size_t)i * ranges_per_exec + r];
			ex[i].more_end[r - 1] = chunk_end[(size_t)i * ranges_per_exec + r];
		}
		ex[i].morsel_cursor = morsel_chunks ? cursor_dev : nullptr;
		ex[i].morsel_end = morsel_end;
		ex[i].morsel_chunks = morsel_chunks;
		ex[i].path_plus1 = backpressure ? i + 1 : 0;
		ex[i].chunk_offsets = m->chunk_offsets_dev;
		ex[i].n_chunks = m->n_chunks;
		ex[i].n_tuples = p->n_tuples;
		ex[i].flags = flags;
		ex[i].pad = 0;
		ex[i].stats_out = m->stats_host_dev;
		m->stats_in_host = (flags & POLR_RUN_FINISH) != 0;
		((volatile uint32_t *)m->done_host)[1] = 0;
	}
	// A pass that repeats the previous one (same executors, ranges, flags: every step of a measurement loop) finds its
	// descriptors on the device already; only the two words the device writes (routers_done, abort) are cleared.
	// Otherwise: one copy (pageable source: staged by the runtime before the call returns).
	const size_t used = POOL_HEADER_BYTES + (size_t)n * sizeof(ResidentExec);
	if (m0->execs_host.size() == used && memcmp(m0->execs_host.data(), host.data(), used) == 0) {
		static_assert(offsetof(PoolRun, abort) == offsetof(PoolRun, routers_done) + 4, "cleared together");
		HIPCHK(ctx, hipMemsetAsync(m0->execs_dev + offsetof(PoolRun, routers_done), 0, 8, st));
	} else {
		HIPCHK(ctx, hipMemcpyAsync(m0->execs_dev, host.data(), used, hipMemcpyHostToDevice, st));
		m0->execs_host.assign(host.begin(), host.begin() + used);
	}
	if (morsel_chunks) {
		const unsigned long long first = morsel_begin;
		HIPCHK(ctx, hipMemcpyAsync(cursor_dev, &first, 8, hipMemcpyHostToDevice, st));
	}
	DevOut dout;
	memset(&dout, 0, sizeof(dout));
	if (out) {
		dout = out->dev;
		out->stats_valid = false;
	}
	// order this launch behind the pool launches of other streams it must not share the device with
	{
		int rc_o = order_pool_launch(ctx, st, share);
		if (rc_o) {
			return rc_o;
		}
	}
	size_t ev = 0;
	if (m0->timing) {
		ev = m0->ev_used++;
		if (ev >= m0->ev_start.size()) {
			hipEvent_t a, b;
			HIPCHK(ctx, hipEventCreate(&a));
			HIPCHK(ctx, hipEventCreate(&b));
			m0->ev_start.push_back(a);
			m0->ev_stop.push_back(b);
		}
		HIPCHK(ctx, hipEventRecord(m0->ev_start[ev], st));
	}
	const ResidentExec *execs_dev = (const ResidentExec *)(m0->execs_dev + POOL_HEADER_BYTES);
	// a fused GROUP BY sink keeps its cells in the workgroup's LDS when they fit behind the bit tables and the queues
	// (160 KB per workgroup), else in its table in global memory
	uint32_t fused_words = 0;
	if (flat && out && out->fused_dev) {
		const uint64_t words = (uint64_t)out->fused_groups * (1u + 2u * out->fused_aggs);
		if (polr_pool_flat_lds_bytes(dp.k, wpb, dp.lds_table_dwords) + 8 + words * 8 <= 160u * 1024u) {
			fused_words = (uint32_t)words;
		}
	}
	hipError_t e =
	    flat ? polr_launch_pool_flat_kernel(dp.k, n_blocks, wpb, dp.lds_table_dwords, st, p->dev_count, execs_dev,
	                                        (PoolRun *)m0->execs_dev, dout, materialize, fused_words)
	         : polr_launch_pool_kernel(wq, dp.k, n_blocks, st, materialize ? p->dev_mat : p->dev_count, execs_dev,
	                                   (PoolRun *)m0->execs_dev, dout, dp.ext != 0);
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "pool kernel launch failed: %s", hipGetErrorString(e));
	}
	if (m0->timing) {
		HIPCHK(ctx, hipEventRecord(m0->ev_stop[ev], st));
	}
	HIPCHK(ctx, hipEventRecord(ctx->pool_launches.back().done, st)); // (the entry order_pool_launch made for this launch)
	for (uint32_t i = 0; i < n; i++) {
		ms[i]->pending_sync = true;
	}
	return POLR_OK;
}

int polr_mpx_run_resident_ranges(polr_mpx **ms, void *stream, const uint64_t *range_begin, const uint64_t *range_end,
                                 uint32_t ranges_per_executor, uint32_t n, polr_out *out, uint32_t flags) {
	POLR_ENTRY();
	if (ranges_per_executor < 1 || ranges_per_executor > POLR_MORE_RANGES + 1) {
		return POLR_E_INVALID;
	}
	return run_resident_impl(ms, stream, range_begin, range_end, n, out, flags, 0, 0, 0, false, ranges_per_executor);
}

int polr_mpx_run_resident(polr_mpx **ms, void *stream, const uint64_t *chunk_begin, const uint64_t *chunk_end,
                          uint32_t n, polr_out *out, uint32_t flags) {
	POLR_ENTRY();
	return run_resident_impl(ms, stream, chunk_begin, chunk_end, n, out, flags, 0, 0, 0);
}

int polr_mpx_run_resident_morsels(polr_mpx **ms, void *stream, uint64_t chunk_begin, uint64_t chunk_end,
                                  uint32_t morsel_chunks, uint32_t n, polr_out *out, uint32_t flags) {
	POLR_ENTRY();
	if (morsel_chunks == 0) {
		return POLR_E_INVALID;
	}
	return run_resident_impl(ms, stream, nullptr, nullptr, n, out, flags, chunk_begin, chunk_end, morsel_chunks);
}

// BACKPRESSURE routing (MultiplexerRouting::BACKPRESSURE): one executor per join order, all pulling morsels from one
// cursor -- the join orders race for the source (src/parallel/pipeline.cpp:147-156: one PipelineTask per join order
// over ONE shared source state; polar_config.cpp:128-147).  Executor i runs join order i.
int polr_mpx_run_backpressure(polr_mpx **ms, void *stream, uint64_t chunk_begin, uint64_t chunk_end,
                              uint32_t morsel_chunks, polr_out *out, uint32_t flags) {
	POLR_ENTRY();
	if (!ms || !ms[0] || morsel_chunks == 0) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = ms[0]->pipe;
	polr_ctx *ctx = p->ctx;
	const uint32_t n = p->n_paths;
	for (uint32_t i = 0; i < n; i++) {
		if (!ms[i] || ms[i]->pipe != p) {
			POLR_FAIL(ctx, POLR_E_INVALID, "BACKPRESSURE needs one multiplexer per join order (%u) of the same pipeline", n);
		}
		if (ms[i]->cfg.routing != POLR_ROUTE_BACKPRESSURE && ms[i]->cfg.routing != POLR_ROUTE_DEFAULT_PATH) {
			POLR_FAIL(ctx, POLR_E_INVALID, "multiplexer %u does not route BACKPRESSURE / DEFAULT_PATH", i);
		}
	}
	return run_resident_impl(ms, stream, nullptr, nullptr, n, out, flags, chunk_begin, chunk_end, morsel_chunks, true);
}

int polr_mpx_reset(polr_mpx *m, void *stream) {
	POLR_ENTRY();
	if (!m) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = m->pipe->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = pick_stream(m, stream);
	HIPCHK(ctx, adopt_stream(m, st));
	if (m->pending_sync) { // launches of the previous pass may still be queued on its stream
		HIPCHK(ctx, hipStreamSynchronize(st));
		m->pending_sync = false;
	}
	m->stats_in_host = false;
	HIPCHK(ctx, hipMemsetAsync(m->counts_dev, 0, POLR_SLOTS * POLR_NSHARD * POLR_KMAX * 8, st));
	hipLaunchKernelGGL(polr_mpx_init_kernel, dim3(1), dim3(1), 0, st, m->dev, m->cfg, m->pipe->n_paths,
	                   m->pipe->n_tuples, m->n_chunks, m->log_path, m->log_tuples, m->log_inter, m->wide0_mask,
	                   m->progress_dev, ((volatile uint32_t *)m->done_host)[0]);
	return POLR_OK;
}

#ifdef POLR_DIAG_STAMPS
// diagnostic: dump the stamps of the first `n` launches (100 MHz ticks)
extern "C" int polr_mpx_dump_stamps(polr_mpx *m, unsigned long long *dst, uint32_t n) {
	return hipMemcpy(dst, m->stamps_dev, (size_t)n * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

int polr_mpx_enable_timing(polr_mpx *m, int enable) {
	POLR_ENTRY();
	if (!m) {
		return POLR_E_INVALID;
	}
	m->timing = enable != 0;
	return POLR_OK;
}

int polr_mpx_kernel_time(polr_mpx *m, double *total_ms, uint64_t *n_launches) {
	POLR_ENTRY();
	if (!m || !total_ms || !n_launches) {
		return POLR_E_INVALID;
	}
	*total_ms = m->timed_ms;
	*n_launches = m->timed_launches;
	m->timed_ms = 0;
	m->timed_launches = 0;
	return POLR_OK;
}

int polr_mpx_finish(polr_mpx *m, void *stream, polr_mpx_stats *stats) {
	POLR_ENTRY();
	if (!m || !stats) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = m->pipe->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = pick_stream(m, stream);
	HIPCHK(ctx, adopt_stream(m, st));
	hipError_t e = hipSuccess;
	if (m->stats_in_host) { // the resident run closed itself: nothing to launch
		e = hipStreamSynchronize(st);
		memcpy(stats, m->stats_host, sizeof(polr_mpx_stats));
	} else {
		if (!m->stats_dev) {
			HIPCHK(ctx, hipMalloc((void **)&m->stats_dev, sizeof(polr_mpx_stats)));
		}
		hipLaunchKernelGGL(polr_mpx_finish_kernel, dim3(1), dim3(64), 0, st, m->dev, m->counts_dev, m->pipe->k,
		                   m->stats_dev);
		e = hipMemcpyAsync(stats, m->stats_dev, sizeof(polr_mpx_stats), hipMemcpyDeviceToHost, st);
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	}
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "multiplexer finish failed: %s", hipGetErrorString(e));
	}
	m->pending_sync = false;
	if (((volatile uint32_t *)m->done_host)[2]) {
		((volatile uint32_t *)m->done_host)[2] = 0;
		m->pool_dirty = true;
		if (m->leader) {
			m->leader->pool_dirty = true; // (the rings of the run belong to its first multiplexer)
		}
		POLR_FAIL(ctx, POLR_E_HIP, "run timed out waiting for its probe waves (results incomplete)");
	}
	if (m->timing) {
		drain_events(m);
	}
	return POLR_OK;
}

// finish several executors: all closing kernels and read-backs are queued first, then each stream is
// synchronised once (saves n-1 serial round trips)
int polr_mpx_finish_many(polr_mpx **ms, uint32_t n, polr_mpx_stats *stats) {
	POLR_ENTRY();
	if (!ms || !stats || n == 0) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = ms[0]->pipe->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	for (uint32_t i = 0; i < n; i++) {
		polr_mpx *m = ms[i];
		hipStream_t st = m->last_stream ? m->last_stream : m->own_stream;
		if (m->stats_in_host) {
			continue; // closed inside its resident run
		}
		if (!m->stats_dev) {
			HIPCHK(ctx, hipMalloc((void **)&m->stats_dev, sizeof(polr_mpx_stats)));
		}
		hipLaunchKernelGGL(polr_mpx_finish_kernel, dim3(1), dim3(64), 0, st, m->dev, m->counts_dev, m->pipe->k,
		                   m->stats_dev);
		HIPCHK(ctx, hipMemcpyAsync(&stats[i], m->stats_dev, sizeof(polr_mpx_stats), hipMemcpyDeviceToHost, st));
	}
	bool timed_out = false;
	hipStream_t synced = nullptr;
	for (uint32_t i = 0; i < n; i++) {
		polr_mpx *m = ms[i];
		hipStream_t st = m->last_stream ? m->last_stream : m->own_stream;
		if (st != synced) { // (executors of a resident run share one stream)
			HIPCHK(ctx, hipStreamSynchronize(st));
			synced = st;
		}
		if (m->stats_in_host) {
			memcpy(&stats[i], m->stats_host, sizeof(polr_mpx_stats));
		}
		m->pending_sync = false;
		if (m->timing) {
			drain_events(m);
		}
		if (((volatile uint32_t *)m->done_host)[2]) {
			((volatile uint32_t *)m->done_host)[2] = 0;
			timed_out = true;
		}
		if (timed_out) {
			ms[0]->pool_dirty = true;
			m->pool_dirty = true;
			if (m->leader) {
				m->leader->pool_dirty = true; // (the multiplexer that led the run owns its rings)
			}
		}
	}
	if (timed_out) {
		// whose watchdog fired, and on what (the routers that saw it leave theirs in their pinned words)
		std::string who;
		for (uint32_t i = 0; i < n; i++) {
			volatile uint32_t *w = (volatile uint32_t *)ms[i]->done_host;
			if (w && w[4]) {
				char buf[200];
				snprintf(buf, sizeof(buf), "%s executor %u: slot %u, %u round(s) in flight, %llu of %llu arrival tokens (65536 per unit)",
				         who.empty() ? ";" : ",", w[5], w[6], w[7], ((unsigned long long)w[11] << 32) | w[10],
				         ((unsigned long long)w[9] << 32) | w[8]);
				who += buf;
				w[4] = 0;
			}
		}
		POLR_FAIL(ctx, POLR_E_HIP, "run timed out waiting for its probe waves (results incomplete)%s", who.c_str());
	}
	return POLR_OK;
}

int polr_mpx_fetch_log(polr_mpx *m, void *stream, uint32_t *path, uint64_t *tuples, uint64_t *intermediates,
                       uint64_t max_rounds, uint64_t *n_rounds) {
	POLR_ENTRY();
	if (!m || !n_rounds) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = m->pipe->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = pick_stream(m, stream);
	DevMpx h;
	HIPCHK(ctx, hipMemcpyAsync(&h, m->dev, sizeof(DevMpx), hipMemcpyDeviceToHost, st));
	HIPCHK(ctx, hipStreamSynchronize(st));
	const uint64_t n = std::min<uint64_t>(h.n_log, max_rounds);
	*n_rounds = n;
	if (n) {
		if (path) {
			HIPCHK(ctx, hipMemcpyAsync(path, m->log_path, n * 4, hipMemcpyDeviceToHost, st));
		}
		if (tuples) {
			HIPCHK(ctx, hipMemcpyAsync(tuples, m->log_tuples, n * 8, hipMemcpyDeviceToHost, st));
		}
		if (intermediates) {
			HIPCHK(ctx, hipMemcpyAsync(intermediates, m->log_inter, n * 8, hipMemcpyDeviceToHost, st));
		}
		HIPCHK(ctx, hipStreamSynchronize(st));
	}
	return POLR_OK;
}

void polr_mpx_destroy(polr_mpx *m) {
	POLR_ENTRY();
	if (!m) {
		return;
	}
	hipSetDevice(m->ctx->device);
	if (m->dev) {
		hipFree(m->dev);
	}
	if (m->round_dev) {
		hipFree(m->round_dev);
	}
	if (m->prefix_dev) {
		hipFree(m->prefix_dev);
	}
	if (m->unit_size_dev) {
		hipFree(m->unit_size_dev);
	}
	if (m->ticket_dev) {
		hipFree(m->ticket_dev);
	}
	if (m->own_stream) {
		hipStreamSynchronize(m->own_stream);
		hipStreamDestroy(m->own_stream);
	}
	if (m->stats_dev) {
		hipFree(m->stats_dev);
	}
	if (m->counts_dev) {
		hipFree(m->counts_dev);
	}
	if (m->chunk_offsets_dev && m->chunk_offsets_owned) {
		hipFree(m->chunk_offsets_dev);
	}
	if (m->log_path) {
		hipFree(m->log_path);
	}
	if (m->log_tuples) {
		hipFree(m->log_tuples);
	}
	if (m->log_inter) {
		hipFree(m->log_inter);
	}
	if (m->done_host) {
		hipHostFree(m->done_host);
	}
	if (m->sync_dev) {
		hipFree(m->sync_dev);
	}
	if (m->stats_host) {
		hipHostFree(m->stats_host);
	}
	if (m->execs_dev) {
		hipFree(m->execs_dev);
	}
	if (m->share_dev) {
		hipFree(m->share_dev);
	}
	if (m->pool_dev) {
		hipFree(m->pool_dev);
	}
	for (auto e : m->ev_start) {
		hipEventDestroy(e);
	}
	for (auto e : m->ev_stop) {
		hipEventDestroy(e);
	}
	polr_ctx *ctx_ = m->ctx;
	delete m;
	polr_ctx_release(ctx_);
}

} // extern "C"
