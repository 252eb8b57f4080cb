// duckdb-polr_amd/csrc/polr_mpx_device.h -- device-resident multiplexer state and the routing step.
// Included by polr_mpx.hip (init / finish kernels, host API) and by the path kernel (polr_probe.hip):
// the LAST workgroup of a path-kernel launch runs polr_router_step() itself, so one launch = probe
// round r + route round r+1 and the host never sits between two routing decisions.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/polr_hip.h"
#include "polr_device.h"
#include "polr_routing.h"

#define POLR_SLOTS 4 // rounds one executor can have in flight in a one-launch run (see ResidentSync)

struct DevMpx {
	polr::MultiplexerCore core;
	uint64_t chunk_idx, chunk_end;
	uint64_t n_tuples, n_chunks;
	uint32_t chunk_size;
	uint32_t done;
	const uint64_t *chunk_offsets; // nullptr: fixed chunk_size chunks
	uint64_t num_intermediates_total;
	uint64_t num_rounds;
	uint32_t log_enabled, pad;
	uint64_t max_log, n_log;
	uint32_t *log_path;
	uint64_t *log_tuples;
	uint64_t *log_inter;
	uint64_t last_path; // path of the round whose counters are still to be absorbed
	uint32_t wide0_mask; // bit p: stage 0 of join order p takes the wide (256 tuples per step) path
	uint32_t pad2;
	// host-visible progress words (pinned, mapped host memory): [0] = routing steps completed, [1] = done
	volatile uint32_t *progress;
	uint32_t steps_done, pad3;
	// what a fresh MultiplexerState is built from (a resident run can reset itself)
	polr_mpx_config cfg;
	uint32_t n_paths;
	// bookkeeping of resident runs, carried from run to run so that a run starts without extra round trips:
	// res_valid: res_target[] are the current sums of the arrival counters and the counter banks beyond 0 are empty
	// (false after a run that was given up -- the next one re-reads and drops)
	uint32_t res_valid;
	unsigned long long res_target[POLR_SLOTS];
	uint64_t stage_out[POLR_MAX_PATHS][POLR_MAX_JOINS];
};

// window of the chunk-offset array kept in LDS by a resident router (the boundaries a routing step needs
// are then LDS reads instead of a chain of dependent HBM loads)
#define POLR_OFFS_CACHE 2048
struct OffsCache {
	uint64_t base, n; // offsets [base, base + n) are cached
	uint64_t *data;
};

__device__ __forceinline__ uint64_t chunk_start(const DevMpx *m, uint64_t c, const OffsCache *oc = nullptr) {
	if (m->chunk_offsets) {
		if (oc && c - oc->base < oc->n) {
			return oc->data[c - oc->base];
		}
		return m->chunk_offsets[c];
	}
	const uint64_t s = c * (uint64_t)m->chunk_size;
	return s < m->n_tuples ? s : m->n_tuples;
}

__device__ __forceinline__ void log_round(DevMpx *m, uint64_t path, uint64_t tuples, uint64_t inter) {
	m->num_rounds++;
	if (m->log_enabled && m->n_log < m->max_log) {
		m->log_path[m->n_log] = (uint32_t)path;
		m->log_tuples[m->n_log] = tuples;
		m->log_inter[m->n_log] = inter;
		m->n_log++;
	}
}

// what a self-routing launch of the path kernel needs (mpx == nullptr: plain launch)
struct SelfRoute {
	DevMpx *mpx;
	DevRound *rounds_base;   // [2] double-buffered round descriptor
	uint64_t *prefix_base;   // [2][2]
	uint32_t *unit_base;     // [2]
	uint32_t *ticket;        // arrivals of busy workgroups
	uint32_t iter;           // launch index: descriptor slot = iter & 1
	uint32_t resident_waves;
	unsigned long long *stamps; // diagnostic builds only (POLR_DIAG_STAMPS), else nullptr
};

// tell the host how far the device-side routing has got (it throttles its launch look-ahead on this)
__device__ __forceinline__ void polr_publish_progress(DevMpx *m) {
	m->steps_done++;
	if (m->progress) {
		// plain stores to mapped host memory: they land in order of issue soon enough; a stale read on the
		// host costs at most one extra (empty) launch, never correctness -- no system-scope fence on the
		// critical path of every routing step
		m->progress[1] = m->done;
		m->progress[0] = m->steps_done;
	}
}

// one routing decision
__device__ __forceinline__ void polr_router_step_impl(DevMpx *m, DevMpx *mg, DevRound *round,
                                                      uint64_t *unit_prefix, uint32_t *unit_size_out,
                                                      unsigned long long *counts, uint32_t k, uint32_t resident_waves,
                                                      uint32_t lane, bool coherent, const OffsCache *oc = nullptr,
                                                      bool discard = false);
// Executed by ONE full wave (lane = 0..63).  `coherent`: the counters were just written by other
// workgroups of the same launch -> read them with device-scope atomics (exchange with 0).
__device__ __forceinline__ void polr_router_step(DevMpx *mg, DevRound *round, uint64_t *unit_prefix,
                                                 uint32_t *unit_size_out, unsigned long long *counts, uint32_t k,
                                                 uint32_t resident_waves, uint32_t lane, bool coherent,
                                                 uint32_t *scratch_lds) {
	// The routing arithmetic touches a few dozen fields of the ~2 KB state one after the other; in HBM that
	// is a chain of dependent loads.  Stage the state in LDS (one cooperative copy in, one out).
	// (stage_out, the big per-(path, position) statistics block at the end of DevMpx, stays in HBM: only
	// k of its cells are touched per step.)
	constexpr uint32_t kHot = offsetof(DevMpx, stage_out) / 4;
	static_assert(offsetof(DevMpx, stage_out) % 4 == 0, "hot part must be whole dwords");
	DevMpx *m = mg;
	if (scratch_lds) {
		const uint32_t *src = (const uint32_t *)mg;
		for (uint32_t i = lane; i < kHot; i += 64) {
			scratch_lds[i] = src[i];
		}
		m = (DevMpx *)scratch_lds;
	}
	polr_router_step_impl(m, mg, round, unit_prefix, unit_size_out, counts, k, resident_waves, lane, coherent);
	if (scratch_lds) {
		uint32_t *dst = (uint32_t *)mg;
		for (uint32_t i = lane; i < kHot; i += 64) {
			dst[i] = scratch_lds[i];
		}
	}
}

__device__ __forceinline__ void polr_router_route(DevMpx *m, DevRound *round, uint64_t *unit_prefix,
                                                  uint32_t *unit_size_out, uint32_t resident_waves,
                                                  const OffsCache *oc, bool size_units = true);

// unit = what one wave takes per visit: every wave gets work, never less than `gran` tuples, never more than 2048
__device__ __forceinline__ void polr_size_units(uint64_t tuples, uint32_t waves, uint64_t gran, uint32_t *unit_size_out,
                                                uint64_t *unit_prefix) {
	// whole passes over the waves: a round of 1.18 x (waves x 2048) tuples is cut into 2 x waves units of ~1200, not
	// into 1.18 x waves units of 2048 (where a fifth of the waves would do double duty while the rest wait)
	const uint64_t per_pass = (uint64_t)waves * 2048u;
	const uint64_t passes = tuples ? (tuples + per_pass - 1) / per_pass : 1;
	uint64_t us = (tuples + waves * passes - 1) / (waves * passes);
	us = ((us + gran - 1) / gran) * gran;
	us = us < gran ? gran : (us > 2048 ? 2048 : us);
	unit_size_out[0] = (uint32_t)us;
	unit_prefix[1] = (tuples + us - 1) / us;
}

// Absorb a counter bank: what RunPath feeds AddNumIntermediates (polar_pipeline_executor.cpp:486-487).  One
// wave: lanes 0..31 sum the shards of counter j, lanes 32..63 those of j+1, a shuffle tree adds the shards.
// Returns the sum of all k counters in lane 0 (0 elsewhere, and 0 when `discard`).
// m: the state to work on (LDS copy or HBM), mg: the HBM object (for stage_out)
__device__ __forceinline__ uint64_t polr_router_step_absorb(DevMpx *m, DevMpx *mg, unsigned long long *counts,
                                                            uint32_t k, uint32_t lane, bool coherent, bool discard) {
	uint64_t s = 0;
	for (uint32_t j0 = 0; j0 < k; j0 += 2) {
		const uint32_t j = j0 + (lane >> 5);
		const uint32_t shard = lane & 31u;
		unsigned long long v = 0;
		if (j < k) {
			if (coherent) {
				v = atomicExch(&counts[(uint64_t)shard * k + j], 0ull);
			} else {
				v = counts[(uint64_t)shard * k + j];
				counts[(uint64_t)shard * k + j] = 0;
			}
		}
		for (int d = 16; d > 0; d >>= 1) {
			v += __shfl_down(v, d, 32);
		}
		const unsigned long long v_hi = __shfl(v, 32, 64);
		if (lane == 0 && !discard) {
			// (fire-and-forget adds: a load-add-store here would put an HBM round trip on the routing path)
			s += v;
			if (v) {
				atomicAdd((unsigned long long *)&mg->stage_out[m->last_path][j0], v);
			}
			if (j0 + 1 < k) {
				s += v_hi;
				if (v_hi) {
					atomicAdd((unsigned long long *)&mg->stage_out[m->last_path][j0 + 1], v_hi);
				}
			}
		}
	}
	return s;
}

__device__ __forceinline__ void polr_router_step_impl(DevMpx *m, DevMpx *mg, DevRound *round,
                                                      uint64_t *unit_prefix, uint32_t *unit_size_out,
                                                      unsigned long long *counts, uint32_t k, uint32_t resident_waves,
                                                      uint32_t lane, bool coherent, const OffsCache *oc, bool discard) {
	const uint64_t s = polr_router_step_absorb(m, mg, counts, k, lane, coherent, discard);
	if (lane != 0) {
		return;
	}
	m->core.AddNumIntermediates(s);
	m->num_intermediates_total += s;
	polr_router_route(m, round, unit_prefix, unit_size_out, resident_waves, oc);
}

// The routing decision proper (lane 0 of the caller): FinalizePathRun of the run whose intermediates have just
// been added, Route, fold the routing window, write the round.  m: the state to work on.
__device__ __forceinline__ void polr_router_route(DevMpx *m, DevRound *round, uint64_t *unit_prefix,
                                                  uint32_t *unit_size_out, uint32_t resident_waves,
                                                  const OffsCache *oc, bool size_units) {
	polr::MultiplexerCore &core = m->core;
	round->begin = 0;
	round->count = 0;
	round->path = 0;
	round->emit = 0;
	unit_prefix[0] = 0;
	unit_prefix[1] = 0;
	unit_size_out[0] = 64;
	if (m->chunk_idx >= m->chunk_end) {
		m->done = 1;
		polr_publish_progress(m);
		return;
	}
	uint64_t begin, tuples, path;
	if (core.num_cache_flushing_skips > 0) {
		// the window continues (a previous run() ended inside it): whole chunks bypass routing
		const uint64_t left = m->chunk_end - m->chunk_idx;
		const uint64_t n = core.num_cache_flushing_skips < left ? core.num_cache_flushing_skips : left;
		begin = chunk_start(m, m->chunk_idx, oc);
		tuples = chunk_start(m, m->chunk_idx + n, oc) - begin;
		core.IncreaseInputTupleCount(tuples);
		if (core.num_cache_flushing_skips != polr::kIdxMax) {
			core.num_cache_flushing_skips -= n;
		}
		m->chunk_idx += n;
		path = core.current_path_idx;
	} else {
		const uint64_t c0 = chunk_start(m, m->chunk_idx, oc);
		const uint64_t size = chunk_start(m, m->chunk_idx + 1, oc) - c0;
		const uint64_t prev_path = core.current_path_idx;
		const uint64_t prev_tuples = core.current_path_tuple_count;
		bool finalized;
		uint64_t closed = 0;
		const polr::RouteDecision d = core.Execute(size, &finalized, &closed);
		if (finalized) {
			log_round(m, prev_path, prev_tuples, closed);
		}
		begin = c0 + d.offset;
		tuples = d.count;
		path = d.path;
		if (!d.have_more_output) {
			m->chunk_idx++;
			if (core.num_cache_flushing_skips > 0 && m->chunk_idx < m->chunk_end) {
				const uint64_t left = m->chunk_end - m->chunk_idx;
				const uint64_t n = core.num_cache_flushing_skips < left ? core.num_cache_flushing_skips : left;
				const uint64_t extra = chunk_start(m, m->chunk_idx + n, oc) - chunk_start(m, m->chunk_idx, oc);
				core.IncreaseInputTupleCount(extra);
				if (core.num_cache_flushing_skips != polr::kIdxMax) {
					core.num_cache_flushing_skips -= n;
				}
				m->chunk_idx += n;
				tuples += extra;
			}
		}
	}
	m->last_path = path;
	round->begin = begin;
	round->count = tuples;
	round->path = (uint32_t)path;
	// ALTERNATE forwards only path 0's output (polar_pipeline_executor.cpp:445-447,514-523)
	round->emit = (core.routing != polr::ALTERNATE || path == 0) ? 1u : 0u;
	// unit size: one unit = what one wave takes per visit; a table-sized round gives every resident wave a
	// few units.  Per-round launches: every busy workgroup costs an arrival atomic at the end of the launch, and a wide
	// stage-0 step eats 256 tuples, so units are multiples of 256.  Resident run (the caller passes its
	// chunk-offset cache): waves are there anyway, a small round is spread 64 tuples per wave -- the dependent-load chain of
	// a step is the same for 64 and for 256 tuples, so more waves in parallel is strictly faster.
	if (size_units) { // (the pool cuts its rounds itself: polr_pool_size_units -- four 64-bit divisions saved per step)
		const uint64_t gran = (oc == nullptr && ((m->wide0_mask >> path) & 1u)) ? 256 : 64;
		polr_size_units(tuples, resident_waves, gran, unit_size_out, unit_prefix);
	}
	polr_publish_progress(m);
}


// ==== one-launch runs: what a router and the probe pool share per executor ==========================
// (protocol: polr_pool_device.h)
#define POLR_RES_TIMEOUT_TICKS 400000000ull // 4 s of the 100 MHz wall clock: a wait this long is a lost run

// Round SLOTS per executor: while the pool probes round r the router may already have published the rounds after it
// whose decisions cannot depend on the intermediates still outstanding (the exploration rounds of an init phase,
// ALTERNATE): up to POLR_SLOTS dependent-latency rounds overlap.  Each slot has its own counter bank and arrivals.
#define POLR_ARRIVE_SHARDS 8 // arrival counters and counter banks are sharded (ring & 7), a cache line each
struct ResidentSync {
	struct {
		unsigned long long v, pad[7];
	} arrived[POLR_SLOTS][POLR_ARRIVE_SHARDS]; // per slot; monotonic across runs; the router sums the shards
};

#define POLR_MORE_RANGES 7 // polr_mpx_run_resident_ranges: up to 8 ranges per executor
struct ResidentExec {
	DevMpx *mpx;
	ResidentSync *sync;
	unsigned long long *counts;
	uint64_t chunk_begin, chunk_end;
	const uint64_t *chunk_offsets;
	uint64_t n_chunks, n_tuples;
	uint32_t flags; // POLR_RUN_RESET | POLR_RUN_FINISH
	uint32_t pad;
	polr_mpx_stats *stats_out; // POLR_RUN_FINISH: where the closing statistics go (pinned host memory)
	// morsel mode (morsel_cursor != nullptr): the executors of the run share the chunk range up to morsel_end and
	// pull it `morsel_chunks` chunks at a time from the cursor, as the reference's worker threads pull morsels from
	// the parallel scan state; chunk_begin / chunk_end are ignored
	unsigned long long *morsel_cursor;
	uint64_t morsel_end;
	uint32_t morsel_chunks;
	uint32_t path_plus1; // BACKPRESSURE: this executor sends everything down join order path_plus1 - 1 (0: as routed)
	// further chunk ranges of this executor, routed one after the other behind [chunk_begin, chunk_end) with the same
	// multiplexer state (polr_mpx_run_resident_ranges: a static list of morsels per executor)
	uint32_t n_more;
	uint32_t pad2;
	uint64_t more_begin[POLR_MORE_RANGES], more_end[POLR_MORE_RANGES];
};

#define POLR_RES_HOT_DWORDS (offsetof(DevMpx, stage_out) / 4)
#define POLR_RES_ROUTER_DWORDS (POLR_RES_HOT_DWORDS + 32)

// PushFinalize's closing FinalizePathRun (polar_pipeline_executor.cpp:150-151); lane 0 of the caller,
// counters already absorbed.  m: the state being worked on (LDS copy or HBM)
__device__ __forceinline__ void polr_close_run(DevMpx *m) {
	polr::MultiplexerCore &core = m->core;
	if (!core.first_mpx_run) {
		const uint64_t path = core.current_path_idx, tuples = core.current_path_tuple_count;
		const uint64_t closed = core.FinalizePathRun();
		log_round(m, path, tuples, closed);
		// a finalized run must not be finalized twice if the caller keeps routing afterwards
		core.current_path_tuple_count = 0;
	}
}

// one wave; m: hot state (LDS copy or HBM), mg: the HBM object (stage_out)
__device__ __forceinline__ void polr_write_stats(const DevMpx *m, DevMpx *mg, polr_mpx_stats *stats, uint32_t lane) {
	const polr::MultiplexerCore &core = m->core;
	if (lane == 0) {
		stats->num_tuples_processed = core.num_tuples_processed;
		stats->num_intermediates = m->num_intermediates_total;
		stats->num_rounds = m->num_rounds;
	}
	for (uint32_t i = lane; i < POLR_MAX_PATHS; i += 64) {
		stats->input_tuple_count_per_path[i] = i < core.path_count ? core.input_tuple_count_per_path[i] : 0;
		stats->path_resistances[i] = i < core.path_count ? core.path_resistances[i] : 0;
	}
	// (only the cells of real join orders can be non-zero: one dependent round trip instead of four)
	const uint32_t live = core.path_count * POLR_MAX_JOINS;
	for (uint32_t i = lane; i < POLR_MAX_PATHS * POLR_MAX_JOINS; i += 64) {
		unsigned long long v = 0;
		if (i < live) {
			v = __hip_atomic_load((unsigned long long *)&mg->stage_out[i / POLR_MAX_JOINS][i % POLR_MAX_JOINS],
			                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		stats->stage_out[i / POLR_MAX_JOINS][i % POLR_MAX_JOINS] = v;
	}
}

// (re)load the LDS window of chunk boundaries starting at chunk `from`; full wave
__device__ __forceinline__ void polr_offs_cache_fill(OffsCache &oc, const ResidentExec &x, uint64_t from,
                                                     uint32_t cap, uint32_t lane) {
	uint64_t n = x.n_chunks + 1 - from; // entries [from, n_chunks] exist
	if (from > x.n_chunks) {
		n = 0;
	}
	if (n > cap) {
		n = cap;
	}
	// (never beyond the executor's own range: the boundary after its last chunk is the last entry it reads)
	if (!x.morsel_cursor && x.chunk_end >= from && n > x.chunk_end - from + 1) {
		n = x.chunk_end - from + 1;
	}
	__builtin_amdgcn_wave_barrier();
	// 8 independent loads per lane in flight (a plain loop waits for every load before it issues the next one)
	for (uint64_t i0 = 0; i0 < n; i0 += 64 * 8) {
		uint64_t v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const uint64_t i = i0 + (uint64_t)j * 64 + lane;
			v[j] = i < n ? x.chunk_offsets[from + i] : 0;
		}
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const uint64_t i = i0 + (uint64_t)j * 64 + lane;
			if (i < n) {
				oc.data[i] = v[j];
			}
		}
	}
	__builtin_amdgcn_wave_barrier();
	oc.base = from;
	oc.n = n;
}

// May the decision AFTER the round just routed be made before that round's intermediates are known?
// Yes exactly when it is an exploration round of an init phase (or ALTERNATE): FinalizePathRun will store a
// non-zero resistance for the current path whatever the intermediates are (r = I/T + 0.5 > 0), so
//   INIT_ONCE:                        the next path is num_paths_initialized (routing_strategy.cpp:55-82),
//   ADAPTIVE_REINIT / EXP. BACKOFF:   the next path is the first one whose resistance is still 0, other than the
//                                     current one (FirstUninitialised, :94-180 / :198-252),
// and the tuple count is min(init_tuple_count, rest of the chunk) (:84-92, :182-196, :254-265) -- none of which
// reads a reward.  ALTERNATE cycles through the paths per chunk (:440-452).  Everything else: no.
__device__ __forceinline__ bool polr_can_speculate(const volatile polr::MultiplexerCore &core) {
	if (core.num_cache_flushing_skips != 0) {
		return false; // a routing window is open
	}
	switch (core.routing) {
	case polr::ALTERNATE:
		return true;
	case polr::INIT_ONCE:
		return !core.init_phase_done && core.num_paths_initialized < core.path_count;
	case polr::ADAPTIVE_REINIT:
	case polr::EXPONENTIAL_BACKOFF: {
		if (core.init_phase_done) {
			return false;
		}
		for (uint32_t i = 0; i < core.path_count; i++) {
			if (i != core.current_path_idx && core.path_resistances[i] == 0) {
				return true;
			}
		}
		return false;
	}
	default:
		return false;
	}
}
