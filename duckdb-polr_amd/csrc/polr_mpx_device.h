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
	uint64_t stage_out[POLR_MAX_PATHS][POLR_MAX_JOINS];
};

__device__ __forceinline__ uint64_t chunk_start(const DevMpx *m, uint64_t c) {
	if (m->chunk_offsets) {
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
                                                      uint32_t lane, bool coherent);
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

// m: the state to work on (LDS copy or HBM), mg: the HBM object (for stage_out)
__device__ __forceinline__ void polr_router_step_impl(DevMpx *m, DevMpx *mg, DevRound *round,
                                                      uint64_t *unit_prefix, uint32_t *unit_size_out,
                                                      unsigned long long *counts, uint32_t k, uint32_t resident_waves,
                                                      uint32_t lane, bool coherent) {
	// absorb the previous round's per-join outputs: what RunPath feeds AddNumIntermediates (:486-487).
	// One wave: lane s sums shard s of the k counters, a shuffle tree adds the shards, lane 0 routes.
	uint64_t s = 0;
	{
		// two counters per pass: lanes 0..31 sum the shards of counter j, lanes 32..63 those of j+1
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
			if (lane == 0) {
				s += v;
				mg->stage_out[m->last_path][j0] += v;
				if (j0 + 1 < k) {
					s += v_hi;
					mg->stage_out[m->last_path][j0 + 1] += v_hi;
				}
			}
		}
		if (lane != 0) {
			return;
		}
	}
	polr::MultiplexerCore &core = m->core;
	core.AddNumIntermediates(s);
	m->num_intermediates_total += s;

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
		begin = chunk_start(m, m->chunk_idx);
		tuples = chunk_start(m, m->chunk_idx + n) - begin;
		core.IncreaseInputTupleCount(tuples);
		if (core.num_cache_flushing_skips != polr::kIdxMax) {
			core.num_cache_flushing_skips -= n;
		}
		m->chunk_idx += n;
		path = core.current_path_idx;
	} else {
		const uint64_t c0 = chunk_start(m, m->chunk_idx);
		const uint64_t size = chunk_start(m, m->chunk_idx + 1) - c0;
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
				const uint64_t extra = chunk_start(m, m->chunk_idx + n) - chunk_start(m, m->chunk_idx);
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
	// unit size: one unit = what one wave takes per visit.  A wide stage-0 step eats 256 tuples, and
	// every busy workgroup costs an arrival atomic at the end of the launch, so never go below 256; a
	// table-sized round gives every resident wave a few units.
	const uint64_t gran = ((m->wide0_mask >> path) & 1u) ? 256 : 64; // tuples one stage-0 step takes
	uint64_t us = (tuples + resident_waves - 1) / resident_waves;
	us = ((us + gran - 1) / gran) * gran;
	us = us < gran ? gran : (us > 2048 ? 2048 : us);
	unit_size_out[0] = (uint32_t)us;
	unit_prefix[1] = (tuples + us - 1) / us;
	polr_publish_progress(m);
}

