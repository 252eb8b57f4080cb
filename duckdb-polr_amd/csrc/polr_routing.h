// duckdb-polr_amd/csrc/polr_routing.h -- the multiplexer's routing core, compiled for BOTH sides:
// the C++ host mirror (duckdb-polr_amd/host: PhysicalMultiplexer / RoutingStrategy classes) and the
// device routing step (polr_mpx_device.h).  One source => host and device decisions are the same double
// arithmetic (IEEE add/mul/div/round, compiled with -ffp-contract=off on both sides).
//
// Reference semantics (file:line in d-justen/duckdb-polr):
//   src/execution/operator/polr/routing_strategy.cpp:7-463   strategies
//   src/include/duckdb/execution/operator/polr/routing_strategy.hpp:15-213   their state
//   src/execution/operator/polr/physical_multiplexer.cpp:100-184   Execute / FinalizePathRun
// State is a plain struct so it can live in HBM; no allocation, no virtual dispatch.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define POLR_HD __host__ __device__
#else
#define POLR_HD
#endif

namespace polr {

static const uint32_t kMaxPaths = 32;
static const uint64_t kIdxMax = 0xFFFFFFFFFFFFFFFFull;

enum Routing : uint32_t {
	ALTERNATE = 0,
	ADAPTIVE_REINIT = 1,
	DYNAMIC = 2,
	INIT_ONCE = 3,
	OPPORTUNISTIC = 4,
	DEFAULT_PATH = 5,
	BACKPRESSURE = 6,
	EXPONENTIAL_BACKOFF = 7
};

// std::round for doubles (half away from zero), usable on the device
POLR_HD inline double round_half_away(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return ::round(x);
#else
	return __builtin_round(x);
#endif
}

struct RouteDecision {
	uint64_t offset;      // first tuple of the slice inside the input chunk
	uint64_t count;       // tuples routed
	uint64_t path;        // join order index
	uint64_t cache_skips; // whole chunks that follow on the same path without asking again
	bool have_more_output; // the input chunk has more slices (HAVE_MORE_OUTPUT)
};

struct MultiplexerCore {
	// configuration
	uint32_t routing;
	uint32_t path_count;
	double regret_budget;
	uint64_t init_tuple_count;
	uint64_t multiplier;
	// MultiplexerState
	double path_resistances[kMaxPaths];
	double historic_resistances[kMaxPaths];
	uint64_t input_tuple_count_per_path[kMaxPaths];
	uint32_t first_mpx_run;
	uint32_t alternate_mode;
	uint64_t num_intermediates_current_path;
	uint64_t num_tuples_processed;
	uint64_t current_path_tuple_count;
	uint64_t current_path_idx;
	uint64_t num_cache_flushing_skips;
	// RoutingStrategyState and subclasses
	uint64_t chunk_size, next_path_idx, next_tuple_count, chunk_offset, rs_cache_skips;
	uint64_t best_path_after_init, num_paths_initialized;
	uint32_t init_phase_done;
	uint32_t visited_paths[kMaxPaths];
	uint64_t window_offset, window_size, max_window_size;
	uint64_t eb_min_path;
	double eb_min_resistance;
	uint64_t remaining_tuples[kMaxPaths];
	int64_t remaining_tuples_diff[kMaxPaths];
	double path_weights[kMaxPaths];

	POLR_HD void Init(uint32_t routing_p, uint32_t path_count_p, double regret_budget_p, uint64_t init_tuple_count_p,
	                  uint64_t multiplier_p) {
		// zero everything, field by field (usable from device code)
		routing = routing_p;
		path_count = path_count_p;
		regret_budget = regret_budget_p;
		init_tuple_count = init_tuple_count_p;
		multiplier = multiplier_p;
		for (uint32_t i = 0; i < kMaxPaths; i++) {
			path_resistances[i] = 0;
			historic_resistances[i] = 0;
			input_tuple_count_per_path[i] = 0;
			visited_paths[i] = 0;
			remaining_tuples[i] = 0;
			remaining_tuples_diff[i] = 0;
			path_weights[i] = 0;
		}
		first_mpx_run = 1;
		alternate_mode = 0;
		num_intermediates_current_path = 0;
		num_tuples_processed = 0;
		current_path_tuple_count = 0;
		current_path_idx = 0;
		num_cache_flushing_skips = 0;
		chunk_size = next_path_idx = next_tuple_count = chunk_offset = rs_cache_skips = 0;
		best_path_after_init = num_paths_initialized = 0;
		init_phase_done = 0;
		window_offset = window_size = 0;
		// EXPONENTIAL_BACKOFF gets (idx_t)regret_budget as its window cap (physical_multiplexer.cpp:50-52)
		max_window_size = routing_p == EXPONENTIAL_BACKOFF ? (uint64_t)regret_budget_p : 0;
		eb_min_path = kIdxMax;
		eb_min_resistance = 1.7976931348623157e308;
	}

	// Init() for an object whose bytes are all zero already (a device router clears the struct with the whole wave:
	// Init's field-by-field zeroing is some 230 dependent LDS stores on its single routing lane)
	POLR_HD void InitAfterZero(uint32_t routing_p, uint32_t path_count_p, double regret_budget_p,
	                           uint64_t init_tuple_count_p, uint64_t multiplier_p) {
		routing = routing_p;
		path_count = path_count_p;
		regret_budget = regret_budget_p;
		init_tuple_count = init_tuple_count_p;
		multiplier = multiplier_p;
		first_mpx_run = 1;
		max_window_size = routing_p == EXPONENTIAL_BACKOFF ? (uint64_t)regret_budget_p : 0;
		eb_min_path = kIdxMax;
		eb_min_resistance = 1.7976931348623157e308;
	}

	// ---- reward -------------------------------------------------------------------------------
	POLR_HD void AddNumIntermediates(uint64_t n) {
		num_intermediates_current_path += n; // physical_multiplexer.cpp:181-184
	}
	POLR_HD void IncreaseInputTupleCount(uint64_t n) {
		current_path_tuple_count += n; // :127-130
	}
	// returns the intermediates of the run that was closed (what log_tuples_routed records)
	POLR_HD uint64_t FinalizePathRun() {
		// physical_multiplexer.cpp:132-174 with use_time_resistance = false
		input_tuple_count_per_path[current_path_idx] += current_path_tuple_count;
		num_tuples_processed += current_path_tuple_count;
		const uint64_t closed = num_intermediates_current_path;
		if (alternate_mode) {
			num_intermediates_current_path = 0;
			return closed;
		}
		const double constant_overhead = 0.5;
		double r = (double)num_intermediates_current_path / (double)current_path_tuple_count + constant_overhead;
		if (historic_resistances[current_path_idx] != 0) {
			const double smoothing = 0.5; // SMOOTHING_FACTOR, physical_multiplexer.hpp:25
			r = historic_resistances[current_path_idx] * smoothing + (1 - smoothing) * r;
		}
		path_resistances[current_path_idx] = r;
		historic_resistances[current_path_idx] = r;
		num_intermediates_current_path = 0;
		return closed;
	}

	// ---- strategies -----------------------------------------------------------------------------
	POLR_HD uint64_t ArgMin(double &mn) const {
		// first strictly smaller value wins (routing_strategy.cpp:38-46)
		mn = path_resistances[0];
		uint64_t idx = 0;
		for (uint32_t i = 1; i < path_count; i++) {
			if (path_resistances[i] < mn) {
				mn = path_resistances[i];
				idx = i;
			}
		}
		return idx;
	}

	POLR_HD int64_t FirstUninitialised() const {
		for (uint32_t i = 0; i < path_count; i++) {
			if (path_resistances[i] == 0) {
				return (int64_t)i;
			}
		}
		return -1;
	}

	POLR_HD uint64_t NextPathInitOnce() {
		// routing_strategy.cpp:55-82
		if (init_phase_done) {
			rs_cache_skips = kIdxMax;
			return best_path_after_init;
		}
		if (num_paths_initialized == path_count) {
			double mn;
			init_phase_done = 1;
			best_path_after_init = ArgMin(mn);
			return best_path_after_init;
		}
		return num_paths_initialized++;
	}

	POLR_HD uint64_t NextPathAdaptiveReinit() {
		// routing_strategy.cpp:94-180; the two tail recursions of the reference are loops here
		for (;;) {
			if (init_phase_done) {
				double mn;
				uint64_t best = ArgMin(mn);
				if (mn * 1.05 >= path_resistances[0]) {
					mn = path_resistances[0];
					best = 0;
				}
				if (window_offset == 0 || !visited_paths[best]) {
					visited_paths[best] = 1;
					double reinit_cost = 0;
					for (uint32_t i = 0; i < path_count; i++) {
						if (!visited_paths[i]) {
							reinit_cost += path_resistances[i] * (double)init_tuple_count;
						}
					}
					if (reinit_cost == 0) {
						for (uint32_t i = 0; i < path_count; i++) {
							visited_paths[i] = 0;
						}
						visited_paths[best] = 1;
						for (uint32_t i = 0; i < path_count; i++) {
							reinit_cost += path_resistances[i] * (double)init_tuple_count;
						}
					}
					const double tuples_before_reinit = reinit_cost / (regret_budget * mn);
					window_size = (uint64_t)tuples_before_reinit;
				}
				if (mn <= 0.525) { // RESISTANCE_TOLERANCE (routing_strategy.hpp:110)
					window_offset = 0;
					return best;
				}
				if (window_offset >= window_size) {
					window_offset = 0;
					for (uint32_t i = 0; i < path_count; i++) {
						if (!visited_paths[i]) {
							path_resistances[i] = 0;
						} else {
							visited_paths[i] = 0;
						}
					}
					init_phase_done = 0;
					continue;
				}
				return best;
			}
			const int64_t u = FirstUninitialised();
			if (u >= 0) {
				return (uint64_t)u;
			}
			init_phase_done = 1;
		}
	}

	POLR_HD uint64_t NextPathExponentialBackoff() {
		// routing_strategy.cpp:198-252
		for (;;) {
			if (init_phase_done) {
				double mn;
				const uint64_t cur = ArgMin(mn);
				if (window_offset == 0) {
					if (window_size == 0) {
						window_size = 1;
					} else if (cur == eb_min_path || mn * 1.1 >= path_resistances[eb_min_path]) {
						const uint64_t doubled = window_size * 2;
						window_size = max_window_size < doubled ? max_window_size : doubled;
					} else {
						window_size = 1;
					}
				} else if (window_offset >= window_size) {
					window_offset = 0;
					init_phase_done = 0;
					for (uint32_t i = 0; i < path_count; i++) {
						if (i != eb_min_path) {
							path_resistances[i] = 0;
						}
					}
					continue;
				}
				eb_min_resistance = mn;
				eb_min_path = cur;
				return cur;
			}
			const int64_t u = FirstUninitialised();
			if (u >= 0) {
				return (uint64_t)u;
			}
			init_phase_done = 1;
		}
	}

	// CalculateJoinPathWeights (routing_strategy.cpp:267-316): bottom-up bounded regret.  The
	// reference walks a std::multimap<double, idx_t> backwards; equal keys keep insertion order, so a
	// stable insertion sort gives the same sequence.
	POLR_HD void CalculateJoinPathWeights() {
		double key[kMaxPaths];
		uint32_t idx[kMaxPaths];
		const uint32_t n = path_count;
		for (uint32_t i = 0; i < n; i++) {
			uint32_t pos = i;
			while (pos > 0 && key[pos - 1] > path_resistances[i]) {
				key[pos] = key[pos - 1];
				idx[pos] = idx[pos - 1];
				pos--;
			}
			key[pos] = path_resistances[i];
			idx[pos] = i;
		}
		double cost_bottom = key[n - 1];
		for (int32_t it = (int32_t)n - 2; it >= 0; it--) {
			const double cost_next = key[it];
			const double next_rounded = round_half_away(cost_next / 0.001) * 0.001;
			const double bottom_rounded = round_half_away(cost_bottom / 0.001) * 0.001;
			if (next_rounded == bottom_rounded) {
				cost_bottom += 0.001;
			}
			double cost_target = cost_next * (1 + regret_budget);
			const double cost_avg = (cost_next + cost_bottom) / 2;
			if (cost_target >= cost_avg) {
				cost_target = 0.6 * cost_next + 0.4 * cost_bottom;
			}
			const double weight_bottom = (cost_next - cost_target) / (cost_next - cost_bottom);
			for (int32_t it2 = (int32_t)n - 1; it2 > it; it2--) {
				path_weights[idx[it2]] *= weight_bottom;
			}
			path_weights[idx[it]] = 1 - weight_bottom;
			cost_bottom = cost_target;
		}
	}

	POLR_HD uint64_t MaxRemaining(uint64_t &max_remaining) const {
		max_remaining = remaining_tuples[0];
		uint64_t idx = 0;
		for (uint32_t i = 1; i < path_count; i++) {
			if (remaining_tuples[i] > max_remaining) {
				max_remaining = remaining_tuples[i];
				idx = i;
			}
		}
		return idx;
	}

	POLR_HD uint64_t NextPathDynamic() {
		// routing_strategy.cpp:318-406
		for (;;) {
			if (init_phase_done) {
				uint64_t max_remaining;
				const uint64_t max_idx = MaxRemaining(max_remaining);
				if (max_remaining > 0) {
					return max_idx;
				}
				for (uint32_t i = 0; i < path_count; i++) {
					path_weights[i] = 1;
				}
				CalculateJoinPathWeights();
				const uint64_t input_tuples = chunk_size * multiplier - chunk_offset;
				uint64_t sum = 0;
				for (uint32_t i = 0; i < path_count; i++) {
					// `int remaining_tuples = diff + std::round(weight * input_tuples)` (:338)
					const int remaining =
					    (int)((double)remaining_tuples_diff[i] + round_half_away(path_weights[i] * (double)input_tuples));
					if (remaining < 0) {
						remaining_tuples_diff[i] += (int64_t)remaining_tuples[i];
						remaining_tuples[i] = 0;
					} else {
						remaining_tuples[i] = (uint64_t)remaining;
						remaining_tuples_diff[i] = 0;
					}
					sum += remaining_tuples[i];
				}
				uint64_t sum_after = 0;
				for (uint32_t i = 0; i < path_count; i++) {
					remaining_tuples[i] =
					    (uint64_t)round_half_away((double)remaining_tuples[i] / (double)sum * (double)input_tuples);
					if (remaining_tuples[i] < 64) {
						remaining_tuples_diff[i] = (int64_t)remaining_tuples[i];
						remaining_tuples[i] = 0;
					}
					sum_after += remaining_tuples[i];
				}
				if (sum_after != input_tuples) {
					uint64_t control_sum = 0, max_normalized = 0, max_normalized_idx = 0;
					for (uint32_t i = 0; i < path_count; i++) {
						if (remaining_tuples[i] > 0) {
							const uint64_t normalized = (uint64_t)round_half_away(
							    (double)remaining_tuples[i] / (double)sum_after * (double)input_tuples);
							remaining_tuples_diff[i] -= (int64_t)(normalized - remaining_tuples[i]);
							remaining_tuples[i] = normalized;
							control_sum += normalized;
							if (normalized > max_normalized) {
								max_normalized = normalized;
								max_normalized_idx = i;
							}
						}
					}
					if (control_sum != input_tuples) {
						remaining_tuples[max_normalized_idx] -= control_sum - (uint64_t)(int)input_tuples;
					}
				}
				continue;
			}
			const int64_t u = FirstUninitialised();
			if (u >= 0) {
				return (uint64_t)u;
			}
			init_phase_done = 1;
		}
	}

	POLR_HD uint64_t DetermineNextPath() {
		switch (routing) {
		case OPPORTUNISTIC: {
			double mn;
			return ArgMin(mn); // routing_strategy.cpp:35-49
		}
		case INIT_ONCE:
			return NextPathInitOnce();
		case ADAPTIVE_REINIT:
			return NextPathAdaptiveReinit();
		case EXPONENTIAL_BACKOFF:
			return NextPathExponentialBackoff();
		case DYNAMIC:
			return NextPathDynamic();
		default: // DEFAULT_PATH, BACKPRESSURE (routing_strategy.cpp:454-457, physical_multiplexer.cpp:47-49)
			rs_cache_skips = kIdxMax;
			return 0;
		}
	}

	POLR_HD uint64_t MinU64(uint64_t a, uint64_t b) const {
		return a < b ? a : b;
	}

	POLR_HD uint64_t DetermineNextTupleCount() {
		switch (routing) {
		case INIT_ONCE: // routing_strategy.cpp:84-92
			if (init_phase_done) {
				return chunk_size - chunk_offset;
			}
			return MinU64(init_tuple_count, chunk_size - chunk_offset);
		case ADAPTIVE_REINIT: // :182-196
			if (init_phase_done) {
				if (window_offset < window_size) {
					rs_cache_skips = (uint64_t)round_half_away((double)window_size / (double)chunk_size);
					window_offset += window_size;
				} else {
					rs_cache_skips = 0;
				}
				return chunk_size - chunk_offset;
			}
			rs_cache_skips = 0;
			return MinU64(init_tuple_count, chunk_size - chunk_offset);
		case EXPONENTIAL_BACKOFF: // :254-265
			if (init_phase_done) {
				rs_cache_skips = window_size;
				window_offset += window_size;
				return chunk_size - chunk_offset;
			}
			rs_cache_skips = 0;
			return MinU64(init_tuple_count, chunk_size - chunk_offset);
		case DYNAMIC: { // :408-438
			rs_cache_skips = 0;
			if (init_phase_done) {
				uint64_t max_remaining;
				const uint64_t max_idx = MaxRemaining(max_remaining);
				if (max_remaining > 0) {
					const uint64_t remaining_input = chunk_size - chunk_offset;
					if (max_remaining > remaining_input) {
						rs_cache_skips = (max_remaining - remaining_input) / chunk_size;
						remaining_tuples[max_idx] -= rs_cache_skips * chunk_size + remaining_input;
						return remaining_input;
					}
					remaining_tuples[max_idx] = 0;
					return max_remaining;
				}
			}
			return MinU64(init_tuple_count, chunk_size - chunk_offset);
		}
		default: // OPPORTUNISTIC (:51-53), DEFAULT_PATH (:459-461)
			return chunk_size;
		}
	}

	// PhysicalMultiplexer::Execute (physical_multiplexer.cpp:100-121) + RoutingStrategy::Route
	// (routing_strategy.hpp:47-53) + SelectTuples (routing_strategy.cpp:7-33) +
	// AlternateRoutingStrategy::Route (:440-452).  *closed_run receives the intermediates of the run
	// FinalizePathRun just closed (valid when the return flag `finalized` is set).
	POLR_HD RouteDecision Execute(uint64_t input_size, bool *finalized, uint64_t *closed_run) {
		if (!first_mpx_run) {
			*closed_run = FinalizePathRun();
			*finalized = true;
		} else {
			*finalized = false;
			first_mpx_run = 0;
			if (routing == ALTERNATE) {
				alternate_mode = 1;
			}
		}
		RouteDecision d;
		d.offset = 0;
		if (routing == ALTERNATE) {
			next_path_idx = next_tuple_count == 0 ? 0 : (next_path_idx + 1) % path_count;
			next_tuple_count = input_size;
			d.have_more_output = next_path_idx != (uint64_t)path_count - 1;
		} else {
			chunk_size = input_size;
			next_path_idx = DetermineNextPath();
			next_tuple_count = DetermineNextTupleCount();
			if (next_tuple_count == input_size) {
				d.have_more_output = false; // chunk.Reference(input)
			} else {
				d.offset = chunk_offset;
				if (chunk_offset + next_tuple_count == input_size) {
					chunk_offset = 0;
					d.have_more_output = false;
				} else {
					chunk_offset += next_tuple_count;
					d.have_more_output = true;
				}
			}
		}
		current_path_tuple_count = next_tuple_count;
		current_path_idx = next_path_idx;
		num_cache_flushing_skips = rs_cache_skips;
		d.count = next_tuple_count;
		d.path = current_path_idx;
		d.cache_skips = num_cache_flushing_skips;
		return d;
	}
};

} // namespace polr
