#include "polar_pipeline_executor.hpp"

#include <algorithm>
#include <limits>
#include <sstream>

namespace duckdb_polr {

static void Check(polr_ctx *ctx, int rc, const char *what) {
	if (rc != POLR_OK) {
		throw InternalException(string(what) + ": " + polr_last_error(ctx));
	}
}

POLARPipelineExecutor::POLARPipelineExecutor(ClientContext &context_p, POLARConfig &polar_p, polr_ctx *ctx_p,
                                             polr_pipeline *pipe_p, idx_t n_tuples_p, vector<idx_t> chunk_offsets_p)
    : context(context_p), polar(polar_p), ctx(ctx_p), pipe(pipe_p), n_tuples(n_tuples_p),
      chunk_offsets(std::move(chunk_offsets_p)) {
	if (!ctx || !pipe) {
		throw InternalException("POLARPipelineExecutor needs a device pipeline: there is no host probe path");
	}
	n_chunks = chunk_offsets.empty() ? (n_tuples + STANDARD_VECTOR_SIZE - 1) / STANDARD_VECTOR_SIZE
	                                 : chunk_offsets.size() - 1;
	alternate = polar.multiplexer->routing == MultiplexerRouting::ALTERNATE;
}

POLARPipelineExecutor::~POLARPipelineExecutor() {
}

idx_t POLARPipelineExecutor::ChunkStart(idx_t c) const {
	if (!chunk_offsets.empty()) {
		return chunk_offsets[c];
	}
	return std::min<idx_t>(c * STANDARD_VECTOR_SIZE, n_tuples);
}

bool POLARPipelineExecutor::Execute(RoutingPlacement placement, polr_out *out) {
	num_intermediates_produced = 0;
	intermediates_per_round.clear();
	path_per_round.clear();
	tuples_per_round.clear();
	if (placement == RoutingPlacement::HOST_ROUTED) {
		ExecuteHostRouted(out);
	} else {
		ExecuteDeviceRouted(out, placement == RoutingPlacement::DEVICE_RESIDENT);
	}
	return true;
}

// The reference's loop (polar_pipeline_executor.cpp:255-425 + RunPath :427-538) with each path run
// turned into one polr_probe_rounds call.
void POLARPipelineExecutor::ExecuteHostRouted(polr_out *out) {
	auto &multiplexer = *polar.multiplexer;
	ExecutionContext exec_context(context, thread);
	const bool log_was = context.config.log_tuples_routed;
	context.config.log_tuples_routed = true; // the per-round trace is part of this executor's result
	auto mpx_state = multiplexer.GetOperatorState(exec_context);
	const idx_t k = polar.joins.size();
	vector<LogicalType> no_columns;
	DataChunk source_chunk, mpx_output_chunk;
	source_chunk.InitializeEmpty(no_columns);
	mpx_output_chunk.InitializeEmpty(no_columns);
	vector<uint64_t> counts(k);

	idx_t c = 0;
	bool mpx_in_process = false; // in_process_operators holds the multiplexer (HAVE_MORE_OUTPUT)
	while (c < n_chunks) {
		const idx_t c0 = ChunkStart(c);
		const idx_t size = ChunkStart(c + 1) - c0;
		if (size == 0) {
			c++;
			continue;
		}
		idx_t &cache_skips_left = multiplexer.GetNumCacheFlushingSkips(*mpx_state);
		idx_t begin, tuples;
		if (cache_skips_left > 0 && !mpx_in_process) {
			// chunks that bypass routing (:322-329): IncreaseInputTupleCount + RunPath on the current path
			idx_t n = std::min<idx_t>(cache_skips_left, n_chunks - c);
			begin = c0;
			tuples = ChunkStart(c + n) - c0;
			multiplexer.IncreaseInputTupleCount(*mpx_state, tuples);
			if (cache_skips_left != std::numeric_limits<idx_t>::max()) {
				cache_skips_left -= n;
			}
			c += n;
		} else {
			source_chunk.SetCardinality(size);
			auto result = multiplexer.Execute(exec_context, source_chunk, mpx_output_chunk, *multiplexer.op_state,
			                                  *mpx_state);
			const auto &core = multiplexer.Core(*mpx_state);
			tuples = mpx_output_chunk.size();
			// where SelectTuples cut the slice (routing_strategy.cpp:7-33): a whole-chunk Reference starts at
			// 0; otherwise the slice ends at the advanced chunk_offset, or at the end of the chunk
			if (tuples == size) {
				begin = c0;
			} else if (result == OperatorResultType::HAVE_MORE_OUTPUT) {
				begin = c0 + core.chunk_offset - tuples;
			} else {
				begin = c0 + size - tuples;
			}
			if (result == OperatorResultType::HAVE_MORE_OUTPUT) {
				mpx_in_process = true;
			} else {
				mpx_in_process = false;
				c++;
				// fold the routing window into this run
				idx_t &skips = multiplexer.GetNumCacheFlushingSkips(*mpx_state);
				if (skips > 0 && c < n_chunks) {
					idx_t n = std::min<idx_t>(skips, n_chunks - c);
					idx_t extra = ChunkStart(c + n) - ChunkStart(c);
					multiplexer.IncreaseInputTupleCount(*mpx_state, extra);
					if (skips != std::numeric_limits<idx_t>::max()) {
						skips -= n;
					}
					c += n;
					tuples += extra;
				}
			}
		}
		const idx_t path = multiplexer.GetCurrentPathIndex(*mpx_state);
		thread.current_join_path = &polar.join_paths[path];
		polr_round round;
		round.begin = begin;
		round.count = tuples;
		round.path = (uint32_t)path;
		round.emit = (!alternate || path == 0) ? 1u : 0u; // ALTERNATE forwards path 0 only (:445-447)
		Check(ctx, polr_probe_rounds(pipe, nullptr, &round, 1, round.emit ? out : nullptr, counts.data()),
		      "polr_probe_rounds");
		idx_t produced = 0;
		for (idx_t j = 0; j < k; j++) {
			produced += counts[j];
		}
		multiplexer.AddNumIntermediates(*mpx_state, produced); // :486
		num_intermediates_produced += produced;                // :487
		path_per_round.push_back((uint32_t)path);
		tuples_per_round.push_back(tuples);
	}
	// PushFinalize (:150-151)
	if (!multiplexer.Core(*mpx_state).first_mpx_run) {
		multiplexer.FinalizePathRun(*mpx_state, true);
	}
	intermediates_per_round = multiplexer.IntermediatesPerRound(*mpx_state);
	const auto &core = multiplexer.Core(*mpx_state);
	input_tuple_count_per_path.assign(core.input_tuple_count_per_path,
	                                  core.input_tuple_count_per_path + polar.join_paths.size());
	path_resistances.assign(core.path_resistances, core.path_resistances + polar.join_paths.size());
	context.config.log_tuples_routed = log_was;
}

void POLARPipelineExecutor::ExecuteDeviceRouted(polr_out *out, bool resident) {
	auto &multiplexer = *polar.multiplexer;
	polr_mpx_config cfg;
	memset(&cfg, 0, sizeof(cfg));
	cfg.routing = (uint32_t)multiplexer.routing;
	cfg.chunk_size = (uint32_t)STANDARD_VECTOR_SIZE;
	cfg.regret_budget = multiplexer.regret_budget;
	cfg.init_tuple_count = context.config.init_tuple_count;
	cfg.atc_multiplier = context.config.atc_multiplier;
	cfg.log_rounds = 1;
	cfg.max_log_rounds = (uint32_t)std::min<idx_t>(n_chunks * polar.join_paths.size() + 16, 1u << 24);
	polr_mpx *mpx = nullptr;
	Check(ctx, polr_mpx_create(pipe, &cfg, &mpx), "polr_mpx_create");
	try {
		if (!chunk_offsets.empty()) {
			Check(ctx, polr_mpx_set_chunk_offsets(mpx, chunk_offsets.data(), n_chunks), "polr_mpx_set_chunk_offsets");
		}
		if (resident) {
			const uint64_t begin = 0, end = n_chunks;
			Check(ctx, polr_mpx_run_resident(&mpx, nullptr, &begin, &end, 1, out, POLR_RUN_FINISH),
			      "polr_mpx_run_resident");
		} else {
			Check(ctx, polr_mpx_run(mpx, nullptr, 0, n_chunks, out), "polr_mpx_run");
		}
		polr_mpx_stats stats;
		Check(ctx, polr_mpx_finish(mpx, nullptr, &stats), "polr_mpx_finish");
		num_intermediates_produced = stats.num_intermediates;
		input_tuple_count_per_path.assign(stats.input_tuple_count_per_path,
		                                  stats.input_tuple_count_per_path + polar.join_paths.size());
		path_resistances.assign(stats.path_resistances, stats.path_resistances + polar.join_paths.size());
		uint64_t n = 0;
		path_per_round.assign(cfg.max_log_rounds, 0);
		tuples_per_round.assign(cfg.max_log_rounds, 0);
		intermediates_per_round.assign(cfg.max_log_rounds, 0);
		Check(ctx, polr_mpx_fetch_log(mpx, nullptr, path_per_round.data(), tuples_per_round.data(),
		                              intermediates_per_round.data(), cfg.max_log_rounds, &n),
		      "polr_mpx_fetch_log");
		path_per_round.resize(n);
		tuples_per_round.resize(n);
		intermediates_per_round.resize(n);
	} catch (...) {
		polr_mpx_destroy(mpx);
		throw;
	}
	polr_mpx_destroy(mpx);
}

string POLARPipelineExecutor::LogCsv() const {
	std::stringstream log;
	const idx_t P = polar.join_paths.size();
	if (alternate) {
		for (idx_t i = 0; i < P; i++) {
			log << "path_" << i << ",";
		}
		log << "\n";
		for (idx_t i = 0; i + P <= intermediates_per_round.size(); i += P) {
			for (idx_t j = 0; j < P; j++) {
				log << intermediates_per_round[i + j] << ",";
			}
			log << "\n";
		}
	} else {
		log << "intermediates\n";
		for (auto v : intermediates_per_round) {
			log << v << "\n";
		}
	}
	return log.str();
}

} // namespace duckdb_polr
