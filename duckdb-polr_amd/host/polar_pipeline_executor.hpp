// duckdb-polr_amd/host/polar_pipeline_executor.hpp -- host mirror of POLARPipelineExecutor
// (src/include/duckdb/parallel/polar_pipeline_executor.hpp, src/parallel/polar_pipeline_executor.cpp:21-538).
//
// The reference drives one 1024-tuple chunk at a time through virtual Execute() calls.  The MI355X
// driver keeps the reference's control flow at the granularity that matters for routing -- one
// *path run* (the tuples between two FinalizePathRun calls: a routed slice plus the
// num_cache_flushing_skips chunks that follow it on the same path) -- and turns each run into one
// launch of the path kernel:
//
//   HOST_ROUTED    PhysicalMultiplexer::Execute on the host per routing decision, polr_probe_rounds per
//                  run, AddNumIntermediates with the counters that come back.  One device round trip per
//                  decision: the literal transcription of RunPath, used to check the device router.
//   DEVICE_ROUTED  the multiplexer state lives in HBM (polr_mpx_run): the path kernel routes for itself,
//                  one launch per routing round, no host round trip per decision.
//   DEVICE_RESIDENT the whole run is one launch (polr_mpx_run_resident): router wave + probe workgroups
//                  stay on the device until the source is exhausted.
//
// Chunk caches and the in_process_joins stack of the reference only regroup tuples into fuller chunks;
// they change neither the intermediates of a run nor the output row set, so they have no counterpart
// here (the path kernel's per-wave LDS queues play that role on the device).
#pragma once

#include "polar_config.hpp"

namespace duckdb_polr {

enum class RoutingPlacement : uint8_t { HOST_ROUTED, DEVICE_ROUTED, DEVICE_RESIDENT };

class POLARPipelineExecutor {
public:
	// `pipe`: the device pipeline (probe columns + build sides + polar.join_paths) the joins run on;
	// chunk_offsets: source chunk boundaries in tuple positions (empty: fixed STANDARD_VECTOR_SIZE chunks)
	POLARPipelineExecutor(ClientContext &context, POLARConfig &polar, polr_ctx *ctx, polr_pipeline *pipe,
	                      idx_t n_tuples, vector<idx_t> chunk_offsets = vector<idx_t>());
	~POLARPipelineExecutor();

	// PipelineExecutor::Execute(max_chunks) + PushFinalize over the whole source
	bool Execute(RoutingPlacement placement, polr_out *out = nullptr);

	idx_t num_intermediates_produced = 0;
	vector<idx_t> input_tuple_count_per_path;
	vector<idx_t> intermediates_per_round;       // one entry per FinalizePathRun (log_tuples_routed)
	vector<uint32_t> path_per_round;
	vector<idx_t> tuples_per_round;
	vector<double> path_resistances;
	// the reference's log artefacts (physical_multiplexer.cpp:194-219): "intermediates\n.." or "path_i,.."
	string LogCsv() const;

private:
	void ExecuteHostRouted(polr_out *out);
	void ExecuteDeviceRouted(polr_out *out, bool resident);
	idx_t ChunkStart(idx_t c) const;

	ClientContext &context;
	POLARConfig &polar;
	polr_ctx *ctx;
	polr_pipeline *pipe;
	idx_t n_tuples;
	vector<idx_t> chunk_offsets;
	idx_t n_chunks;
	ThreadContext thread;
	bool alternate = false;
};

} // namespace duckdb_polr
