// duckdb-polr_amd/host/physical_multiplexer.hpp -- host mirror of the reference's PhysicalMultiplexer operator
// (interface: src/include/duckdb/execution/operator/polr/physical_multiplexer.hpp:16-49, behaviour:
// src/execution/operator/polr/physical_multiplexer.cpp:14-231).
//
// The operator keeps the reference's constructor, its virtual operator interface and the helpers the pipeline
// executor calls; all routing arithmetic is delegated to polr::MultiplexerCore (csrc/polr_routing.h), the one source
// that also runs on the device, so host-routed and device-routed runs decide identically.
#pragma once

#include <fstream>

#include "routing_strategy.hpp"

namespace duckdb_polr {

class PhysicalMultiplexer : public PhysicalOperator {
public:
	PhysicalMultiplexer(vector<LogicalType> types, idx_t estimated_cardinality, idx_t path_count_p,
	                    double regret_budget_p, MultiplexerRouting routing);

	// ---- operator interface ---------------------------------------------------------------------------------
	unique_ptr<OperatorState> GetOperatorState(ExecutionContext &context) const override;
	// closes the previous path run (reward), asks the strategy for the next slice of `input`
	OperatorResultType Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
	                           GlobalOperatorState &gstate, OperatorState &state) const override;
	bool ParallelOperator() const override { return true; }
	bool RequiresCache() const override { return false; }

	// ---- what the pipeline executor feeds back and asks -----------------------------------------------------
	void AddNumIntermediates(OperatorState &mpx_state, idx_t count) const;
	void IncreaseInputTupleCount(OperatorState &mpx_state, idx_t tuple_count) const;
	void FinalizePathRun(OperatorState &mpx_state, bool log_tuples_routed) const;
	idx_t GetCurrentPathIndex(OperatorState &mpx_state) const;
	idx_t &GetNumCacheFlushingSkips(OperatorState &mpx_state) const;
	bool WasExecuted(OperatorState &mpx_state) const;

	// ---- log artefacts (same shapes as the reference's files) -----------------------------------------------
	void PrintStatistics(OperatorState &mpx_state) const;
	void WriteLogToFile(OperatorState &mpx_state, std::ostream &file) const;

	// ---- read-only views for tests and for the executor's logs ----------------------------------------------
	const polr::MultiplexerCore &Core(OperatorState &mpx_state) const;
	const vector<idx_t> &IntermediatesPerRound(OperatorState &mpx_state) const;
	const vector<vector<idx_t>> &IntermediatesAlternateMode(OperatorState &mpx_state) const;

	idx_t path_count;
	double regret_budget;
	MultiplexerRouting routing;
	const double SMOOTHING_FACTOR = 0.5; // (the value MultiplexerCore::FinalizePathRun uses)
};

} // namespace duckdb_polr
