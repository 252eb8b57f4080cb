// duckdb-polr_amd/host/physical_multiplexer.hpp -- host mirror of PhysicalMultiplexer
// (src/include/duckdb/execution/operator/polr/physical_multiplexer.hpp:16-49,
//  src/execution/operator/polr/physical_multiplexer.cpp:14-231): same constructor, same virtual
// operator interface, same helpers the executor calls.
#pragma once

#include <fstream>

#include "routing_strategy.hpp"

namespace duckdb_polr {

class PhysicalMultiplexer : public PhysicalOperator {
public:
	PhysicalMultiplexer(vector<LogicalType> types, idx_t estimated_cardinality, idx_t path_count_p,
	                    double regret_budget_p, MultiplexerRouting routing);

	idx_t path_count;
	double regret_budget;
	MultiplexerRouting routing;
	const double SMOOTHING_FACTOR = 0.5;

public:
	unique_ptr<OperatorState> GetOperatorState(ExecutionContext &context) const override;
	OperatorResultType Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
	                           GlobalOperatorState &gstate, OperatorState &state) const override;

	void FinalizePathRun(OperatorState &state_p, bool log_tuples_routed) const;
	void AddNumIntermediates(OperatorState &state_p, idx_t count) const;
	idx_t GetCurrentPathIndex(OperatorState &state_p) const;

	bool ParallelOperator() const override {
		return true;
	}
	bool RequiresCache() const override {
		return false;
	}
	void PrintStatistics(OperatorState &state) const;
	void WriteLogToFile(OperatorState &state, std::ostream &file) const;
	bool WasExecuted(OperatorState &state_p) const;
	idx_t &GetNumCacheFlushingSkips(OperatorState &state_p) const;
	void IncreaseInputTupleCount(OperatorState &state_p, idx_t tuple_count) const;

	// read-only views for tests and for the executor's logs
	const polr::MultiplexerCore &Core(OperatorState &state_p) const;
	const vector<idx_t> &IntermediatesPerRound(OperatorState &state_p) const;
	const vector<vector<idx_t>> &IntermediatesAlternateMode(OperatorState &state_p) const;
};

} // namespace duckdb_polr
