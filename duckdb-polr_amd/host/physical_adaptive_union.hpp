// duckdb-polr_amd/host/physical_adaptive_union.hpp -- host mirror of the reference's PhysicalAdaptiveUnion operator
// (interface: src/include/duckdb/execution/operator/polr/physical_adaptive_union.hpp:15-40, behaviour:
// src/execution/operator/polr/physical_adaptive_union.cpp:12-82).
//
// A join path emits its build columns in PATH order; the operators above the multiplexed section expect them in the
// ORIGINAL join order.  The union re-references columns, it never copies data.  This mirror resolves, once at
// construction, where every join's build columns live in the original layout (`spans`), so Execute is a walk over
// the current path that points output columns at input columns.  (On the device the same permutation is the
// tuple-slot map of the probe kernels: slot 1 + j always belongs to join j.)
#pragma once

#include "polr_host_types.hpp"

namespace duckdb_polr {

class PhysicalAdaptiveUnion : public PhysicalOperator {
	struct ColumnSpan {
		idx_t first; // first output column of the join's build columns (original layout)
		idx_t width; // number of build columns the join contributes
	};

public:
	// out_types: output schema; n_probe_columns: probe-side columns that pass through unchanged;
	// cumulative_widths[i]: output width after original join i
	PhysicalAdaptiveUnion(vector<LogicalType> out_types, idx_t n_probe_columns, vector<idx_t> cumulative_widths,
	                      idx_t estimated_cardinality);

	// points the output columns at the input's columns: probe side first, then every join's build columns at the
	// place the original join order gives them
	OperatorResultType Execute(ExecutionContext &context, DataChunk &path_ordered, DataChunk &original_ordered,
	                           GlobalOperatorState &gstate, OperatorState &union_state) const override;

	// state whose path is read from the thread context at every call ...
	unique_ptr<OperatorState> GetOperatorState(ExecutionContext &context) const override;
	// ... or fixed by the caller (an executor that only ever runs `input_join_order`)
	unique_ptr<OperatorState> GetOperatorStateWithStaticJoinOrder(ExecutionContext &context,
	                                                              vector<idx_t> *input_join_order) const;

	bool RequiresCache() const override { return false; }
	bool ParallelOperator() const override { return true; }

	const idx_t num_columns_from_left;        // (names of the reference's members)
	const vector<idx_t> num_columns_per_join;

private:
	vector<ColumnSpan> spans; // per original join
};

} // namespace duckdb_polr
