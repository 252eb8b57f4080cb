// duckdb-polr_amd/host/physical_adaptive_union.hpp -- host mirror of PhysicalAdaptiveUnion
// (src/include/duckdb/execution/operator/polr/physical_adaptive_union.hpp:15-40,
//  src/execution/operator/polr/physical_adaptive_union.cpp:12-82): re-references the build columns
// of a path-ordered chunk in the original join order.  No data movement.  (On the device the same
// permutation is the tuple-slot map of the path kernel: slot 1+j = join j.)
#pragma once

#include "polr_host_types.hpp"

namespace duckdb_polr {

class PhysicalAdaptiveUnion : public PhysicalOperator {
public:
	PhysicalAdaptiveUnion(vector<LogicalType> types, idx_t num_columns_from_left_p,
	                      vector<idx_t> num_columns_per_join_p, idx_t estimated_cardinality);

	unique_ptr<OperatorState> GetOperatorState(ExecutionContext &context) const override;
	unique_ptr<OperatorState> GetOperatorStateWithStaticJoinOrder(ExecutionContext &context,
	                                                              vector<idx_t> *input_join_order) const;
	OperatorResultType Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
	                           GlobalOperatorState &gstate, OperatorState &state) const override;
	bool ParallelOperator() const override {
		return true;
	}
	bool RequiresCache() const override {
		return false;
	}

	const idx_t num_columns_from_left;
	const vector<idx_t> num_columns_per_join; // cumulative output width after join i (joins[i]->types.size())
};

} // namespace duckdb_polr
