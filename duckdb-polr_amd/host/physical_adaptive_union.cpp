#include "physical_adaptive_union.hpp"

namespace duckdb_polr {

namespace {

// which path produced the chunk: pinned by the caller, else whatever the executor put in the thread context
struct UnionState : public OperatorState {
	explicit UnionState(vector<idx_t> *pinned_path_p) : pinned_path(pinned_path_p) {
	}
	vector<idx_t> *pinned_path;
};

} // namespace

PhysicalAdaptiveUnion::PhysicalAdaptiveUnion(vector<LogicalType> types, idx_t num_columns_from_left_p,
                                             vector<idx_t> num_columns_per_join_p, idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::ADAPTIVE_UNION, std::move(types), estimated_cardinality),
      num_columns_from_left(num_columns_from_left_p), num_columns_per_join(std::move(num_columns_per_join_p)) {
	// cumulative widths -> (first column, width) of every join's build columns in the original layout
	idx_t begin = num_columns_from_left;
	for (idx_t end : num_columns_per_join) {
		spans.push_back(ColumnSpan {begin, end - begin});
		begin = end;
	}
}

unique_ptr<OperatorState> PhysicalAdaptiveUnion::GetOperatorState(ExecutionContext &) const {
	return unique_ptr<OperatorState>(new UnionState(nullptr));
}

unique_ptr<OperatorState>
PhysicalAdaptiveUnion::GetOperatorStateWithStaticJoinOrder(ExecutionContext &, vector<idx_t> *input_join_order) const {
	return unique_ptr<OperatorState>(new UnionState(input_join_order));
}

// reference behaviour: physical_adaptive_union.cpp:37-76
OperatorResultType PhysicalAdaptiveUnion::Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
                                                  GlobalOperatorState &, OperatorState &state_p) const {
	const vector<idx_t> *path = static_cast<UnionState &>(state_p).pinned_path;
	if (!path) {
		path = context.thread.current_join_path;
	}
	if (!path) {
		throw InternalException("adaptive union without a current join path");
	}
	chunk.SetCardinality(input);
	// the probe side passes through
	idx_t src = 0;
	for (; src < num_columns_from_left; src++) {
		chunk.data[src].Reference(input.data[src]);
	}
	// the input carries the build columns join by join in path order: hand each group to its home span
	for (const idx_t join : *path) {
		const ColumnSpan &home = spans[join];
		for (idx_t c = 0; c < home.width; c++) {
			chunk.data[home.first + c].Reference(input.data[src++]);
		}
	}
	return OperatorResultType::NEED_MORE_INPUT;
}

} // namespace duckdb_polr
