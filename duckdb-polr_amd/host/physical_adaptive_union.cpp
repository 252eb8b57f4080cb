#include "physical_adaptive_union.hpp"

namespace duckdb_polr {

PhysicalAdaptiveUnion::PhysicalAdaptiveUnion(vector<LogicalType> types, idx_t num_columns_from_left_p,
                                             vector<idx_t> num_columns_per_join_p, idx_t estimated_cardinality)
    : PhysicalOperator(PhysicalOperatorType::ADAPTIVE_UNION, std::move(types), estimated_cardinality),
      num_columns_from_left(num_columns_from_left_p), num_columns_per_join(std::move(num_columns_per_join_p)) {
}

class AdaptiveUnionState : public OperatorState {
public:
	explicit AdaptiveUnionState(vector<idx_t> *input_join_order_p = nullptr) : input_join_order(input_join_order_p) {
	}
	vector<idx_t> *input_join_order;
};

unique_ptr<OperatorState> PhysicalAdaptiveUnion::GetOperatorState(ExecutionContext &context) const {
	return unique_ptr<OperatorState>(new AdaptiveUnionState());
}

unique_ptr<OperatorState>
PhysicalAdaptiveUnion::GetOperatorStateWithStaticJoinOrder(ExecutionContext &context,
                                                           vector<idx_t> *input_join_order) const {
	return unique_ptr<OperatorState>(new AdaptiveUnionState(input_join_order));
}

// physical_adaptive_union.cpp:37-76
OperatorResultType PhysicalAdaptiveUnion::Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
                                                  GlobalOperatorState &gstate_p, OperatorState &state_p) const {
	auto &state = (AdaptiveUnionState &)state_p;
	if (!context.thread.current_join_path && !state.input_join_order) {
		throw InternalException("adaptive union without a current join path");
	}
	vector<idx_t> &current_join_path =
	    state.input_join_order ? *state.input_join_order : *context.thread.current_join_path;
	chunk.SetCardinality(input);
	for (idx_t i = 0; i < num_columns_from_left; i++) {
		chunk.data[i].Reference(input.data[i]);
	}
	idx_t current_offset = num_columns_from_left;
	for (idx_t i = 0; i < current_join_path.size(); i++) {
		idx_t join_idx = current_join_path[i];
		idx_t target_columns_begin = join_idx == 0 ? num_columns_from_left : num_columns_per_join[join_idx - 1];
		for (idx_t j = target_columns_begin; j < num_columns_per_join[join_idx]; j++) {
			chunk.data[j].Reference(input.data[current_offset]);
			current_offset++;
		}
	}
	return OperatorResultType::NEED_MORE_INPUT;
}

} // namespace duckdb_polr
