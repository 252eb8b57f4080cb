#include "routing_strategy.hpp"

namespace duckdb_polr {

// RoutingStrategy::SelectTuples, routing_strategy.cpp:7-33: whole chunk -> Reference, else a
// zero-copy slice sel[i] = chunk_offset + i
OperatorResultType RoutingStrategy::SelectTuples(DataChunk &input, DataChunk &chunk) const {
	auto &core = *routing_state->core;
	if (core.next_tuple_count == input.size()) {
		chunk.Reference(input);
		return OperatorResultType::NEED_MORE_INPUT;
	}
	if (core.chunk_offset + core.next_tuple_count > input.size()) {
		throw InternalException("routing slice exceeds the input chunk");
	}
	// a fresh selection buffer per slice: the sliced chunk may still be referenced downstream
	routing_state->sel.Initialize(core.next_tuple_count ? core.next_tuple_count : 1);
	auto *sel_vector = routing_state->sel.data();
	for (idx_t i = 0; i < core.next_tuple_count; i++) {
		sel_vector[i] = (sel_t)(core.chunk_offset + i);
	}
	chunk.Slice(input, routing_state->sel, core.next_tuple_count);
	if (core.chunk_offset + core.next_tuple_count == input.size()) {
		core.chunk_offset = 0;
		return OperatorResultType::NEED_MORE_INPUT;
	}
	core.chunk_offset += core.next_tuple_count;
	return OperatorResultType::HAVE_MORE_OUTPUT;
}

OperatorResultType AlternateRoutingStrategy::Route(DataChunk &input, DataChunk &chunk) const {
	auto &core = *routing_state->core;
	core.next_path_idx = core.next_tuple_count == 0 ? 0 : (core.next_path_idx + 1) % core.path_count;
	core.next_tuple_count = input.size();
	chunk.Reference(input);
	if (core.next_path_idx == (idx_t)core.path_count - 1) {
		return OperatorResultType::NEED_MORE_INPUT;
	}
	return OperatorResultType::HAVE_MORE_OUTPUT;
}

void CalculateJoinPathWeights(const vector<double> &join_path_costs, vector<double> &path_weights,
                              double regret_budget) {
	polr::MultiplexerCore core;
	core.Init(polr::DYNAMIC, (uint32_t)join_path_costs.size(), regret_budget, 0, 1);
	path_weights.resize(join_path_costs.size(), 1);
	for (idx_t i = 0; i < join_path_costs.size(); i++) {
		core.path_resistances[i] = join_path_costs[i];
		core.path_weights[i] = path_weights[i];
	}
	core.CalculateJoinPathWeights();
	for (idx_t i = 0; i < join_path_costs.size(); i++) {
		path_weights[i] = core.path_weights[i];
	}
}

} // namespace duckdb_polr
