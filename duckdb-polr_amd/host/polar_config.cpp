#include "polar_config.hpp"

#include <chrono>

namespace duckdb_polr {

POLARConfig::POLARConfig(ClientContext &context_p, vector<PhysicalHashJoin *> joins_p,
                         idx_t source_estimated_cardinality_p, unique_ptr<JoinEnumerationAlgo> enumerator_p)
    : context(context_p), enumerator(std::move(enumerator_p)), joins(std::move(joins_p)),
      source_estimated_cardinality(source_estimated_cardinality_p),
      measure_polr_pipeline(context_p.config.measure_polr_pipeline),
      log_tuples_routed(context_p.config.log_tuples_routed) {
}

bool POLARConfig::GenerateJoinOrders() {
	const auto begin = std::chrono::system_clock::now();
	// Step I (polar_config.cpp:31-46): consecutive INNER hash joins, at least two
	for (idx_t i = 0; i < joins.size(); i++) {
		if (joins[i]->join_type != JoinType::INNER) {
			return false;
		}
		hash_join_idxs.push_back(i);
	}
	if (hash_join_idxs.size() <= 1) {
		return false;
	}
	vector<idx_t> num_columns_per_join;
	for (auto *j : joins) {
		num_columns_per_join.push_back(j->types.size());
	}
	// Step II (:57-95): which join's build side provides each probe key
	vector<idx_t> column_counts;
	column_counts.push_back(joins.front()->probe_types.size());
	for (idx_t i = 0; i < joins.size(); i++) {
		join_prerequisites[i] = vector<idx_t>();
	}
	for (idx_t i = 0; i < joins.size(); i++) {
		auto *join = joins[i];
		idx_t num_columns_from_right =
		    join->right_projection_map.empty() ? join->build_types.size() : join->right_projection_map.size();
		column_counts.push_back(column_counts.back() + num_columns_from_right);
		for (idx_t j = 0; j < join->conditions.size(); j++) {
			auto &condition = join->conditions[j];
			if (!condition.left_is_bound_ref) {
				return false; // "Let's not POLAR, weird stuff going on" (:78-81)
			}
			if (condition.left_index >= column_counts.front()) {
				for (idx_t k = 1; k < column_counts.size(); k++) {
					if (column_counts[k] > condition.left_index) {
						join_prerequisites[i].push_back(k - 1);
						break;
					}
				}
			}
		}
	}
	enumerator->GenerateJoinOrders(hash_join_idxs, join_prerequisites, joins, join_paths);
	if (join_paths.size() < 2) {
		return false;
	}
	auto routing = context.config.multiplexer_routing;
	auto prev_types = joins.front()->probe_types;
	adaptive_union.reset(new PhysicalAdaptiveUnion(joins.back()->types, prev_types.size(), num_columns_per_join,
	                                               joins.back()->estimated_cardinality));
	double regret_budget = context.config.regret_budget;
	if (routing == MultiplexerRouting::EXPONENTIAL_BACKOFF) {
		// re-purposed as the window cap (:115-120)
		idx_t max_threads = std::max<idx_t>(1, context.config.threads);
		regret_budget = source_estimated_cardinality / 10240.0 / 10 / max_threads;
	}
	multiplexer.reset(new PhysicalMultiplexer(prev_types, source_estimated_cardinality, join_paths.size(),
	                                          regret_budget, routing));
	multiplexer_idx = hash_join_idxs.front();

	// per-path probe-column rebinding (:152-229)
	left_expression_bindings.reserve(join_paths.size());
	const idx_t seed_table_column_count = prev_types.size();
	vector<idx_t> column_offsets;
	column_offsets.push_back(seed_table_column_count);
	for (idx_t i = 0; i < joins.size(); i++) {
		auto &join = joins[i];
		idx_t num_columns_from_right =
		    join->right_projection_map.empty() ? join->build_types.size() : join->right_projection_map.size();
		column_offsets.push_back(column_offsets.back() + num_columns_from_right);
		for (idx_t j = 0; j < join->conditions.size(); j++) {
			auto &condition = join->conditions[j];
			if (condition.left_index >= seed_table_column_count) {
				for (idx_t k = 1; k < column_offsets.size(); k++) {
					if (column_offsets[k] > condition.left_index) {
						idx_t join_idx = k - 1;
						idx_t relative_column_idx = condition.left_index - column_offsets[join_idx];
						relative_column_binding_map[i][j] = std::make_pair(join_idx, relative_column_idx);
						break;
					}
				}
			}
		}
	}
	for (idx_t join_path_idx = 0; join_path_idx < join_paths.size(); join_path_idx++) {
		auto &join_path = join_paths[join_path_idx];
		vector<std::map<idx_t, idx_t>> expression_bindings(join_path.size());
		vector<idx_t> current_offsets;
		current_offsets.push_back(seed_table_column_count);
		for (idx_t j = 0; j < join_path.size(); j++) {
			auto join_idx = join_path[j];
			auto column_bindings = relative_column_binding_map.find(join_idx);
			if (column_bindings != relative_column_binding_map.cend()) {
				for (auto &binding : column_bindings->second) {
					auto probe_join_idx = binding.second.first;
					auto relative_column_idx = binding.second.second;
					for (idx_t i = 0; i < current_offsets.size(); i++) {
						if (join_path[i] == probe_join_idx) {
							expression_bindings[j][binding.first] = current_offsets[i] + relative_column_idx;
						}
					}
				}
			}
			idx_t additional_columns = joins[join_idx]->right_projection_map.empty()
			                               ? joins[join_idx]->build_types.size()
			                               : joins[join_idx]->right_projection_map.size();
			current_offsets.push_back(current_offsets.back() + additional_columns);
		}
		left_expression_bindings.push_back(expression_bindings);
	}
	const auto end = std::chrono::system_clock::now();
	enumeration_time_ms = std::chrono::duration<double, std::milli>(end - begin).count();
	return true;
}

} // namespace duckdb_polr
