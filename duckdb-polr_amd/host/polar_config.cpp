#include "polar_config.hpp"

#include <chrono>

namespace duckdb_polr {

namespace {

// build columns a join appends to the chunk (its projection map, if any, decides)
idx_t BuildWidth(const PhysicalHashJoin &join) {
	return join.right_projection_map.empty() ? join.build_types.size() : join.right_projection_map.size();
}

// `ends[i]` = column count after the probe side (i = 0) / after the (i-1)-th join of the sequence: which entry of the
// sequence does `column` belong to?  -1: the probe side (or nothing).
int64_t OwnerOf(const vector<idx_t> &ends, idx_t column) {
	if (column < ends.front()) {
		return -1;
	}
	for (idx_t i = 1; i < ends.size(); i++) {
		if (column < ends[i]) {
			return (int64_t)i - 1;
		}
	}
	return -1;
}

} // namespace

POLARConfig::POLARConfig(ClientContext &context_p, JoinList joins_p, idx_t source_estimated_cardinality_p,
                         unique_ptr<JoinEnumerationAlgo> enumerator_p)
    : context(context_p), enumerator(std::move(enumerator_p)), joins(std::move(joins_p)),
      source_estimated_cardinality(source_estimated_cardinality_p),
      measure_polr_pipeline(context_p.config.measure_polr_pipeline),
      log_tuples_routed(context_p.config.log_tuples_routed) {
}

// Reference behaviour: polar_config.cpp:19-249.  Four steps: (1) is the run of joins multiplexable at all,
// (2) which join needs which other join's build columns, (3) the bank of orders + the two POLAR operators,
// (4) for every order, where each dependent probe key sits in that order's column layout.
bool POLARConfig::GenerateJoinOrders() {
	const auto t_begin = std::chrono::system_clock::now();
	const idx_t k = joins.size();
	const idx_t probe_width = k ? joins.front()->probe_types.size() : 0;

	// (1) INNER hash joins only, and at least two of them (:31-46)
	for (idx_t j = 0; j < k; j++) {
		if (joins[j]->join_type != JoinType::INNER) {
			return false;
		}
		hash_join_idxs.push_back(j);
	}
	if (k < 2) {
		return false;
	}

	// (2) the original column layout and, from it, dependencies + the (source join, relative column) of every
	// probe key that is a build column (:57-95, :152-190)
	vector<idx_t> layout_ends {probe_width};
	vector<idx_t> output_widths;
	for (idx_t j = 0; j < k; j++) {
		join_prerequisites[j] = JoinOrder();
		layout_ends.push_back(layout_ends.back() + BuildWidth(*joins[j]));
		output_widths.push_back(joins[j]->types.size());
		for (idx_t c = 0; c < joins[j]->conditions.size(); c++) {
			const auto &cond = joins[j]->conditions[c];
			if (!cond.left_is_bound_ref || cond.left_is_cast) {
				// the reference gives up on anything but column references (:78-81) -- and on CASTs of them too: it compares
				// the bound cast's type (OPERATOR_CAST) with ExpressionType::CAST, so "Let's not POLAR" is all a CAST'ed key
				// ever gets (pinned: tests/golden/key_semantics.json, with_cast.reference_multiplexed = false)
				return false;
			}
			const int64_t source = OwnerOf(layout_ends, cond.left_index);
			if (source >= 0) {
				join_prerequisites[j].push_back((idx_t)source);
				relative_column_binding_map[j][c] = std::make_pair((idx_t)source, cond.left_index - layout_ends[source]);
			}
		}
	}

	// (3) the orders, then the operators around them (:97-150)
	enumerator->GenerateJoinOrders(hash_join_idxs, join_prerequisites, joins, join_paths);
	if (join_paths.size() < 2) {
		return false;
	}
	const auto routing = context.config.multiplexer_routing;
	double budget = context.config.regret_budget;
	if (routing == MultiplexerRouting::EXPONENTIAL_BACKOFF) {
		// this strategy re-purposes the knob as its window cap (:115-120)
		budget = source_estimated_cardinality / 10240.0 / 10 / std::max<idx_t>(1, context.config.threads);
	}
	multiplexer.reset(new PhysicalMultiplexer(joins.front()->probe_types, source_estimated_cardinality,
	                                          join_paths.size(), budget, routing));
	multiplexer_idx = hash_join_idxs.front();
	adaptive_union.reset(new PhysicalAdaptiveUnion(joins.back()->types, probe_width, output_widths,
	                                               joins.back()->estimated_cardinality));

	// (4) per order: absolute column of every dependent probe key in that order's layout (:192-229).  The source
	// join is looked for among the joins placed so far INCLUDING the current position, as the reference does.
	left_expression_bindings.reserve(join_paths.size());
	for (const JoinOrder &order : join_paths) {
		vector<std::map<idx_t, idx_t>> per_position(order.size());
		vector<idx_t> begins {probe_width}; // begins[i]: first build column of the join at position i
		for (idx_t pos = 0; pos < order.size(); pos++) {
			const auto dependent = relative_column_binding_map.find(order[pos]);
			if (dependent != relative_column_binding_map.cend()) {
				for (const auto &key : dependent->second) {
					for (idx_t earlier = 0; earlier < begins.size(); earlier++) {
						if (order[earlier] == key.second.first) {
							per_position[pos][key.first] = begins[earlier] + key.second.second;
						}
					}
				}
			}
			begins.push_back(begins.back() + BuildWidth(*joins[order[pos]]));
		}
		left_expression_bindings.push_back(std::move(per_position));
	}
	enumeration_time_ms = std::chrono::duration<double, std::milli>(std::chrono::system_clock::now() - t_begin).count();
	return true;
}

std::unique_ptr<POLARConfig> MakePolarConfigForPipeline(ClientContext &context, JoinList joins,
                                                        idx_t source_estimated_cardinality) {
	std::unique_ptr<POLARConfig> polar(new POLARConfig(context, joins, source_estimated_cardinality,
	                                                   JoinEnumerationAlgo::CreateEnumerationAlgo(context)));
	bool generated = polar->GenerateJoinOrders();
	if (!generated && context.config.join_enumerator != JoinEnumerator::BFS_MIN_CARD) {
		polar.reset(new POLARConfig(context, joins, source_estimated_cardinality,
		                            unique_ptr<JoinEnumerationAlgo>(new BFSEnumeration(
		                                unique_ptr<CandidateSelector>(new MinCardinalitySelector())))));
		generated = polar->GenerateJoinOrders();
		if (generated) {
			polar->multiplexer->routing = MultiplexerRouting::DEFAULT_PATH;
		}
	}
	if (!generated) {
		return nullptr;
	}
	return polar;
}

} // namespace duckdb_polr
