// duckdb-polr_amd/host/polar_config.hpp -- host mirror of POLARConfig
// (src/include/duckdb/parallel/polar_config.hpp:20-42, src/parallel/polar_config.cpp:19-249):
// finds the dependencies between the consecutive INNER hash joins of a pipeline, asks the enumerator
// for the bank of join orders, creates the multiplexer / adaptive union and the per-path probe-column
// rebinding (left_expression_bindings).
#pragma once

#include <unordered_map>

#include "physical_adaptive_union.hpp"
#include "physical_hash_join.hpp"
#include "physical_multiplexer.hpp"
#include "polar_enumeration_algo.hpp"

namespace duckdb_polr {

class POLARConfig {
public:
	// `joins`: the run of consecutive INNER hash joins in the optimizer's order; joins[0]->probe_types are
	// the columns that reach the first join (what the multiplexer sees)
	POLARConfig(ClientContext &context, vector<PhysicalHashJoin *> joins_p, idx_t source_estimated_cardinality,
	            unique_ptr<JoinEnumerationAlgo> enumerator_p);
	bool GenerateJoinOrders();

	ClientContext &context;
	const unique_ptr<JoinEnumerationAlgo> enumerator;
	vector<PhysicalHashJoin *> joins;
	idx_t source_estimated_cardinality;
	vector<vector<idx_t>> join_paths;
	vector<vector<std::map<idx_t, idx_t>>> left_expression_bindings;
	std::unordered_map<idx_t, vector<idx_t>> join_prerequisites;
	// (source join, relative column) of every condition whose probe key is a build column of another join
	std::map<idx_t, std::map<idx_t, std::pair<idx_t, idx_t>>> relative_column_binding_map;
	idx_t multiplexer_idx = 0;
	std::unique_ptr<PhysicalMultiplexer> multiplexer;
	std::unique_ptr<PhysicalAdaptiveUnion> adaptive_union;
	bool measure_polr_pipeline;
	bool log_tuples_routed;
	vector<idx_t> hash_join_idxs;
	double enumeration_time_ms = 0;
};

} // namespace duckdb_polr
