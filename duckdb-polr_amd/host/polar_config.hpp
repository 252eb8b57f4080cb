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
	// joins_p: the run of consecutive hash joins of the pipeline in the optimizer's order (joins_p[0]->probe_types
	// = the columns that reach the first join, i.e. what the multiplexer sees); source_estimated_cardinality: the
	// pipeline source's estimate (EXPONENTIAL_BACKOFF derives its window cap from it)
	POLARConfig(ClientContext &context, JoinList joins_p, idx_t source_estimated_cardinality,
	            unique_ptr<JoinEnumerationAlgo> enumerator_p);

	// false: POLAR does not apply to this run of joins (non-INNER join, fewer than two joins, a probe key that is not a
	// column reference, fewer than two orders); true: everything below is filled in
	bool GenerateJoinOrders();

	// ---- inputs ------------------------------------------------------------------------------------------------
	ClientContext &context;
	const unique_ptr<JoinEnumerationAlgo> enumerator;
	JoinList joins;
	idx_t source_estimated_cardinality;
	bool measure_polr_pipeline;
	bool log_tuples_routed;

	// ---- results -----------------------------------------------------------------------------------------------
	vector<idx_t> hash_join_idxs;   // the multiplexed joins (positions in the pipeline's operator list)
	DependencyMap join_prerequisites; // join -> joins whose build columns its probe keys read
	// join -> condition -> (source join, column relative to that join's build columns), for dependent keys only
	std::map<idx_t, std::map<idx_t, std::pair<idx_t, idx_t>>> relative_column_binding_map;
	vector<JoinOrder> join_paths;   // the bank of orders; join_paths[0] = the original order
	// per order, per position: condition -> absolute column of the probe key in that order's layout
	vector<vector<std::map<idx_t, idx_t>>> left_expression_bindings;
	std::unique_ptr<PhysicalMultiplexer> multiplexer;
	std::unique_ptr<PhysicalAdaptiveUnion> adaptive_union;
	idx_t multiplexer_idx = 0;      // where the multiplexer goes in the operator list
	double enumeration_time_ms = 0;
};

// The POLAR part of Pipeline::Ready (src/parallel/pipeline.cpp:213-232): the configured enumerator first; if it finds
// fewer than two orders and is not BFS_MIN_CARD itself, BFS + MinCardinality (max_join_orders at its default of 24) is
// tried and -- when that finds a bank -- the multiplexer routes DEFAULT_PATH, so that intermediates are still counted
// ("to enable comparing enumerators").  nullptr: POLAR does not apply to this run of joins.
std::unique_ptr<POLARConfig> MakePolarConfigForPipeline(ClientContext &context, JoinList joins,
                                                        idx_t source_estimated_cardinality);

} // namespace duckdb_polr
