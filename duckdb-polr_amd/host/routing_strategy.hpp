// duckdb-polr_amd/host/routing_strategy.hpp -- host mirror of the reference's RoutingStrategy family
// (src/include/duckdb/execution/operator/polr/routing_strategy.hpp:15-213,
//  src/execution/operator/polr/routing_strategy.cpp:7-463).  Same class names and the same Route()
// contract; the arithmetic lives in csrc/polr_routing.h so that the device routing step and these
// classes are one implementation.
#pragma once

#include "../csrc/polr_routing.h"
#include "polr_host_types.hpp"

namespace duckdb_polr {

class RoutingStrategyState {
public:
	explicit RoutingStrategyState(polr::MultiplexerCore *core_p) : core(core_p) {
		sel.Initialize();
	}
	polr::MultiplexerCore *core; // path_resistances + all strategy state (shared with the device layout)
	SelectionVector sel;
	idx_t &chunk_size() {
		return core->chunk_size;
	}
	idx_t &next_path_idx() {
		return core->next_path_idx;
	}
	idx_t &next_tuple_count() {
		return core->next_tuple_count;
	}
	idx_t &chunk_offset() {
		return core->chunk_offset;
	}
	idx_t &num_cache_flushing_skips() {
		return core->rs_cache_skips;
	}
};

class RoutingStrategy {
public:
	RoutingStrategy(polr::MultiplexerCore *core, idx_t init_tuple_count_p)
	    : routing_state(new RoutingStrategyState(core)), init_tuple_count(init_tuple_count_p) {
	}
	virtual ~RoutingStrategy() {
	}
	// routing_strategy.hpp:47-53
	virtual OperatorResultType Route(DataChunk &input, DataChunk &chunk) const {
		auto &core = *routing_state->core;
		core.chunk_size = input.size();
		core.next_path_idx = DetermineNextPath();
		core.next_tuple_count = DetermineNextTupleCount();
		return SelectTuples(input, chunk);
	}
	std::unique_ptr<RoutingStrategyState> routing_state;
	const idx_t init_tuple_count;

protected:
	OperatorResultType SelectTuples(DataChunk &input, DataChunk &chunk) const;
	virtual idx_t DetermineNextPath() const {
		return routing_state->core->DetermineNextPath();
	}
	virtual idx_t DetermineNextTupleCount() const {
		return routing_state->core->DetermineNextTupleCount();
	}
};

#define POLR_STRATEGY(NAME)                                                                                            \
	class NAME : public RoutingStrategy {                                                                              \
	public:                                                                                                            \
		NAME(polr::MultiplexerCore *core, idx_t init_tuple_count_p) : RoutingStrategy(core, init_tuple_count_p) {      \
		}                                                                                                              \
	};
POLR_STRATEGY(OpportunisticRoutingStrategy)
POLR_STRATEGY(InitOnceRoutingStrategy)
POLR_STRATEGY(AdaptiveReinitRoutingStrategy)
POLR_STRATEGY(ExponentialBackoffRoutingStrategy)
POLR_STRATEGY(DynamicRoutingStrategy)
POLR_STRATEGY(DefaultPathRoutingStrategy)
#undef POLR_STRATEGY

class AlternateRoutingStrategy : public RoutingStrategy {
public:
	explicit AlternateRoutingStrategy(polr::MultiplexerCore *core) : RoutingStrategy(core, 0) {
	}
	OperatorResultType Route(DataChunk &input, DataChunk &chunk) const override; // routing_strategy.cpp:440-452
};

// CalculateJoinPathWeights, routing_strategy.cpp:267-316
void CalculateJoinPathWeights(const vector<double> &join_path_costs, vector<double> &path_weights,
                              double regret_budget);

} // namespace duckdb_polr
