#include "physical_multiplexer.hpp"

#include <algorithm>
#include <iostream>

namespace duckdb_polr {

namespace {

// per-executor state of the operator (the reference's MultiplexerState, physical_multiplexer.cpp:20-82): the
// routing core, the strategy object in front of it and the two logs
struct MpxState : public OperatorState {
	polr::MultiplexerCore core; // resistances, counters, strategy state
	unique_ptr<RoutingStrategy> strategy;
	vector<idx_t> per_round;               // log_tuples_routed: intermediates of every closed run
	vector<vector<idx_t>> per_path_alt;    // ALTERNATE: one column per path
};

MpxState &Of(OperatorState &s) {
	return static_cast<MpxState &>(s);
}

template <class S>
RoutingStrategy *Make(polr::MultiplexerCore *core, idx_t init_tuples) {
	return new S(core, init_tuples);
}
RoutingStrategy *MakeAlternate(polr::MultiplexerCore *core, idx_t) {
	return new AlternateRoutingStrategy(core);
}
RoutingStrategy *MakeDefault(polr::MultiplexerCore *core, idx_t) {
	return new DefaultPathRoutingStrategy(core, 0);
}

// strategy objects by MultiplexerRouting value (config.hpp:41-50)
struct StrategyEntry {
	MultiplexerRouting routing;
	RoutingStrategy *(*make)(polr::MultiplexerCore *, idx_t);
};
const StrategyEntry kStrategies[] = {
    {MultiplexerRouting::ALTERNATE, MakeAlternate},
    {MultiplexerRouting::ADAPTIVE_REINIT, Make<AdaptiveReinitRoutingStrategy>},
    {MultiplexerRouting::DYNAMIC, Make<DynamicRoutingStrategy>},
    {MultiplexerRouting::INIT_ONCE, Make<InitOnceRoutingStrategy>},
    {MultiplexerRouting::OPPORTUNISTIC, Make<OpportunisticRoutingStrategy>},
    {MultiplexerRouting::DEFAULT_PATH, MakeDefault},
    {MultiplexerRouting::BACKPRESSURE, MakeDefault},
    {MultiplexerRouting::EXPONENTIAL_BACKOFF, Make<ExponentialBackoffRoutingStrategy>},
};

} // namespace

PhysicalMultiplexer::PhysicalMultiplexer(vector<LogicalType> types, idx_t estimated_cardinality, idx_t path_count_p,
                                         double regret_budget_p, MultiplexerRouting routing_p)
    : PhysicalOperator(PhysicalOperatorType::MULTIPLEXER, std::move(types), estimated_cardinality),
      path_count(path_count_p), regret_budget(regret_budget_p), routing(routing_p) {
	if (path_count_p > polr::kMaxPaths) {
		throw NotImplementedException("more join orders than the multiplexer supports");
	}
}

unique_ptr<OperatorState> PhysicalMultiplexer::GetOperatorState(ExecutionContext &context) const {
	const auto &cfg = context.client.config;
	if (cfg.time_resistance) {
		// wall-clock resistances are nondeterministic by construction; the device path keeps the default
		throw NotImplementedException("time_resistance is outside the MI355X path");
	}
	unique_ptr<MpxState> st(new MpxState());
	st->core.Init((uint32_t)routing, (uint32_t)path_count, regret_budget, cfg.init_tuple_count, cfg.atc_multiplier);
	for (const auto &entry : kStrategies) {
		if (entry.routing == routing) {
			st->strategy.reset(entry.make(&st->core, cfg.init_tuple_count));
		}
	}
	if (!st->strategy) {
		throw InternalException("unknown routing strategy");
	}
	return unique_ptr<OperatorState>(st.release());
}

// reference behaviour: physical_multiplexer.cpp:100-121
OperatorResultType PhysicalMultiplexer::Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
                                                GlobalOperatorState &, OperatorState &mpx_state) const {
	MpxState &st = Of(mpx_state);
	polr::MultiplexerCore &core = st.core;
	if (core.first_mpx_run) {
		// very first call of this executor: nothing to close yet
		core.first_mpx_run = 0;
		if (routing == MultiplexerRouting::ALTERNATE) {
			core.alternate_mode = 1;
			st.per_path_alt.assign(path_count, vector<idx_t>());
		}
	} else {
		FinalizePathRun(mpx_state, context.client.config.log_tuples_routed);
	}
	const OperatorResultType verdict = st.strategy->Route(input, chunk);
	// what the run that starts now is accounted under
	core.current_path_idx = core.next_path_idx;
	core.current_path_tuple_count = chunk.size();
	core.num_cache_flushing_skips = core.rs_cache_skips;
	return verdict;
}

void PhysicalMultiplexer::AddNumIntermediates(OperatorState &mpx_state, idx_t count) const {
	Of(mpx_state).core.AddNumIntermediates(count);
}

void PhysicalMultiplexer::IncreaseInputTupleCount(OperatorState &mpx_state, idx_t tuple_count) const {
	Of(mpx_state).core.IncreaseInputTupleCount(tuple_count);
}

// reference behaviour: physical_multiplexer.cpp:132-174 (the reward itself: MultiplexerCore::FinalizePathRun)
void PhysicalMultiplexer::FinalizePathRun(OperatorState &mpx_state, bool log_tuples_routed) const {
	MpxState &st = Of(mpx_state);
	const idx_t closed_path = st.core.current_path_idx;
	const idx_t intermediates = st.core.FinalizePathRun();
	if (!st.per_path_alt.empty()) {
		st.per_path_alt[closed_path].push_back(intermediates);
	}
	if (log_tuples_routed) {
		st.per_round.push_back(intermediates);
	}
}

idx_t PhysicalMultiplexer::GetCurrentPathIndex(OperatorState &mpx_state) const {
	return Of(mpx_state).core.current_path_idx;
}

idx_t &PhysicalMultiplexer::GetNumCacheFlushingSkips(OperatorState &mpx_state) const {
	return Of(mpx_state).core.num_cache_flushing_skips;
}

bool PhysicalMultiplexer::WasExecuted(OperatorState &mpx_state) const {
	const uint64_t *routed = Of(mpx_state).core.input_tuple_count_per_path;
	return std::any_of(routed, routed + path_count, [](uint64_t n) { return n != 0; });
}

// same text as the reference prints (physical_multiplexer.cpp:186-192)
void PhysicalMultiplexer::PrintStatistics(OperatorState &mpx_state) const {
	const polr::MultiplexerCore &core = Of(mpx_state).core;
	std::cout << "Input tuple counts per path\n";
	for (idx_t path = 0; path < path_count; path++) {
		std::cout << path << ": " << core.input_tuple_count_per_path[path] << "\n";
	}
}

// the two CSV shapes of the reference's log files (physical_multiplexer.cpp:194-219): in ALTERNATE mode one row
// per source chunk with one column per path ("path_0,path_1,..."), otherwise one intermediates value per closed run
void PhysicalMultiplexer::WriteLogToFile(OperatorState &mpx_state, std::ostream &file) const {
	const MpxState &st = Of(mpx_state);
	if (st.per_path_alt.empty()) {
		file << "intermediates\n";
		for (const idx_t v : st.per_round) {
			file << v << "\n";
		}
		return;
	}
	const idx_t n_paths = st.per_path_alt.size();
	for (idx_t path = 0; path < n_paths; path++) {
		file << "path_" << path << ",";
	}
	file << "\n";
	const idx_t n_chunks = st.per_path_alt.front().size();
	for (idx_t chunk_no = 0; chunk_no < n_chunks; chunk_no++) {
		for (idx_t path = 0; path < n_paths; path++) {
			file << st.per_path_alt[path][chunk_no] << ",";
		}
		file << "\n";
	}
}

const polr::MultiplexerCore &PhysicalMultiplexer::Core(OperatorState &mpx_state) const {
	return Of(mpx_state).core;
}
const vector<idx_t> &PhysicalMultiplexer::IntermediatesPerRound(OperatorState &mpx_state) const {
	return Of(mpx_state).per_round;
}
const vector<vector<idx_t>> &PhysicalMultiplexer::IntermediatesAlternateMode(OperatorState &mpx_state) const {
	return Of(mpx_state).per_path_alt;
}

} // namespace duckdb_polr
