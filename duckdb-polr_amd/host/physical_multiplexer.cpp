#include "physical_multiplexer.hpp"

#include <iostream>
#include <sstream>

namespace duckdb_polr {

PhysicalMultiplexer::PhysicalMultiplexer(vector<LogicalType> types, idx_t estimated_cardinality, idx_t path_count_p,
                                         double regret_budget_p, MultiplexerRouting routing_p)
    : PhysicalOperator(PhysicalOperatorType::MULTIPLEXER, std::move(types), estimated_cardinality),
      path_count(path_count_p), regret_budget(regret_budget_p), routing(routing_p) {
	if (path_count_p > polr::kMaxPaths) {
		throw NotImplementedException("more join orders than the multiplexer supports");
	}
}

// MultiplexerState, physical_multiplexer.cpp:20-82
class MultiplexerState : public OperatorState {
public:
	MultiplexerState(idx_t path_count, MultiplexerRouting routing, double regret_budget, idx_t init_tuple_count,
	                 idx_t multiplier) {
		core.Init((uint32_t)routing, (uint32_t)path_count, regret_budget, init_tuple_count, multiplier);
		switch (routing) {
		case MultiplexerRouting::ADAPTIVE_REINIT:
			routing_strategy.reset(new AdaptiveReinitRoutingStrategy(&core, init_tuple_count));
			break;
		case MultiplexerRouting::ALTERNATE:
			routing_strategy.reset(new AlternateRoutingStrategy(&core));
			break;
		case MultiplexerRouting::DYNAMIC:
			routing_strategy.reset(new DynamicRoutingStrategy(&core, init_tuple_count));
			break;
		case MultiplexerRouting::INIT_ONCE:
			routing_strategy.reset(new InitOnceRoutingStrategy(&core, init_tuple_count));
			break;
		case MultiplexerRouting::OPPORTUNISTIC:
			routing_strategy.reset(new OpportunisticRoutingStrategy(&core, init_tuple_count));
			break;
		case MultiplexerRouting::DEFAULT_PATH:
		case MultiplexerRouting::BACKPRESSURE:
			routing_strategy.reset(new DefaultPathRoutingStrategy(&core, 0));
			break;
		case MultiplexerRouting::EXPONENTIAL_BACKOFF:
			routing_strategy.reset(new ExponentialBackoffRoutingStrategy(&core, init_tuple_count));
			break;
		default:
			throw InternalException("unknown routing strategy");
		}
	}
	polr::MultiplexerCore core; // path_resistances, historic_resistances, counters, strategy state
	unique_ptr<RoutingStrategy> routing_strategy;
	vector<vector<idx_t>> intermediates_alternate_mode;
	vector<idx_t> intermediates_per_round;
};

unique_ptr<OperatorState> PhysicalMultiplexer::GetOperatorState(ExecutionContext &context) const {
	if (context.client.config.time_resistance) {
		// wall-clock resistances are nondeterministic by construction; the device path keeps the default
		throw NotImplementedException("time_resistance is outside the MI355X path");
	}
	return unique_ptr<OperatorState>(new MultiplexerState(path_count, routing, regret_budget,
	                                                      context.client.config.init_tuple_count,
	                                                      context.client.config.atc_multiplier));
}

idx_t &PhysicalMultiplexer::GetNumCacheFlushingSkips(OperatorState &state_p) const {
	return ((MultiplexerState &)state_p).core.num_cache_flushing_skips;
}

// physical_multiplexer.cpp:100-121
OperatorResultType PhysicalMultiplexer::Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
                                                GlobalOperatorState &gstate_p, OperatorState &state_p) const {
	auto &state = (MultiplexerState &)state_p;
	if (!state.core.first_mpx_run) {
		FinalizePathRun(state, context.client.config.log_tuples_routed);
	} else {
		state.core.first_mpx_run = 0;
		if (routing == MultiplexerRouting::ALTERNATE) {
			state.core.alternate_mode = 1;
			state.intermediates_alternate_mode = vector<vector<idx_t>>(path_count);
		}
	}
	auto result = state.routing_strategy->Route(input, chunk);
	state.core.current_path_tuple_count = chunk.size();
	state.core.current_path_idx = state.core.next_path_idx;
	state.core.num_cache_flushing_skips = state.core.rs_cache_skips;
	return result;
}

void PhysicalMultiplexer::IncreaseInputTupleCount(OperatorState &state_p, idx_t tuple_count) const {
	((MultiplexerState &)state_p).core.IncreaseInputTupleCount(tuple_count);
}

// physical_multiplexer.cpp:132-174
void PhysicalMultiplexer::FinalizePathRun(OperatorState &state_p, bool log_tuples_routed) const {
	auto &state = (MultiplexerState &)state_p;
	const idx_t path = state.core.current_path_idx;
	const idx_t closed = state.core.FinalizePathRun();
	if (log_tuples_routed) {
		state.intermediates_per_round.push_back(closed);
	}
	if (!state.intermediates_alternate_mode.empty()) {
		state.intermediates_alternate_mode[path].push_back(closed);
	}
}

idx_t PhysicalMultiplexer::GetCurrentPathIndex(OperatorState &state_p) const {
	return ((MultiplexerState &)state_p).core.current_path_idx;
}

void PhysicalMultiplexer::AddNumIntermediates(OperatorState &state_p, idx_t count) const {
	((MultiplexerState &)state_p).core.AddNumIntermediates(count);
}

// physical_multiplexer.cpp:186-192
void PhysicalMultiplexer::PrintStatistics(OperatorState &state_p) const {
	auto &state = (MultiplexerState &)state_p;
	std::cout << "Input tuple counts per path\n";
	for (idx_t i = 0; i < path_count; i++) {
		std::cout << i << ": " << state.core.input_tuple_count_per_path[i] << "\n";
	}
}

// physical_multiplexer.cpp:194-219: same two CSV shapes
void PhysicalMultiplexer::WriteLogToFile(OperatorState &state_p, std::ostream &file) const {
	auto &state = (MultiplexerState &)state_p;
	std::stringstream log;
	if (!state.intermediates_alternate_mode.empty()) {
		for (idx_t i = 0; i < state.intermediates_alternate_mode.size(); i++) {
			log << "path_" << i << ",";
		}
		log << "\n";
		for (idx_t i = 0; i < state.intermediates_alternate_mode.front().size(); i++) {
			for (idx_t j = 0; j < state.intermediates_alternate_mode.size(); j++) {
				log << state.intermediates_alternate_mode[j][i] << ",";
			}
			log << "\n";
		}
	} else {
		log << "intermediates\n";
		for (idx_t i = 0; i < state.intermediates_per_round.size(); i++) {
			log << state.intermediates_per_round[i] << "\n";
		}
	}
	file << log.str();
}

bool PhysicalMultiplexer::WasExecuted(OperatorState &state_p) const {
	auto &state = (MultiplexerState &)state_p;
	for (idx_t i = 0; i < path_count; i++) {
		if (state.core.input_tuple_count_per_path[i] > 0) {
			return true;
		}
	}
	return false;
}

const polr::MultiplexerCore &PhysicalMultiplexer::Core(OperatorState &state_p) const {
	return ((MultiplexerState &)state_p).core;
}
const vector<idx_t> &PhysicalMultiplexer::IntermediatesPerRound(OperatorState &state_p) const {
	return ((MultiplexerState &)state_p).intermediates_per_round;
}
const vector<vector<idx_t>> &PhysicalMultiplexer::IntermediatesAlternateMode(OperatorState &state_p) const {
	return ((MultiplexerState &)state_p).intermediates_alternate_mode;
}

} // namespace duckdb_polr
