// tests/conformance/check_signatures.cpp -- BUILD-CONTAINER-ONLY conformance check (tests/test_conformance.py).
//
// Compiles the reference's own operator headers (from /root/reference/src/include, read where they lie) TOGETHER with
// the host mirror's headers (duckdb-polr_amd/host, namespace duckdb_polr) and static_asserts that every member function
// the planner / pipeline executor calls on the POLAR operators has the same SHAPE on both sides: const-qualification,
// arity, value category and constness of every parameter, and the kind of the return type; that the result /
// routing enumerations carry the same values; and that the mirror's classes derive from their PhysicalOperator as the
// reference's do.  (The two sides live in different namespaces over different DataChunk / OperatorState types, so the
// member-function types cannot be *identical*; their shape is what a drop-in has to keep.)
//
//   reference interface                                                          file:line
//   PhysicalOperator::Execute / GetOperatorState / ParallelOperator / RequiresCache   physical_operator.hpp:130-160
//   PhysicalMultiplexer (+ FinalizePathRun, AddNumIntermediates, ...)                 polr/physical_multiplexer.hpp:27-48
//   PhysicalAdaptiveUnion (+ GetOperatorStateWithStaticJoinOrder)                     polr/physical_adaptive_union.hpp:21-33
//   PhysicalHashJoin (+ GetOperatorStateWithBindings)                                 join/physical_hash_join.hpp:61-73
#include "duckdb/execution/physical_operator.hpp"
#include "duckdb/execution/operator/polr/physical_multiplexer.hpp"
#include "duckdb/execution/operator/polr/physical_adaptive_union.hpp"
#include "duckdb/execution/operator/join/physical_hash_join.hpp"
#include "duckdb/common/enums/operator_result_type.hpp"
#include "duckdb/common/enums/join_enumerator.hpp"
#include "duckdb/main/config.hpp"

// (the reference's vector size is a macro, the mirror's a constant of the same name in its own namespace)
#undef STANDARD_VECTOR_SIZE

#include "physical_multiplexer.hpp"
#include "physical_adaptive_union.hpp"
#include "physical_hash_join.hpp"

#include <tuple>
#include <type_traits>

namespace conf {

template <class F>
struct sig;
template <class R, class C, class... A>
struct sig<R (C::*)(A...) const> {
	static constexpr bool is_const = true;
	static constexpr size_t arity = sizeof...(A);
	using ret = R;
	using args = std::tuple<A...>;
};
template <class R, class C, class... A>
struct sig<R (C::*)(A...)> {
	static constexpr bool is_const = false;
	static constexpr size_t arity = sizeof...(A);
	using ret = R;
	using args = std::tuple<A...>;
};

// value category + constness of a parameter (what a call site has to provide)
template <class T>
constexpr int shape() {
	using U = typename std::remove_reference<T>::type;
	return (std::is_lvalue_reference<T>::value ? 1 : 0) | (std::is_const<U>::value ? 2 : 0) |
	       (std::is_pointer<U>::value ? 4 : 0) | (std::is_arithmetic<U>::value ? 8 : 0) | (std::is_class<U>::value ? 16 : 0);
}
// kind of a return type
template <class T>
constexpr int ret_kind() {
	using U = typename std::remove_reference<T>::type;
	return (std::is_void<U>::value ? 1 : 0) | (std::is_enum<U>::value ? 2 : 0) | (std::is_arithmetic<U>::value ? 4 : 0) |
	       (std::is_class<U>::value ? 8 : 0) | (std::is_lvalue_reference<T>::value ? 16 : 0);
}

template <class TA, class TB, size_t... I>
constexpr bool same_arg_shapes(std::index_sequence<I...>) {
	bool ok = true;
	int dummy[] = {0, (ok = ok && shape<typename std::tuple_element<I, TA>::type>() ==
	                              shape<typename std::tuple_element<I, TB>::type>(),
	                   0)...};
	(void)dummy;
	return ok;
}

template <class FA, class FB>
constexpr bool same_shape() {
	using A = sig<FA>;
	using B = sig<FB>;
	if (A::is_const != B::is_const || A::arity != B::arity) {
		return false;
	}
	if (ret_kind<typename A::ret>() != ret_kind<typename B::ret>()) {
		return false;
	}
	return same_arg_shapes<typename A::args, typename B::args>(std::make_index_sequence<A::arity> {});
}

} // namespace conf

#define SAME(ref_member, our_member)                                                                                   \
	static_assert(conf::same_shape<decltype(&ref_member), decltype(&our_member)>(),                                    \
	              #our_member " does not have the shape of " #ref_member)

// ---- the virtual operator interface (physical_operator.hpp) -------------------------------------------------------
SAME(duckdb::PhysicalOperator::Execute, duckdb_polr::PhysicalOperator::Execute);
SAME(duckdb::PhysicalOperator::GetOperatorState, duckdb_polr::PhysicalOperator::GetOperatorState);
SAME(duckdb::PhysicalOperator::ParallelOperator, duckdb_polr::PhysicalOperator::ParallelOperator);
SAME(duckdb::PhysicalOperator::RequiresCache, duckdb_polr::PhysicalOperator::RequiresCache);

// ---- PhysicalMultiplexer ------------------------------------------------------------------------------------------
SAME(duckdb::PhysicalMultiplexer::Execute, duckdb_polr::PhysicalMultiplexer::Execute);
SAME(duckdb::PhysicalMultiplexer::GetOperatorState, duckdb_polr::PhysicalMultiplexer::GetOperatorState);
SAME(duckdb::PhysicalMultiplexer::ParallelOperator, duckdb_polr::PhysicalMultiplexer::ParallelOperator);
SAME(duckdb::PhysicalMultiplexer::RequiresCache, duckdb_polr::PhysicalMultiplexer::RequiresCache);
SAME(duckdb::PhysicalMultiplexer::FinalizePathRun, duckdb_polr::PhysicalMultiplexer::FinalizePathRun);
SAME(duckdb::PhysicalMultiplexer::AddNumIntermediates, duckdb_polr::PhysicalMultiplexer::AddNumIntermediates);
SAME(duckdb::PhysicalMultiplexer::IncreaseInputTupleCount, duckdb_polr::PhysicalMultiplexer::IncreaseInputTupleCount);
SAME(duckdb::PhysicalMultiplexer::GetCurrentPathIndex, duckdb_polr::PhysicalMultiplexer::GetCurrentPathIndex);
SAME(duckdb::PhysicalMultiplexer::GetNumCacheFlushingSkips, duckdb_polr::PhysicalMultiplexer::GetNumCacheFlushingSkips);
SAME(duckdb::PhysicalMultiplexer::WasExecuted, duckdb_polr::PhysicalMultiplexer::WasExecuted);
SAME(duckdb::PhysicalMultiplexer::PrintStatistics, duckdb_polr::PhysicalMultiplexer::PrintStatistics);
static_assert(std::is_base_of<duckdb::PhysicalOperator, duckdb::PhysicalMultiplexer>::value &&
                  std::is_base_of<duckdb_polr::PhysicalOperator, duckdb_polr::PhysicalMultiplexer>::value,
              "PhysicalMultiplexer is a PhysicalOperator on both sides");

// ---- PhysicalAdaptiveUnion ----------------------------------------------------------------------------------------
SAME(duckdb::PhysicalAdaptiveUnion::Execute, duckdb_polr::PhysicalAdaptiveUnion::Execute);
SAME(duckdb::PhysicalAdaptiveUnion::GetOperatorState, duckdb_polr::PhysicalAdaptiveUnion::GetOperatorState);
SAME(duckdb::PhysicalAdaptiveUnion::GetOperatorStateWithStaticJoinOrder,
     duckdb_polr::PhysicalAdaptiveUnion::GetOperatorStateWithStaticJoinOrder);
SAME(duckdb::PhysicalAdaptiveUnion::RequiresCache, duckdb_polr::PhysicalAdaptiveUnion::RequiresCache);
SAME(duckdb::PhysicalAdaptiveUnion::ParallelOperator, duckdb_polr::PhysicalAdaptiveUnion::ParallelOperator);
static_assert(std::is_base_of<duckdb_polr::PhysicalOperator, duckdb_polr::PhysicalAdaptiveUnion>::value, "base class");

// ---- PhysicalHashJoin (probe side) --------------------------------------------------------------------------------
SAME(duckdb::PhysicalHashJoin::Execute, duckdb_polr::PhysicalHashJoin::Execute);
SAME(duckdb::PhysicalHashJoin::GetOperatorState, duckdb_polr::PhysicalHashJoin::GetOperatorState);
SAME(duckdb::PhysicalHashJoin::GetOperatorStateWithBindings, duckdb_polr::PhysicalHashJoin::GetOperatorStateWithBindings);
SAME(duckdb::PhysicalHashJoin::ParallelOperator, duckdb_polr::PhysicalHashJoin::ParallelOperator);
SAME(duckdb::PhysicalHashJoin::RequiresCache, duckdb_polr::PhysicalHashJoin::RequiresCache);

// ---- enumerations that cross the boundary ---------------------------------------------------------------------------
#define SAME_ENUM(ref_e, our_e)                                                                                        \
	static_assert((int)(ref_e) == (int)(our_e), #our_e " differs from " #ref_e)
SAME_ENUM(duckdb::OperatorResultType::NEED_MORE_INPUT, duckdb_polr::OperatorResultType::NEED_MORE_INPUT);
SAME_ENUM(duckdb::OperatorResultType::HAVE_MORE_OUTPUT, duckdb_polr::OperatorResultType::HAVE_MORE_OUTPUT);
SAME_ENUM(duckdb::OperatorResultType::FINISHED, duckdb_polr::OperatorResultType::FINISHED);
SAME_ENUM(duckdb::MultiplexerRouting::ALTERNATE, duckdb_polr::MultiplexerRouting::ALTERNATE);
SAME_ENUM(duckdb::MultiplexerRouting::ADAPTIVE_REINIT, duckdb_polr::MultiplexerRouting::ADAPTIVE_REINIT);
SAME_ENUM(duckdb::MultiplexerRouting::DYNAMIC, duckdb_polr::MultiplexerRouting::DYNAMIC);
SAME_ENUM(duckdb::MultiplexerRouting::INIT_ONCE, duckdb_polr::MultiplexerRouting::INIT_ONCE);
SAME_ENUM(duckdb::MultiplexerRouting::OPPORTUNISTIC, duckdb_polr::MultiplexerRouting::OPPORTUNISTIC);
SAME_ENUM(duckdb::MultiplexerRouting::DEFAULT_PATH, duckdb_polr::MultiplexerRouting::DEFAULT_PATH);
SAME_ENUM(duckdb::MultiplexerRouting::BACKPRESSURE, duckdb_polr::MultiplexerRouting::BACKPRESSURE);
SAME_ENUM(duckdb::MultiplexerRouting::EXPONENTIAL_BACKOFF, duckdb_polr::MultiplexerRouting::EXPONENTIAL_BACKOFF);
SAME_ENUM(duckdb::JoinEnumerator::DFS_RANDOM, duckdb_polr::JoinEnumerator::DFS_RANDOM);
SAME_ENUM(duckdb::JoinEnumerator::BFS_MIN_CARD, duckdb_polr::JoinEnumerator::BFS_MIN_CARD);
SAME_ENUM(duckdb::JoinEnumerator::EACH_LAST_ONCE, duckdb_polr::JoinEnumerator::EACH_LAST_ONCE);
SAME_ENUM(duckdb::JoinEnumerator::SAMPLE, duckdb_polr::JoinEnumerator::SAMPLE);

#ifdef CONF_NEGATIVE
// (self-test of the checker: a pair that does NOT match must fail to compile)
SAME(duckdb::PhysicalMultiplexer::Execute, duckdb_polr::PhysicalMultiplexer::WasExecuted);
#endif

int main() {
	return 0;
}
