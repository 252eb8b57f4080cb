// duckdb-polr_amd/host/physical_hash_join.hpp -- host mirror of the probe side of PhysicalHashJoin
// (src/include/duckdb/execution/operator/join/physical_hash_join.hpp:24-111,
//  src/execution/operator/join/physical_hash_join.cpp:484-577,637-681) backed by a device-resident
// build side.  Same operator interface: GetOperatorState / GetOperatorStateWithBindings /
// Execute(input chunk -> output chunk) with the reference's result protocol (HAVE_MORE_OUTPUT while a
// probed chunk still has matches to hand out, then NEED_MORE_INPUT; FINISHED on an empty build side).
// There is no host probe: Execute goes through the C ABI (include/polr_hip.h) or throws.
#pragma once

#include "../../include/polr_hip.h"
#include "polr_host_types.hpp"

namespace duckdb_polr {

// the probe side of a JoinCondition: a BoundReferenceExpression on the probe chunk (joinside.hpp:17-36).  A CAST of one
// (`left_is_cast`): POLARConfig::GenerateJoinOrders means to accept it (polar_config.cpp:75-82) but tests for
// ExpressionType::CAST where a bound cast carries OPERATOR_CAST, so the reference never multiplexes a pipeline with a
// CAST'ed key; the mirror does the same (MakePolarConfigForPipeline returns "do not multiplex").  The operator itself
// takes such a key: integer casts are compared by value on the device (POLR_KEY_BY_VALUE), no cast copy is made
// the comparison of a join condition (values of src/include/duckdb/common/enums/expression_type.hpp:34-46)
enum class ExpressionType : uint8_t {
	COMPARE_EQUAL = 25,
	COMPARE_NOTEQUAL = 26,
	COMPARE_LESSTHAN = 27,
	COMPARE_GREATERTHAN = 28,
	COMPARE_LESSTHANOREQUALTO = 29,
	COMPARE_GREATERTHANOREQUALTO = 30,
	COMPARE_NOT_DISTINCT_FROM = 40 // (a key condition: NULL = NULL, JoinHashTable::null_values_are_equal join_hashtable.cpp:35-36)
};

// JoinCondition (src/include/duckdb/planner/joinside.hpp:22-41): left = a column of the probe chunk, right = a column
// the build side hands over at SinkBuildSide (condition c <-> keys[c]), comparison = how they are compared.  Equalities
// key the table; the others are evaluated on every candidate pair (JoinHashTable::predicates, join_hashtable.cpp:50-52).
struct JoinCondition {
	idx_t left_index = 0;
	bool left_is_bound_ref = true;
	bool left_is_cast = false; // CAST(BoundReference): left_index names the column under the cast
	ExpressionType comparison = ExpressionType::COMPARE_EQUAL;
};

// What SelSampleEnumeration reads off the plan below a join (JoinOrderNode, polar_enumeration_algo.hpp:72-83, filled by
// ExtractInfoLinear polar_enumeration_algo.cpp:190-246): the scanned base table's cardinality
// (DataTableInfo::cardinality, exact), whether the scan carries pushed-down table filters or sits under a FILTER
// (`predicate`), and whether a UNIQUE / PRIMARY KEY constraint covers a scanned column (`unique`).  The host mirror has no
// plan trees: the engine that owns them fills these in when it creates the operator.  nested_join_order: the build side
// is itself a join tree -- its source, then the build side of each of its joins, bottom up (CreateJoinOrderNodes recursing,
// polar_enumeration_algo.cpp:248-287); such a node's own cardinality is that of its nested order (:401-409).
struct JoinOrderNodeInfo {
	idx_t base_table_card = 0;
	bool predicate = false;
	bool unique = false;
	vector<JoinOrderNodeInfo> nested_join_order;
};

struct PerfectHashJoinStats {
	bool is_build_small = false; // plan_comparison_join.cpp:63-133
	int64_t build_min = 0, build_max = 0;
};

class PhysicalHashJoin : public PhysicalOperator {
public:
	PhysicalHashJoin(polr_ctx *ctx, vector<LogicalType> probe_types, vector<LogicalType> condition_types,
	                 vector<LogicalType> build_types, vector<JoinCondition> conditions, JoinType join_type,
	                 idx_t estimated_cardinality, PerfectHashJoinStats perfect_join_stats = PerfectHashJoinStats());
	~PhysicalHashJoin() override;

	polr_ctx *ctx;
	JoinType join_type;
	vector<JoinCondition> conditions;
	vector<LogicalType> probe_types, condition_types, build_types;
	vector<idx_t> right_projection_map;
	PerfectHashJoinStats perfect_join_statistics;
	idx_t uncertainty_level = 1; // what UncertainCardinalitySelector's plan walk would return
	JoinOrderNodeInfo build_side_info;   // children[1] of this join, for SelSampleEnumeration
	JoinOrderNodeInfo probe_source_info; // the pipeline's source below the join run (read from the FIRST join of the run)

	// ---- sink side, reduced to what the probe needs: hand the build columns over once -------------
	// (PhysicalHashJoin::Sink/Finalize physical_hash_join.cpp:217-286,337-481 -> HBM residency)
	void SinkBuildSide(const vector<Vector> &keys, const vector<Vector> &payload, idx_t count);
	polr_ht *hash_table = nullptr;
	// conditions by kind (indices into `conditions`): the equalities are the table's keys, the right sides of the others
	// ride as hidden payload columns behind the projected build columns
	vector<idx_t> equality_conditions, other_conditions;
	bool uses_perfect_hash = false;
	idx_t build_count = 0;

	// ---- operator interface ---------------------------------------------------------------------------
	unique_ptr<OperatorState> GetOperatorState(ExecutionContext &context) const override;
	unique_ptr<OperatorState> GetOperatorStateWithBindings(ExecutionContext &context,
	                                                       std::map<idx_t, idx_t> &bindings) const;
	OperatorResultType Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
	                           GlobalOperatorState &gstate, OperatorState &state) const override;
	bool ParallelOperator() const override {
		return true;
	}
	bool RequiresCache() const override {
		return true;
	}

private:
	unique_ptr<OperatorState> MakeState(const vector<idx_t> &key_columns) const;
};

} // namespace duckdb_polr
