#include "physical_hash_join.hpp"

namespace duckdb_polr {

static void Check(polr_ctx *ctx, int rc, const char *what) {
	if (rc != POLR_OK) {
		throw InternalException(string(what) + ": " + polr_last_error(ctx));
	}
}

PhysicalHashJoin::PhysicalHashJoin(polr_ctx *ctx_p, vector<LogicalType> probe_types_p,
                                   vector<LogicalType> condition_types_p, vector<LogicalType> build_types_p,
                                   vector<JoinCondition> conditions_p, JoinType join_type_p,
                                   idx_t estimated_cardinality, PerfectHashJoinStats stats)
    : PhysicalOperator(PhysicalOperatorType::HASH_JOIN, vector<LogicalType>(), estimated_cardinality), ctx(ctx_p),
      join_type(join_type_p), conditions(std::move(conditions_p)), probe_types(std::move(probe_types_p)),
      condition_types(std::move(condition_types_p)), build_types(std::move(build_types_p)),
      perfect_join_statistics(stats) {
	// output = probe columns followed by the (projected) build columns, physical_join.cpp
	types = probe_types;
	types.insert(types.end(), build_types.begin(), build_types.end());
	if (conditions.size() != condition_types.size() || conditions.empty()) {
		throw InternalException("hash join needs one type per condition");
	}
	for (idx_t c = 0; c < conditions.size(); c++) {
		const bool keyed = conditions[c].comparison == ExpressionType::COMPARE_EQUAL ||
		                   conditions[c].comparison == ExpressionType::COMPARE_NOT_DISTINCT_FROM; // (join_hashtable.cpp:24-32)
		(keyed ? equality_conditions : other_conditions).push_back(c);
	}
	if (equality_conditions.empty()) {
		throw InternalException("a hash join needs at least one equality condition (plan_comparison_join.cpp)");
	}
	if (other_conditions.size() > POLR_MAX_PREDS) {
		throw NotImplementedException("more than " + std::to_string(POLR_MAX_PREDS) + " non-equality join conditions");
	}
}

PhysicalHashJoin::~PhysicalHashJoin() {
	if (hash_table) {
		polr_ht_destroy(hash_table);
	}
}

static polr_col ColOf(const Vector &v) {
	if (v.sel.sel_vector) {
		throw NotImplementedException("build columns must be flat vectors");
	}
	polr_col c;
	c.data = v.data;
	c.valid = v.validity;
	c.width = v.type.width;
	c.flags = v.type.is_signed ? POLR_COL_SIGNED : 0;
	return c;
}

void PhysicalHashJoin::SinkBuildSide(const vector<Vector> &keys, const vector<Vector> &payload, idx_t count) {
	if (!ctx) {
		throw InternalException("no device context: the MI355X path has no host fallback");
	}
	if (join_type != JoinType::INNER) {
		throw NotImplementedException("only INNER joins are multiplexed (polar_config.cpp:35)");
	}
	if (keys.size() != conditions.size()) {
		throw InternalException("SinkBuildSide: one build column per join condition");
	}
	vector<polr_col> kc, pc;
	for (auto c : equality_conditions) {
		kc.push_back(ColOf(keys[c]));
	}
	for (auto &p : payload) {
		pc.push_back(ColOf(p));
	}
	for (auto c : other_conditions) { // (hidden: never gathered into the result)
		pc.push_back(ColOf(keys[c]));
	}
	Check(ctx, polr_ht_upload_columns(ctx, kc.data(), (uint32_t)kc.size(), pc.data(), (uint32_t)pc.size(), count,
	                                  &hash_table),
	      "polr_ht_upload_columns");
	// key semantics: IS NOT DISTINCT FROM keeps NULL keys and lets them match; a CAST'ed probe key is compared by value
	bool plain_keys = true;
	for (idx_t i = 0; i < equality_conditions.size(); i++) {
		const JoinCondition &cond = conditions[equality_conditions[i]];
		const uint32_t flags = (cond.comparison == ExpressionType::COMPARE_NOT_DISTINCT_FROM ? POLR_KEY_NULL_EQUAL : 0u) |
		                       (cond.left_is_cast ? POLR_KEY_BY_VALUE : 0u);
		if (flags) {
			Check(ctx, polr_ht_set_key_flags(hash_table, (uint32_t)i, flags), "polr_ht_set_key_flags");
			plain_keys = false;
		}
	}
	build_count = count;
	uses_perfect_hash = false;
	// physical_hash_join.cpp:463-473: try the perfect table when the planner marked the build small;
	// a duplicate key falls back to the hash table (perfect_hash_join_executor.cpp:112-114)
	if (perfect_join_statistics.is_build_small && conditions.size() == 1 && other_conditions.empty() && plain_keys) {
		int rc = polr_ht_finalize_perfect(hash_table, perfect_join_statistics.build_min,
		                                  perfect_join_statistics.build_max, nullptr);
		if (rc == POLR_OK) {
			uses_perfect_hash = true;
			return;
		}
		if (rc != POLR_E_DUPLICATE) {
			Check(ctx, rc, "polr_ht_finalize_perfect");
		}
	}
	Check(ctx, polr_ht_finalize_hash(hash_table, nullptr), "polr_ht_finalize_hash");
}

// HashJoinOperatorState (physical_hash_join.cpp:484-506): the probe-key bindings plus what is left
// of the last probed chunk (the scan_structure of the reference)
class HashJoinOperatorState : public OperatorState {
public:
	~HashJoinOperatorState() override {
		if (out) {
			polr_out_destroy(out);
		}
		if (pipe) {
			polr_pipeline_destroy(pipe);
		}
	}
	vector<idx_t> key_columns; // probe chunk column of every condition (after rebinding)
	polr_pipeline *pipe = nullptr;
	polr_out *out = nullptr;
	// pending output of the chunk probed last
	vector<uint32_t> ids; // n x 2: (probe tuple, build row)
	vector<vector<uint8_t>> build_cells, build_valid;
	idx_t n_pending = 0, cursor = 0;
	bool has_scan_structure = false;
};

unique_ptr<OperatorState> PhysicalHashJoin::MakeState(const vector<idx_t> &key_columns) const {
	if (!hash_table) {
		throw InternalException("probe before the build side was sunk");
	}
	auto state = new HashJoinOperatorState();
	unique_ptr<OperatorState> holder(state);
	state->key_columns = key_columns;
	// a one-join pipeline over staging columns of one vector: the chunk-at-a-time drop-in
	vector<vector<uint8_t>> zeros;
	vector<polr_col> cols;
	vector<uint8_t> ones(STANDARD_VECTOR_SIZE, 1);
	for (idx_t c = 0; c < conditions.size(); c++) {
		// (a CAST'ed left side is staged in the probe column's OWN type: the device compares by value)
		const LogicalType &t = conditions[c].left_is_cast ? probe_types[conditions[c].left_index] : condition_types[c];
		zeros.emplace_back(STANDARD_VECTOR_SIZE * t.width, 0);
		polr_col pc;
		pc.data = zeros.back().data();
		pc.valid = ones.data();
		pc.width = t.width;
		pc.flags = t.is_signed ? POLR_COL_SIGNED : 0;
		cols.push_back(pc);
	}
	polr_join_desc jd;
	memset(&jd, 0, sizeof(jd));
	jd.ht = hash_table;
	jd.n_keys = (uint32_t)equality_conditions.size();
	for (idx_t i = 0; i < equality_conditions.size(); i++) {
		jd.key_src_join[i] = -1;
		jd.key_src_col[i] = (int32_t)equality_conditions[i]; // (staging column c holds the left side of condition c)
	}
	jd.n_preds = (uint32_t)other_conditions.size();
	for (idx_t i = 0; i < other_conditions.size(); i++) {
		uint32_t op;
		switch (conditions[other_conditions[i]].comparison) {
		case ExpressionType::COMPARE_NOTEQUAL:
			op = POLR_CMP_NE;
			break;
		case ExpressionType::COMPARE_LESSTHAN:
			op = POLR_CMP_LT;
			break;
		case ExpressionType::COMPARE_GREATERTHAN:
			op = POLR_CMP_GT;
			break;
		case ExpressionType::COMPARE_LESSTHANOREQUALTO:
			op = POLR_CMP_LE;
			break;
		case ExpressionType::COMPARE_GREATERTHANOREQUALTO:
			op = POLR_CMP_GE;
			break;
		default:
			throw NotImplementedException("join condition comparison outside =, <>, <, >, <=, >=");
		}
		jd.pred_op[i] = op;
		jd.pred_src_join[i] = -1;
		jd.pred_src_col[i] = (int32_t)other_conditions[i];
		jd.pred_build_col[i] = (uint32_t)(build_types.size() + i);
	}
	int32_t path = 0;
	Check(ctx, polr_pipeline_create(ctx, cols.data(), (uint32_t)cols.size(), STANDARD_VECTOR_SIZE, &jd, 1, &path, 1,
	                                &state->pipe),
	      "polr_pipeline_create");
	return holder;
}

unique_ptr<OperatorState> PhysicalHashJoin::GetOperatorState(ExecutionContext &context) const {
	vector<idx_t> key_columns;
	for (auto &cond : conditions) {
		key_columns.push_back(cond.left_index);
	}
	return MakeState(key_columns);
}

// physical_hash_join.cpp:541-577: copy of the condition with the BoundReference index overwritten
unique_ptr<OperatorState> PhysicalHashJoin::GetOperatorStateWithBindings(ExecutionContext &context,
                                                                         std::map<idx_t, idx_t> &bindings) const {
	vector<idx_t> key_columns;
	for (idx_t i = 0; i < conditions.size(); i++) {
		auto binding = bindings.find(i);
		key_columns.push_back(binding != bindings.end() ? binding->second : conditions[i].left_index);
	}
	return MakeState(key_columns);
}

static void EmitPending(const PhysicalHashJoin &join, HashJoinOperatorState &state, DataChunk &input, DataChunk &chunk) {
	const idx_t n = std::min<idx_t>(STANDARD_VECTOR_SIZE, state.n_pending - state.cursor);
	SelectionVector sel;
	sel.Initialize(n ? n : 1);
	for (idx_t i = 0; i < n; i++) {
		sel.set_index(i, state.ids[(state.cursor + i) * 2]);
	}
	// probe side: result.Slice(left, result_vector) (join_hashtable.cpp:555)
	chunk.Slice(input, sel, n);
	// build side: gathered cells (join_hashtable.cpp:558-562)
	for (idx_t b = 0; b < join.build_types.size(); b++) {
		Vector v(join.build_types[b], n ? n : 1);
		const idx_t w = join.build_types[b].width;
		memcpy(v.data, state.build_cells[b].data() + state.cursor * w, n * w);
		bool any_null = false;
		for (idx_t i = 0; i < n; i++) {
			any_null = any_null || !state.build_valid[b][state.cursor + i];
		}
		if (any_null) {
			v.EnsureValidity(n);
			memcpy(v.validity, state.build_valid[b].data() + state.cursor, n);
		}
		chunk.data[input.ColumnCount() + b] = v;
	}
	chunk.SetCardinality(n);
	state.cursor += n;
}

// physical_hash_join.cpp:637-681
OperatorResultType PhysicalHashJoin::Execute(ExecutionContext &context, DataChunk &input, DataChunk &chunk,
                                             GlobalOperatorState &gstate, OperatorState &state_p) const {
	auto &state = (HashJoinOperatorState &)state_p;
	if (build_count == 0) {
		return OperatorResultType::FINISHED; // EmptyResultIfRHSIsEmpty for INNER (:643-645)
	}
	if (state.has_scan_structure) {
		// still have elements remaining from the previous probe (:652-660)
		if (state.cursor < state.n_pending) {
			EmitPending(*this, state, input, chunk);
			return OperatorResultType::HAVE_MORE_OUTPUT;
		}
		state.has_scan_structure = false;
		chunk.SetCardinality(0);
		return OperatorResultType::NEED_MORE_INPUT;
	}
	const idx_t n = input.size();
	if (n > STANDARD_VECTOR_SIZE) {
		throw InternalException("input chunk larger than STANDARD_VECTOR_SIZE");
	}
	// resolve the join keys for the left chunk (:669-670): flatten through the dictionary selection
	for (idx_t c = 0; c < conditions.size(); c++) {
		const Vector &kv = input.data[state.key_columns[c]];
		if (!(kv.type == (conditions[c].left_is_cast ? probe_types[conditions[c].left_index] : condition_types[c]))) {
			throw InternalException("probe key type differs from the build key type and the condition carries no CAST");
		}
		const idx_t w = kv.type.width;
		vector<uint8_t> cells(n * w + 1), valid(n + 1, 1);
		for (idx_t i = 0; i < n; i++) {
			memcpy(cells.data() + i * w, kv.Cell(i), w);
			valid[i] = kv.IsValid(i) ? 1 : 0;
		}
		Check(ctx, polr_pipeline_update_probe(state.pipe, (uint32_t)c, cells.data(), valid.data(), n),
		      "polr_pipeline_update_probe");
	}
	// perform the actual probe (:677): all matches of the chunk in one launch
	polr_round round;
	round.begin = 0;
	round.count = n;
	round.path = 0;
	round.emit = 1;
	uint64_t produced = 0;
	uint64_t max_chunks = 8192; // every emitting wave owns a partially filled chunk: waves + outputs/1024
	for (;;) {
		if (!state.out) {
			Check(ctx, polr_out_create(state.pipe, 1024, max_chunks, &state.out), "polr_out_create");
		}
		Check(ctx, polr_out_reset(state.out, nullptr), "polr_out_reset");
		int rc = polr_probe_rounds(state.pipe, nullptr, &round, 1, state.out, &produced);
		if (rc == POLR_E_OVERFLOW) { // counters are exact: size the output for them and probe again
			polr_out_destroy(state.out);
			state.out = nullptr;
			max_chunks = produced / 1024 + 4096;
			continue;
		}
		Check(ctx, rc, "polr_probe_rounds");
		break;
	}
	uint64_t n_rows = 0, n_chunks = 0;
	uint32_t overflow = 0;
	Check(ctx, polr_out_stats(state.out, nullptr, &n_rows, &n_chunks, &overflow), "polr_out_stats");
	state.n_pending = n_rows;
	state.cursor = 0;
	state.ids.assign(n_rows * 2 + 2, 0);
	Check(ctx, polr_out_fetch_ids(state.out, nullptr, state.ids.data(), n_rows), "polr_out_fetch_ids");
	state.build_cells.assign(build_types.size(), vector<uint8_t>());
	state.build_valid.assign(build_types.size(), vector<uint8_t>());
	for (idx_t b = 0; b < build_types.size(); b++) {
		state.build_cells[b].assign(n_rows * build_types[b].width + 16, 0);
		state.build_valid[b].assign(n_rows + 1, 1);
		Check(ctx, polr_out_materialize(state.out, nullptr, 0, (uint32_t)b, state.build_cells[b].data(),
		                                state.build_valid[b].data(), n_rows, 0),
		      "polr_out_materialize");
	}
	if (uses_perfect_hash) {
		// ProbePerfectHashTable returns everything at once with NEED_MORE_INPUT
		// (perfect_hash_join_executor.cpp:177-207); at most one match per tuple so it fits one chunk
		EmitPending(*this, state, input, chunk);
		return OperatorResultType::NEED_MORE_INPUT;
	}
	state.has_scan_structure = true;
	EmitPending(*this, state, input, chunk);
	return OperatorResultType::HAVE_MORE_OUTPUT; // always, first call (:679-680)
}

} // namespace duckdb_polr
