#include <functional>
// duckdb-polr_amd/host/polr_host_capi.cpp -- a small C surface over the host mirror classes so that
// the Python test-suite (ctypes) can drive them: the multiplexer in isolation, POLARConfig's join-order
// generation + bindings, the chunk-at-a-time PhysicalHashJoin, and the batch POLARPipelineExecutor.
// Not part of the drop-in boundary (that is include/polr_hip.h); test plumbing only.
#include <cstring>
#include <sstream>

#include "polar_pipeline_executor.hpp"

using namespace duckdb_polr;

namespace {

struct HostMpx {
	ClientContext client;
	ThreadContext thread;
	std::unique_ptr<PhysicalMultiplexer> op;
	std::unique_ptr<OperatorState> state;
	DataChunk in, out;
};

thread_local std::string g_err;

// shape-only joins (no device): what POLARConfig needs for dependencies, enumeration and bindings
std::vector<std::unique_ptr<PhysicalHashJoin>> MakeShapeJoins(int k, int n_probe_cols, const int32_t *n_build_cols,
                                                              const int32_t *n_conds, const int32_t *cond_left_index,
                                                              const uint64_t *est_card, polr_ctx *ctx) {
	std::vector<std::unique_ptr<PhysicalHashJoin>> joins;
	idx_t width = (idx_t)n_probe_cols;
	for (int j = 0; j < k; j++) {
		vector<LogicalType> probe_types(width, LogicalType::INTEGER());
		vector<LogicalType> build_types((idx_t)n_build_cols[j], LogicalType::INTEGER());
		vector<JoinCondition> conds;
		vector<LogicalType> ctypes;
		for (int c = 0; c < n_conds[j]; c++) {
			JoinCondition jc;
			jc.left_index = (idx_t)cond_left_index[j * 2 + c];
			conds.push_back(jc);
			ctypes.push_back(LogicalType::INTEGER());
		}
		joins.emplace_back(new PhysicalHashJoin(ctx, probe_types, ctypes, build_types, conds, JoinType::INNER,
		                                        est_card ? est_card[j] : 0));
		width += (idx_t)n_build_cols[j];
	}
	return joins;
}

} // namespace

extern "C" {

const char *polr_host_last_error() {
	return g_err.c_str();
}

// ---- PhysicalMultiplexer in isolation ------------------------------------------------------------
void *polr_host_mpx_create(int n_paths, int routing, double regret_budget, uint64_t init_tuple_count,
                           uint64_t atc_multiplier, int log_tuples_routed) {
	try {
		auto *m = new HostMpx();
		m->client.config.init_tuple_count = init_tuple_count;
		m->client.config.atc_multiplier = atc_multiplier;
		m->client.config.log_tuples_routed = log_tuples_routed != 0;
		m->op.reset(new PhysicalMultiplexer(vector<LogicalType>(), 0, (idx_t)n_paths, regret_budget,
		                                    (MultiplexerRouting)routing));
		ExecutionContext ec(m->client, m->thread);
		m->state = m->op->GetOperatorState(ec);
		m->in.InitializeEmpty(vector<LogicalType>());
		m->out.InitializeEmpty(vector<LogicalType>());
		return m;
	} catch (std::exception &e) {
		g_err = e.what();
		return nullptr;
	}
}

void polr_host_mpx_destroy(void *h) {
	delete (HostMpx *)h;
}

// PhysicalMultiplexer::Execute on a chunk of `input_size` tuples; returns 1 for HAVE_MORE_OUTPUT
int polr_host_mpx_execute(void *h, uint64_t input_size, uint64_t *slice_offset, uint64_t *slice_count,
                          uint64_t *path, uint64_t *cache_skips) {
	auto *m = (HostMpx *)h;
	ExecutionContext ec(m->client, m->thread);
	m->in.SetCardinality(input_size);
	auto r = m->op->Execute(ec, m->in, m->out, *m->op->op_state, *m->state);
	const auto &core = m->op->Core(*m->state);
	const uint64_t cnt = m->out.size();
	uint64_t off = 0;
	if (cnt != input_size) {
		off = r == OperatorResultType::HAVE_MORE_OUTPUT ? core.chunk_offset - cnt : input_size - cnt;
	}
	*slice_offset = off;
	*slice_count = cnt;
	*path = m->op->GetCurrentPathIndex(*m->state);
	*cache_skips = m->op->GetNumCacheFlushingSkips(*m->state);
	return r == OperatorResultType::HAVE_MORE_OUTPUT ? 1 : 0;
}

void polr_host_mpx_add_intermediates(void *h, uint64_t n) {
	auto *m = (HostMpx *)h;
	m->op->AddNumIntermediates(*m->state, n);
}
void polr_host_mpx_increase_input(void *h, uint64_t n) {
	auto *m = (HostMpx *)h;
	m->op->IncreaseInputTupleCount(*m->state, n);
}
void polr_host_mpx_set_skips(void *h, uint64_t n) {
	auto *m = (HostMpx *)h;
	m->op->GetNumCacheFlushingSkips(*m->state) = n;
}
void polr_host_mpx_finalize_path_run(void *h) {
	auto *m = (HostMpx *)h;
	m->op->FinalizePathRun(*m->state, m->client.config.log_tuples_routed);
}
void polr_host_mpx_resistances(void *h, double *out) {
	auto *m = (HostMpx *)h;
	const auto &core = m->op->Core(*m->state);
	for (uint32_t i = 0; i < core.path_count; i++) {
		out[i] = core.path_resistances[i];
	}
}
void polr_host_mpx_tuple_counts(void *h, uint64_t *out) {
	auto *m = (HostMpx *)h;
	const auto &core = m->op->Core(*m->state);
	for (uint32_t i = 0; i < core.path_count; i++) {
		out[i] = core.input_tuple_count_per_path[i];
	}
}
// WriteLogToFile into a caller buffer; returns the length needed
uint64_t polr_host_mpx_log(void *h, char *buf, uint64_t cap) {
	auto *m = (HostMpx *)h;
	std::stringstream ss;
	m->op->WriteLogToFile(*m->state, ss);
	const std::string s = ss.str();
	if (buf && cap) {
		const uint64_t n = std::min<uint64_t>(cap - 1, s.size());
		memcpy(buf, s.data(), n);
		buf[n] = 0;
	}
	return s.size() + 1;
}

void polr_host_join_path_weights(const double *costs, int n, double regret_budget, double *weights) {
	vector<double> c(costs, costs + n), w(weights, weights + n);
	CalculateJoinPathWeights(c, w, regret_budget);
	for (int i = 0; i < n; i++) {
		weights[i] = w[i];
	}
}

// ---- POLARConfig::GenerateJoinOrders: join orders + left_expression_bindings ----------------------
// cond_left_index[j*2 + c]: BoundReference index of condition c of join j in the original pipeline
// layout (probe columns, then each join's build columns).  bindings[(p*k + j)*2 + c] = rebound column or -1.
// returns the number of join orders, 0 when POLAR does not engage, -1 on error
// node_card / node_flags [k + 1] (may be NULL): what SelSampleEnumeration reads off the plan -- entry 0 = the
// pipeline's source, entry 1 + j = the build side of join j; flags bit 0 = predicate, bit 1 = unique
int polr_host_generate_join_orders_ex(int enumerator, int routing, int k, int n_probe_cols, const int32_t *n_build_cols,
                                      const int32_t *n_conds, const int32_t *cond_left_index, const uint64_t *est_card,
                                      int max_join_orders, int32_t *paths, int32_t *bindings, uint8_t *dependencies,
                                      const uint64_t *node_card, const uint8_t *node_flags, int32_t *routing_out);
// the same with NESTED build sides (a build side that is itself a join tree, JoinOrderNodeInfo::nested_join_order):
// n_nodes >= k + 1 entries; node_parent[i] = -1 for the first k + 1, else the index of the node whose nested join order
// node i belongs to (members in listing order: the nested tree's source first, then the build sides of its joins)
int polr_host_generate_join_orders_nested(int enumerator, int routing, int k, int n_probe_cols, const int32_t *n_build_cols,
                                          const int32_t *n_conds, const int32_t *cond_left_index, const uint64_t *est_card,
                                          int max_join_orders, int32_t *paths, int32_t *bindings, uint8_t *dependencies,
                                          int n_nodes, const uint64_t *node_card, const uint8_t *node_flags,
                                          const int32_t *node_parent, int32_t *routing_out);

int polr_host_generate_join_orders(int enumerator, int routing, int k, int n_probe_cols, const int32_t *n_build_cols,
                                   const int32_t *n_conds, const int32_t *cond_left_index, const uint64_t *est_card,
                                   int max_join_orders, int32_t *paths, int32_t *bindings, uint8_t *dependencies) {
	return polr_host_generate_join_orders_ex(enumerator, routing, k, n_probe_cols, n_build_cols, n_conds, cond_left_index,
	                                         est_card, max_join_orders, paths, bindings, dependencies, nullptr, nullptr,
	                                         nullptr);
}

int polr_host_generate_join_orders_ex(int enumerator, int routing, int k, int n_probe_cols, const int32_t *n_build_cols,
                                      const int32_t *n_conds, const int32_t *cond_left_index, const uint64_t *est_card,
                                      int max_join_orders, int32_t *paths, int32_t *bindings, uint8_t *dependencies,
                                      const uint64_t *node_card, const uint8_t *node_flags, int32_t *routing_out) {
	return polr_host_generate_join_orders_nested(enumerator, routing, k, n_probe_cols, n_build_cols, n_conds, cond_left_index,
	                                             est_card, max_join_orders, paths, bindings, dependencies, k + 1, node_card,
	                                             node_flags, nullptr, routing_out);
}

int polr_host_generate_join_orders_nested(int enumerator, int routing, int k, int n_probe_cols, const int32_t *n_build_cols,
                                          const int32_t *n_conds, const int32_t *cond_left_index, const uint64_t *est_card,
                                          int max_join_orders, int32_t *paths, int32_t *bindings, uint8_t *dependencies,
                                          int n_nodes, const uint64_t *node_card, const uint8_t *node_flags,
                                          const int32_t *node_parent, int32_t *routing_out) {
	// routing_out (may be NULL): the routing the multiplexer ends up with -- DEFAULT_PATH when only Pipeline::Ready's
	// BFS_MIN_CARD fallback found a bank (pipeline.cpp:216-225); paths must hold 26 rows in that case (24 + 2)
	try {
		ClientContext client;
		client.config.join_enumerator = (JoinEnumerator)enumerator;
		client.config.multiplexer_routing = (MultiplexerRouting)routing;
		client.config.max_join_orders = (idx_t)max_join_orders;
		auto joins = MakeShapeJoins(k, n_probe_cols, n_build_cols, n_conds, cond_left_index, est_card, nullptr);
		JoinList raw;
		for (auto &j : joins) {
			raw.push_back(j.get());
		}
		if (node_card && node_flags) {
			if (n_nodes < k + 1) {
				throw InternalException("plan statistics for fewer nodes than the source and the joins' build sides");
			}
			std::function<JoinOrderNodeInfo(int)> info = [&](int i) {
				JoinOrderNodeInfo ni;
				ni.base_table_card = node_card[i];
				ni.predicate = (node_flags[i] & 1) != 0;
				ni.unique = (node_flags[i] & 2) != 0;
				for (int c = k + 1; node_parent && c < n_nodes; c++) {
					if (node_parent[c] == i) {
						if (c <= i) {
							throw InternalException("a nested plan node must be listed behind its parent");
						}
						ni.nested_join_order.push_back(info(c));
					}
				}
				return ni;
			};
			raw.front()->probe_source_info = info(0);
			for (int j = 0; j < k; j++) {
				raw[j]->build_side_info = info(1 + j);
			}
		} else if ((JoinEnumerator)enumerator == JoinEnumerator::SAMPLE) {
			throw InternalException("join_enumerator 'sample' needs the base-table cardinalities / predicate / unique "
			                            "flags of the source and of every build side");
		}
		std::unique_ptr<POLARConfig> polar_p = MakePolarConfigForPipeline(client, raw, 0);
		if (!polar_p) {
			return 0;
		}
		POLARConfig &polar = *polar_p;
		if (routing_out) {
			*routing_out = (int32_t)polar.multiplexer->routing;
		}
		const int P = (int)polar.join_paths.size();
		for (int p = 0; p < P; p++) {
			for (int j = 0; j < k; j++) {
				paths[p * k + j] = (int32_t)polar.join_paths[p][j];
				for (int c = 0; c < 2; c++) {
					auto &b = polar.left_expression_bindings[p][j];
					auto it = b.find((idx_t)c);
					bindings[(p * k + j) * 2 + c] = it == b.end() ? -1 : (int32_t)it->second;
				}
			}
		}
		if (dependencies) {
			memset(dependencies, 0, (size_t)k * k);
			for (auto &kv : polar.join_prerequisites) {
				for (auto d : kv.second) {
					dependencies[kv.first * k + d] = 1;
				}
			}
		}
		return P;
	} catch (std::exception &e) {
		g_err = e.what();
		return -1;
	}
}

// ---- batch executor on a device pipeline -----------------------------------------------------------
struct HostRunResult {
	uint64_t num_intermediates;
	uint64_t n_rounds;
	uint64_t input_tuple_count_per_path[POLR_MAX_PATHS];
	double path_resistances[POLR_MAX_PATHS];
};

// placement: 0 = host-routed (PhysicalMultiplexer on the host, one launch per path run), 1 = device-routed.
// rounds_* (capacity max_rounds) receive the per-round trace.  Returns 0 or -1 (polr_host_last_error).
int polr_host_run_pipeline(polr_ctx *ctx, polr_pipeline *pipe, int k, int n_paths, const int32_t *paths, int routing,
                           double regret_budget, uint64_t init_tuple_count, uint64_t atc_multiplier,
                           uint64_t n_tuples, const uint64_t *chunk_offsets, uint64_t n_chunks, int placement,
                           polr_out *out, HostRunResult *res, uint32_t *rounds_path, uint64_t *rounds_tuples,
                           uint64_t *rounds_inter, uint64_t max_rounds) {
	try {
		ClientContext client;
		client.config.multiplexer_routing = (MultiplexerRouting)routing;
		client.config.regret_budget = regret_budget;
		client.config.init_tuple_count = init_tuple_count;
		client.config.atc_multiplier = atc_multiplier;
		// a POLARConfig whose join orders are the ones the device pipeline was created with
		vector<int32_t> nb(k, 0), nc(k, 1), li(k * 2, 0);
		auto joins = MakeShapeJoins(k, 1, nb.data(), nc.data(), li.data(), nullptr, ctx);
		JoinList raw;
		for (auto &j : joins) {
			raw.push_back(j.get());
		}
		POLARConfig polar(client, raw, n_tuples, std::unique_ptr<JoinEnumerationAlgo>(new JoinEnumerationAlgo()));
		for (int p = 0; p < n_paths; p++) {
			polar.join_paths.emplace_back(paths + p * k, paths + (p + 1) * k);
		}
		double budget = regret_budget;
		if ((MultiplexerRouting)routing == MultiplexerRouting::EXPONENTIAL_BACKOFF) {
			budget = n_tuples / 10240.0 / 10 / 1;
		}
		polar.multiplexer.reset(new PhysicalMultiplexer(vector<LogicalType>(), n_tuples, (idx_t)n_paths, budget,
		                                                (MultiplexerRouting)routing));
		vector<idx_t> offs;
		if (chunk_offsets) {
			offs.assign(chunk_offsets, chunk_offsets + n_chunks + 1);
		}
		POLARPipelineExecutor exec(client, polar, ctx, pipe, n_tuples, offs);
		exec.Execute(placement == 2 ? RoutingPlacement::DEVICE_RESIDENT
		                            : (placement ? RoutingPlacement::DEVICE_ROUTED : RoutingPlacement::HOST_ROUTED),
		             out);
		memset(res, 0, sizeof(*res));
		res->num_intermediates = exec.num_intermediates_produced;
		res->n_rounds = exec.intermediates_per_round.size();
		for (int p = 0; p < n_paths; p++) {
			res->input_tuple_count_per_path[p] = exec.input_tuple_count_per_path[p];
			res->path_resistances[p] = exec.path_resistances[p];
		}
		for (uint64_t i = 0; i < res->n_rounds && i < max_rounds; i++) {
			rounds_inter[i] = exec.intermediates_per_round[i];
			if (i < exec.path_per_round.size()) {
				rounds_path[i] = exec.path_per_round[i];
				rounds_tuples[i] = exec.tuples_per_round[i];
			}
		}
		return 0;
	} catch (std::exception &e) {
		g_err = e.what();
		return -1;
	}
}

// ---- chunk-at-a-time drop-in: PhysicalHashJoin::Execute over a whole probe column --------------------
// Builds the join from columns, then feeds the probe keys one STANDARD_VECTOR_SIZE chunk at a time
// through Execute() following the reference's result protocol; collects (probe row, payload cell) pairs.
// returns number of output rows or -1
int64_t polr_host_hash_join_probe(polr_ctx *ctx, const int32_t *build_keys, const int32_t *build_payload,
                                  uint64_t n_build, int perfect, int64_t pmin, int64_t pmax,
                                  const int32_t *probe_keys, const uint8_t *probe_valid, uint64_t n_probe,
                                  uint32_t *out_probe_row, int32_t *out_payload, uint64_t out_cap,
                                  uint64_t *n_execute_calls) {
	try {
		ClientContext client;
		ThreadContext thread;
		ExecutionContext ec(client, thread);
		PerfectHashJoinStats stats;
		stats.is_build_small = perfect != 0;
		stats.build_min = pmin;
		stats.build_max = pmax;
		PhysicalHashJoin join(ctx, {LogicalType::INTEGER()}, {LogicalType::INTEGER()}, {LogicalType::INTEGER()},
		                      {JoinCondition()}, JoinType::INNER, n_build, stats);
		Vector bk(LogicalType::INTEGER(), n_build ? n_build : 1), bp(LogicalType::INTEGER(), n_build ? n_build : 1);
		memcpy(bk.data, build_keys, n_build * 4);
		memcpy(bp.data, build_payload, n_build * 4);
		join.SinkBuildSide({bk}, {bp}, n_build);
		auto state = join.GetOperatorState(ec);
		uint64_t n_out = 0, calls = 0;
		for (uint64_t base = 0; base < n_probe; base += STANDARD_VECTOR_SIZE) {
			const idx_t n = std::min<idx_t>(STANDARD_VECTOR_SIZE, n_probe - base);
			DataChunk input, chunk;
			input.Initialize({LogicalType::INTEGER()});
			memcpy(input.data[0].data, probe_keys + base, n * 4);
			if (probe_valid) {
				input.data[0].EnsureValidity(STANDARD_VECTOR_SIZE);
				memcpy(input.data[0].validity, probe_valid + base, n);
			}
			input.SetCardinality(n);
			for (;;) {
				chunk.Initialize(join.types);
				auto r = join.Execute(ec, input, chunk, *join.op_state, *state);
				calls++;
				if (r == OperatorResultType::FINISHED) {
					break;
				}
				for (idx_t i = 0; i < chunk.size(); i++) {
					if (n_out < out_cap) {
						// the probe column of the result is a dictionary vector over the input buffer
						out_probe_row[n_out] = (uint32_t)(base + chunk.data[0].sel.get_index(i));
						memcpy(&out_payload[n_out], chunk.data[1].Cell(i), 4);
					}
					n_out++;
				}
				if (r == OperatorResultType::NEED_MORE_INPUT) {
					break;
				}
			}
		}
		if (n_execute_calls) {
			*n_execute_calls = calls;
		}
		return (int64_t)n_out;
	} catch (std::exception &e) {
		g_err = e.what();
		return -1;
	}
}

// The same with a second, non-equality condition: probe_keys = build_keys AND probe_other OP build_other
// (comparison: the reference's ExpressionType value, 26..30).  (probe row, payload cell) pairs as above.
int64_t polr_host_hash_join_probe_cond(polr_ctx *ctx, const int32_t *build_keys, const int32_t *build_other,
                                       const int32_t *build_payload, uint64_t n_build, int comparison,
                                       const int32_t *probe_keys, const int32_t *probe_other, uint64_t n_probe,
                                       uint32_t *out_probe_row, int32_t *out_payload, uint64_t out_cap) {
	try {
		ClientContext client;
		ThreadContext thread;
		ExecutionContext ec(client, thread);
		JoinCondition eq, other;
		eq.left_index = 0;
		other.left_index = 1;
		other.comparison = (ExpressionType)comparison;
		PhysicalHashJoin join(ctx, {LogicalType::INTEGER(), LogicalType::INTEGER()},
		                      {LogicalType::INTEGER(), LogicalType::INTEGER()}, {LogicalType::INTEGER()}, {eq, other},
		                      JoinType::INNER, n_build);
		const idx_t nb = n_build ? n_build : 1;
		Vector bk(LogicalType::INTEGER(), nb), bo(LogicalType::INTEGER(), nb), bp(LogicalType::INTEGER(), nb);
		memcpy(bk.data, build_keys, n_build * 4);
		memcpy(bo.data, build_other, n_build * 4);
		memcpy(bp.data, build_payload, n_build * 4);
		join.SinkBuildSide({bk, bo}, {bp}, n_build);
		auto state = join.GetOperatorState(ec);
		uint64_t n_out = 0;
		for (uint64_t base = 0; base < n_probe; base += STANDARD_VECTOR_SIZE) {
			const idx_t n = std::min<idx_t>(STANDARD_VECTOR_SIZE, n_probe - base);
			DataChunk input, chunk;
			input.Initialize({LogicalType::INTEGER(), LogicalType::INTEGER()});
			memcpy(input.data[0].data, probe_keys + base, n * 4);
			memcpy(input.data[1].data, probe_other + base, n * 4);
			input.SetCardinality(n);
			for (;;) {
				chunk.Initialize(join.types);
				auto r = join.Execute(ec, input, chunk, *join.op_state, *state);
				if (r == OperatorResultType::FINISHED) {
					break;
				}
				for (idx_t i = 0; i < chunk.size(); i++) {
					if (n_out < out_cap) {
						out_probe_row[n_out] = (uint32_t)(base + chunk.data[0].sel.get_index(i));
						memcpy(&out_payload[n_out], chunk.data[2].Cell(i), 4);
					}
					n_out++;
				}
				if (r == OperatorResultType::NEED_MORE_INPUT) {
					break;
				}
			}
		}
		return (int64_t)n_out;
	} catch (std::exception &e) {
		g_err = e.what();
		return -1;
	}
}

// One join whose key is compared by VALUE across types and / or with NULL = NULL: probe column INTEGER; build key BIGINT
// when cast != 0 (the condition's left side is then CAST(column), JoinCondition::left_is_cast) else INTEGER; comparison
// IS NOT DISTINCT FROM when null_equal != 0 (COMPARE_NOT_DISTINCT_FROM).  (probe row, payload cell) pairs.
int64_t polr_host_hash_join_probe_keysem(polr_ctx *ctx, const void *build_keys, const uint8_t *build_key_valid,
                                         const int32_t *build_payload, uint64_t n_build, int cast, int null_equal,
                                         const int32_t *probe_keys, const uint8_t *probe_valid, uint64_t n_probe,
                                         uint32_t *out_probe_row, int32_t *out_payload, uint64_t out_cap) {
	try {
		ClientContext client;
		ThreadContext thread;
		ExecutionContext ec(client, thread);
		JoinCondition cond;
		cond.left_index = 0;
		cond.left_is_cast = cast != 0;
		cond.comparison = null_equal ? ExpressionType::COMPARE_NOT_DISTINCT_FROM : ExpressionType::COMPARE_EQUAL;
		const LogicalType key_type = cast ? LogicalType::BIGINT() : LogicalType::INTEGER();
		PhysicalHashJoin join(ctx, {LogicalType::INTEGER()}, {key_type}, {LogicalType::INTEGER()}, {cond}, JoinType::INNER,
		                      n_build);
		const idx_t nb = n_build ? n_build : 1;
		Vector bk(key_type, nb), bp(LogicalType::INTEGER(), nb);
		memcpy(bk.data, build_keys, n_build * key_type.width);
		if (build_key_valid) {
			bk.EnsureValidity(nb);
			memcpy(bk.validity, build_key_valid, n_build);
		}
		memcpy(bp.data, build_payload, n_build * 4);
		join.SinkBuildSide({bk}, {bp}, n_build);
		auto state = join.GetOperatorState(ec);
		uint64_t n_out = 0;
		for (uint64_t base = 0; base < n_probe; base += STANDARD_VECTOR_SIZE) {
			const idx_t n = std::min<idx_t>(STANDARD_VECTOR_SIZE, n_probe - base);
			DataChunk input, chunk;
			input.Initialize({LogicalType::INTEGER()});
			memcpy(input.data[0].data, probe_keys + base, n * 4);
			if (probe_valid) {
				input.data[0].EnsureValidity(STANDARD_VECTOR_SIZE);
				memcpy(input.data[0].validity, probe_valid + base, n);
			}
			input.SetCardinality(n);
			for (;;) {
				chunk.Initialize(join.types);
				auto r = join.Execute(ec, input, chunk, *join.op_state, *state);
				if (r == OperatorResultType::FINISHED) {
					break;
				}
				for (idx_t i = 0; i < chunk.size(); i++) {
					if (n_out < out_cap) {
						out_probe_row[n_out] = (uint32_t)(base + chunk.data[0].sel.get_index(i));
						memcpy(&out_payload[n_out], chunk.data[1].Cell(i), 4);
					}
					n_out++;
				}
				if (r == OperatorResultType::NEED_MORE_INPUT) {
					break;
				}
			}
		}
		return (int64_t)n_out;
	} catch (std::exception &e) {
		g_err = e.what();
		return -1;
	}
}

} // extern "C"
