// tests/conformance/adapter_main.cpp -- TEST INFRASTRUCTURE.  Runs duckdb-polr_amd/host/duckdb_adapter/polr_duckdb_adapter.hpp
// against THE REFERENCE's own objects: a duckdb::JoinHashTable is built and finalized by the reference's code
// (JoinHashTable::Build / InitializePointerTable / Finalize, as PhysicalHashJoin::Finalize does), handed to the device
// through the adapter, and every probe DataChunk is answered twice -- by the reference (JoinHashTable::Probe +
// ScanStructure::Next) and by the device path (PolrStageProbeKeys + PolrFetchInnerJoin).  The two result row multisets
// (probe columns and build columns) must be equal, chunk by chunk.  Linked against oracle/_ref/libduckdb_ref.so (the
// reference compiled from its sources, oracle/ref_build.mk) and libpolr_hip.so; built by `make -f oracle/ref_build.mk
// adapter` in the build container, run on the GPU box by tests/test_conformance.py.
//
//   adapter_test [not_distinct]      exit code 0 = equal;  prints one line per case
#include "duckdb.hpp"
#include "duckdb/main/database.hpp"
#include "duckdb/planner/expression/bound_reference_expression.hpp"
#include "duckdb/planner/joinside.hpp"

#include "duckdb_adapter/polr_duckdb_adapter.hpp"

#include <algorithm>
#include <cstdio>
#include <random>
#include <tuple>

using namespace duckdb;

typedef std::tuple<int64_t, int64_t, int64_t, int64_t> Row; // probe id, probe key (or INT64_MIN for NULL), build payload, valid bits

static void Collect(DataChunk &chunk, std::vector<Row> &rows) {
	chunk.Flatten();
	for (idx_t i = 0; i < chunk.size(); i++) {
		int64_t v[3];
		int64_t valid = 0;
		for (idx_t c = 0; c < 3; c++) {
			auto val = chunk.GetValue(c, i);
			v[c] = val.IsNull() ? INT64_MIN : val.GetValue<int64_t>();
			valid |= (int64_t)(val.IsNull() ? 0 : 1) << c;
		}
		rows.emplace_back(v[0], v[1], v[2], valid);
	}
}

static int RunCase(polr_ctx *ctx, DatabaseInstance &db, bool not_distinct) {
	auto &bm = BufferManager::GetBufferManager(db);
	vector<JoinCondition> conditions;
	{
		JoinCondition cond;
		cond.left = make_unique<BoundReferenceExpression>(LogicalType::INTEGER, 1);
		cond.right = make_unique<BoundReferenceExpression>(LogicalType::INTEGER, 0);
		cond.comparison = not_distinct ? ExpressionType::COMPARE_NOT_DISTINCT_FROM : ExpressionType::COMPARE_EQUAL;
		conditions.push_back(move(cond));
	}
	JoinHashTable ht(bm, conditions, {LogicalType::BIGINT}, JoinType::INNER);
	std::mt19937 rng(7);
	const idx_t n_build = 9000, n_probe = 6500;
	// build side: repeated keys, some NULL keys and NULL payloads
	for (idx_t base = 0; base < n_build; base += STANDARD_VECTOR_SIZE) {
		const idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, n_build - base);
		DataChunk keys, payload;
		keys.Initialize(Allocator::DefaultAllocator(), {LogicalType::INTEGER});
		payload.Initialize(Allocator::DefaultAllocator(), {LogicalType::BIGINT});
		for (idx_t i = 0; i < n; i++) {
			const uint32_t r = rng();
			keys.SetValue(0, i, r % 41 == 0 ? Value(LogicalType::INTEGER) : Value::INTEGER((int32_t)(r % 3000) - 500));
			payload.SetValue(0, i, r % 53 == 0 ? Value(LogicalType::BIGINT) : Value::BIGINT((int64_t)(base + i) * 1000003));
		}
		keys.SetCardinality(n);
		payload.SetCardinality(n);
		ht.Build(keys, payload);
	}
	ht.InitializePointerTable();
	ht.Finalize(0, ht.GetBlockCollection().blocks.size(), false);
	polr_ht *dht = PolrUploadBuildSide(ctx, ht);
	polr_pipeline *pipe = PolrMakeProbePipeline(ctx, ht, dht);
	idx_t total = 0;
	for (idx_t base = 0; base < n_probe; base += STANDARD_VECTOR_SIZE) {
		const idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, n_probe - base);
		DataChunk left, keys;
		left.Initialize(Allocator::DefaultAllocator(), {LogicalType::INTEGER, LogicalType::INTEGER});
		keys.Initialize(Allocator::DefaultAllocator(), {LogicalType::INTEGER});
		for (idx_t i = 0; i < n; i++) {
			const uint32_t r = rng();
			left.SetValue(0, i, Value::INTEGER((int32_t)(base + i)));
			left.SetValue(1, i, r % 29 == 0 ? Value(LogicalType::INTEGER) : Value::INTEGER((int32_t)(r % 3400) - 700));
		}
		left.SetCardinality(n);
		keys.data[0].Reference(left.data[1]);
		keys.SetCardinality(n);
		// the reference
		std::vector<Row> want, got;
		{
			auto ss = ht.Probe(keys);
			for (;;) {
				DataChunk result;
				result.Initialize(Allocator::DefaultAllocator(), {LogicalType::INTEGER, LogicalType::INTEGER, LogicalType::BIGINT});
				ss->Next(keys, left, result);
				if (result.size() == 0) {
					break;
				}
				Collect(result, want);
			}
		}
		// the device, through the adapter
		{
			PolrStageProbeKeys(ctx, pipe, keys);
			std::vector<unique_ptr<DataChunk>> results;
			PolrFetchInnerJoin(ctx, pipe, ht, left, results);
			for (auto &c : results) {
				Collect(*c, got);
			}
		}
		std::sort(want.begin(), want.end());
		std::sort(got.begin(), got.end());
		if (want != got) {
			printf("%s: chunk at %llu: reference %zu rows, device %zu rows -- DIFFERENT\n", not_distinct ? "not_distinct" : "equal",
			       (unsigned long long)base, want.size(), got.size());
			return 1;
		}
		total += want.size();
	}
	printf("%s: %llu build rows, %llu probe rows, %llu result rows: device == reference on every chunk\n",
	       not_distinct ? "IS NOT DISTINCT FROM" : "=", (unsigned long long)ht.Count(), (unsigned long long)n_probe,
	       (unsigned long long)total);
	polr_pipeline_destroy(pipe);
	polr_ht_destroy(dht);
	return total > 1000 ? 0 : 1;
}

int main(int argc, char **argv) {
	polr_ctx *ctx = nullptr;
	if (polr_ctx_create(0, &ctx) != POLR_OK) {
		printf("no device\n");
		return 2;
	}
	DuckDB db(nullptr);
	int rc = 0;
	try {
		rc |= RunCase(ctx, *db.instance, false);
		rc |= RunCase(ctx, *db.instance, true);
	} catch (std::exception &e) {
		printf("exception: %s\n", e.what());
		rc = 3;
	}
	polr_ctx_destroy(ctx);
	return rc;
}
