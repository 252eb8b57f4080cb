// duckdb-polr_amd/host/duckdb_adapter/polr_duckdb_adapter.hpp -- the binding of INTEGRATION.md section 2 over the REAL
// duckdb:: types: compiled only inside the reference's tree (or with -I<reference>/src/include), never by the product
// build.  It shows -- and tests/conformance/adapter_main.cpp runs -- the three hand-overs a maintainer's patch consists of:
//
//   PolrUploadBuildSide   a finalized duckdb::JoinHashTable (src/include/duckdb/execution/join_hashtable.hpp:128-260) ->
//                         polr_ht: its pinned row blocks go to the device as they are (RowLayout: validity bytes, keys,
//                         payload, hash/next slot -- src/common/types/row_layout.cpp:19-80); the bucket array is not needed
//   PolrStageProbeKeys    the key vectors of a probe DataChunk (what PhysicalHashJoin::Execute resolves with its
//                         ExpressionExecutor, physical_hash_join.cpp:669-670) -> the pipeline's staging columns
//   PolrFetchInnerJoin    the matches of the chunk -> a result DataChunk shaped like ScanStructure::NextInnerJoin's
//                         (join_hashtable.cpp:531-565): the probe columns sliced by the match selection, the build
//                         columns gathered (RowOperations::Gather's part done on the device)
//
// Types covered: the constant-size integer types the path supports as keys and payload (TINYINT .. BIGINT, unsigned too).
#pragma once

#include "duckdb/common/types/data_chunk.hpp"
#include "duckdb/common/types/row_data_collection.hpp"
#include "duckdb/execution/join_hashtable.hpp"
#include "duckdb/storage/buffer_manager.hpp"

#include "polr_hip.h"

#include <cstring>
#include <string>
#include <vector>

namespace duckdb {

inline void PolrCheck(polr_ctx *ctx, int rc, const char *what) {
	if (rc != POLR_OK) {
		throw InternalException(std::string(what) + ": " + (ctx ? polr_last_error(ctx) : "polr error"));
	}
}

inline uint32_t PolrColFlags(const LogicalType &type) {
	switch (type.InternalType()) {
	case PhysicalType::INT8:
	case PhysicalType::INT16:
	case PhysicalType::INT32:
	case PhysicalType::INT64:
		return POLR_COL_SIGNED;
	default:
		return 0;
	}
}

// a finalized JoinHashTable's rows -> the device (keys first, then payload: condition_types then build_types)
inline polr_ht *PolrUploadBuildSide(polr_ctx *ctx, JoinHashTable &ht) {
	const RowLayout &layout = ht.layout;
	const auto &types = layout.GetTypes();
	const auto &offsets = layout.GetOffsets();
	const idx_t n_keys = ht.condition_types.size();
	const idx_t n_payload = ht.build_types.size();
	std::vector<uint32_t> col_offset, col_width, col_flags;
	for (idx_t c = 0; c < n_keys + n_payload; c++) {
		if (!TypeIsConstantSize(types[c].InternalType())) {
			throw NotImplementedException("MI355X path: variable-size column in the build side's row layout");
		}
		col_offset.push_back((uint32_t)offsets[c]);
		col_width.push_back((uint32_t)GetTypeIdSize(types[c].InternalType()));
		col_flags.push_back(PolrColFlags(types[c]));
	}
	const idx_t row_width = layout.GetRowWidth();
	std::vector<uint8_t> blob;
	blob.reserve(ht.Count() * row_width);
	const RowDataCollection &rows = ht.GetBlockCollection();
	for (auto &block : rows.blocks) { // (finalized: the rows never move again, join_hashtable.cpp:370-376)
		auto handle = ht.buffer_manager.Pin(block->block);
		const data_ptr_t base = handle.Ptr();
		blob.insert(blob.end(), base, base + block->count * row_width);
	}
	polr_ht *out = nullptr;
	PolrCheck(ctx,
	          polr_ht_upload_rows(ctx, blob.data(), ht.Count(), (uint32_t)row_width, col_offset.data(), col_width.data(),
	                              col_flags.data(), (uint32_t)n_keys, (uint32_t)n_payload, &out),
	          "polr_ht_upload_rows");
	for (idx_t c = 0; c < ht.predicates.size() && c < n_keys; c++) {
		if (ht.predicates[c] == ExpressionType::COMPARE_NOT_DISTINCT_FROM) { // JoinHashTable::null_values_are_equal
			PolrCheck(ctx, polr_ht_set_key_flags(out, (uint32_t)c, POLR_KEY_NULL_EQUAL), "polr_ht_set_key_flags");
		}
	}
	PolrCheck(ctx, polr_ht_finalize_hash(out, nullptr), "polr_ht_finalize_hash");
	return out;
}

// one vector of a chunk as contiguous cells + validity bytes (the chunk may hold dictionary / constant vectors)
inline void PolrFlatten(Vector &v, idx_t count, std::vector<uint8_t> &cells, std::vector<uint8_t> &valid) {
	UnifiedVectorFormat f;
	v.ToUnifiedFormat(count, f);
	const idx_t w = GetTypeIdSize(v.GetType().InternalType());
	cells.assign(count * w + 1, 0);
	valid.assign(count + 1, 1);
	for (idx_t i = 0; i < count; i++) {
		const idx_t idx = f.sel->get_index(i);
		memcpy(cells.data() + i * w, f.data + idx * w, w);
		valid[i] = f.validity.RowIsValid(idx) ? 1 : 0;
	}
}

// the one-join pipeline a chunk-at-a-time operator probes through: staging columns of one vector, typed like the keys
inline polr_pipeline *PolrMakeProbePipeline(polr_ctx *ctx, JoinHashTable &ht, polr_ht *dht) {
	const idx_t n_keys = ht.condition_types.size();
	std::vector<std::vector<uint8_t>> zeros;
	std::vector<uint8_t> ones(STANDARD_VECTOR_SIZE, 1); // (a column that will carry NULLs is created with a validity array)
	std::vector<polr_col> cols;
	for (idx_t c = 0; c < n_keys; c++) {
		const idx_t w = GetTypeIdSize(ht.condition_types[c].InternalType());
		zeros.emplace_back(STANDARD_VECTOR_SIZE * w, 0);
		polr_col pc;
		pc.data = zeros.back().data();
		pc.valid = ones.data();
		pc.width = (uint32_t)w;
		pc.flags = PolrColFlags(ht.condition_types[c]);
		cols.push_back(pc);
	}
	polr_join_desc jd;
	memset(&jd, 0, sizeof(jd));
	jd.ht = dht;
	jd.n_keys = (uint32_t)n_keys;
	for (idx_t c = 0; c < n_keys; c++) {
		jd.key_src_join[c] = -1;
		jd.key_src_col[c] = (int32_t)c;
	}
	int32_t path = 0;
	polr_pipeline *pipe = nullptr;
	PolrCheck(ctx, polr_pipeline_create(ctx, cols.data(), (uint32_t)cols.size(), STANDARD_VECTOR_SIZE, &jd, 1, &path, 1, &pipe),
	          "polr_pipeline_create");
	return pipe;
}

inline void PolrStageProbeKeys(polr_ctx *ctx, polr_pipeline *pipe, DataChunk &keys) {
	std::vector<uint8_t> cells, valid;
	for (idx_t c = 0; c < keys.ColumnCount(); c++) {
		PolrFlatten(keys.data[c], keys.size(), cells, valid);
		PolrCheck(ctx, polr_pipeline_update_probe(pipe, (uint32_t)c, cells.data(), valid.data(), keys.size()),
		          "polr_pipeline_update_probe");
	}
}

// all matches of the staged chunk: `left` sliced by the probe side of every match + the build columns, appended to
// `results` in chunks of at most STANDARD_VECTOR_SIZE rows (what repeated NextInnerJoin calls hand to the pipeline)
inline idx_t PolrFetchInnerJoin(polr_ctx *ctx, polr_pipeline *pipe, JoinHashTable &ht, DataChunk &left,
                                std::vector<unique_ptr<DataChunk>> &results) {
	polr_round round;
	round.begin = 0;
	round.count = left.size();
	round.path = 0;
	round.emit = 1;
	uint64_t produced = 0, max_chunks = 8192;
	polr_out *out = nullptr;
	for (;;) {
		PolrCheck(ctx, polr_out_create(pipe, 1024, max_chunks, &out), "polr_out_create");
		int rc = polr_probe_rounds(pipe, nullptr, &round, 1, out, &produced);
		if (rc == POLR_E_OVERFLOW) {
			polr_out_destroy(out);
			max_chunks = produced / 1024 + 4096;
			continue;
		}
		PolrCheck(ctx, rc, "polr_probe_rounds");
		break;
	}
	uint64_t n_rows = 0, n_chunks = 0;
	uint32_t overflow = 0;
	PolrCheck(ctx, polr_out_stats(out, nullptr, &n_rows, &n_chunks, &overflow), "polr_out_stats");
	std::vector<uint32_t> ids(n_rows * 2 + 2);
	PolrCheck(ctx, polr_out_fetch_ids(out, nullptr, ids.data(), n_rows), "polr_out_fetch_ids");
	std::vector<std::vector<uint8_t>> cells(ht.build_types.size()), valid(ht.build_types.size());
	for (idx_t b = 0; b < ht.build_types.size(); b++) {
		const idx_t w = GetTypeIdSize(ht.build_types[b].InternalType());
		cells[b].assign(n_rows * w + 16, 0);
		valid[b].assign(n_rows + 1, 1);
		PolrCheck(ctx, polr_out_materialize(out, nullptr, 0, (uint32_t)b, cells[b].data(), valid[b].data(), n_rows, 0),
		          "polr_out_materialize");
	}
	polr_out_destroy(out);
	vector<LogicalType> types = left.GetTypes();
	types.insert(types.end(), ht.build_types.begin(), ht.build_types.end());
	for (idx_t base = 0; base < n_rows; base += STANDARD_VECTOR_SIZE) {
		const idx_t n = MinValue<idx_t>(STANDARD_VECTOR_SIZE, n_rows - base);
		auto chunk = make_unique<DataChunk>();
		chunk->Initialize(Allocator::DefaultAllocator(), types);
		SelectionVector sel(STANDARD_VECTOR_SIZE);
		for (idx_t i = 0; i < n; i++) {
			sel.set_index(i, ids[(base + i) * 2]);
		}
		chunk->Slice(left, sel, n); // (the probe columns: dictionary vectors over the input, join_hashtable.cpp:553)
		for (idx_t b = 0; b < ht.build_types.size(); b++) {
			Vector &v = chunk->data[left.ColumnCount() + b];
			const idx_t w = GetTypeIdSize(ht.build_types[b].InternalType());
			memcpy(FlatVector::GetData(v), cells[b].data() + base * w, n * w);
			for (idx_t i = 0; i < n; i++) {
				if (!valid[b][base + i]) {
					FlatVector::SetNull(v, i, true);
				}
			}
		}
		chunk->SetCardinality(n);
		results.push_back(move(chunk));
	}
	return n_rows;
}

} // namespace duckdb
