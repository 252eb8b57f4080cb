// duckdb-polr_amd/csrc/polr_internal.h -- host-side objects behind the opaque handles of
// include/polr_hip.h, shared by polr_capi.hip and polr_mpx.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>
#include <vector>

#include "../../include/polr_hip.h"
#include "polr_device.h"
#include "polr_mpx_device.h"

// Shared ownership: every object created on a context (build sides, pipelines, outputs, multiplexers) holds a
// reference, so destroying them in any order -- also AFTER polr_ctx_destroy, which a garbage-collected binding does
// routinely -- never reads a freed context (it used to: hipSetDevice(<freed>->device) failed with "invalid device
// ordinal" and left that status in the thread's last-error slot for the next launch check to find).
struct polr_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	int n_cus = 256;
	std::string err;
	std::atomic<int> refs {1};
	bool closed = false; // polr_ctx_destroy was called: the stream is gone, the object lives on for its children
	polr_pool_tuning tuning {}; // polr_ctx_set_pool_tuning (all zero: defaults)
	// Pool launches in flight (polr_mpx.hip: order_pool_launch).  A pool launch is sized for the whole device -- or for the
	// share of it its flags declare -- and its waves stay on the device until the run is over; two full-size launches on
	// two streams would each hold part of the device with waves that wait for work only the rest of their own grid can
	// unblock, and the watchdog of whichever router waits longest would give its run up.  The library orders them.
	struct PoolLaunch {
		hipStream_t stream;
		hipEvent_t done;
		uint32_t share;
	};
	std::vector<PoolLaunch> pool_launches;
	std::vector<hipEvent_t> pool_events_free;
};

static inline polr_ctx *polr_ctx_retain(polr_ctx *ctx) {
	ctx->refs.fetch_add(1, std::memory_order_relaxed);
	return ctx;
}
static inline void polr_ctx_release(polr_ctx *ctx) {
	if (ctx && ctx->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
		delete ctx;
	}
}

struct OwnedCol {
	uint8_t *data = nullptr;
	uint8_t *valid = nullptr;
	uint32_t width = 0;
	uint32_t flags = 0;
	bool owned = true;
};

struct polr_ht {
	polr_ctx *ctx = nullptr;
	uint32_t n_keys = 0, n_payload = 0;
	KeyPack pack = {}; // composite keys in packed form (finalize_hash decides)
	uint32_t key_flags[POLR_MAX_KEYS] = {}; // POLR_KEY_* per key column (polr_ht_set_key_flags, before finalize)
	uint64_t n_rows_in = 0; // rows as uploaded (build row ids index these)
	uint64_t n_rows = 0;    // rows kept (NULL keys dropped)
	std::vector<OwnedCol> keys, payload;
	DevCol *keys_dev = nullptr;    // device array [n_keys]
	DevCol *payload_dev = nullptr; // device array [n_payload] the probe kernel reads (by build id)
	uint32_t kind = KIND_NONE;
	uint32_t key_signed = 0;
	// hash
	void *table = nullptr;
	uint64_t capacity = 0;
	uint32_t *rowids = nullptr;
	uint32_t sentinel_start = 0, sentinel_count = 0;
	uint64_t max_run = 0;
	// perfect
	int64_t min_value = 0, max_value = 0;
	uint64_t range = 0;
	uint32_t *bits = nullptr;
	uint32_t *idx_row = nullptr;
	std::vector<OwnedCol> pcols;
	std::vector<void *> heaps; // string heaps of VARCHAR payload columns (polr_ht_set_payload_heap), owned
	uint32_t is_dense = 0, has_null = 0;
	uint64_t device_bytes = 0;
};

struct polr_pipeline {
	polr_ctx *ctx = nullptr;
	uint32_t k = 0, n_paths = 0, n_probe_cols = 0;
	uint64_t n_probe_rows = 0;
	uint64_t n_tuples = 0;
	std::vector<OwnedCol> probe_cols;
	std::vector<void *> heaps; // string heaps of VARCHAR probe columns (polr_pipeline_set_probe_heap), owned
	DevCol *probe_cols_dev = nullptr;
	uint32_t *sel_dev = nullptr;
	bool sel_owned = false;
	// result of polr_pipeline_scan_filter: boundaries of the (non-empty) source chunks in selection positions
	uint64_t *scan_offsets_dev = nullptr;
	uint64_t scan_n_chunks = 0;
	uint32_t scan_vector_size = 0;
	// scan buffers, sized for the worst case and reused by every polr_pipeline_scan_filter call
	uint32_t *scan_sel = nullptr;
	unsigned long long *scan_packed = nullptr, *scan_sums = nullptr, *scan_totals = nullptr;
	uint64_t scan_cap_rows = 0, scan_cap_vec = 0;
	bool scan_valid = false;      // a scan result is installed (selection + chunk boundaries)
	uint64_t scan_generation = 0; // bumped by every scan: multiplexers must re-attach (polr_mpx_use_scan_chunks)
	std::vector<polr_ht *> hts;
	DevPipeline host_count, host_mat; // count-only (narrow tuples) and materialising (all ids) variants
	DevPipeline *dev_count = nullptr, *dev_mat = nullptr;
	StageDesc *stages_count = nullptr, *stages_mat = nullptr; // [n_paths][POLR_KMAX] each
	StageExt *stage_ext = nullptr; // extension records of both variants (those stages that have one)
	int blocks_per_cu_count = 0, blocks_per_cu_mat = 0;       // measured residency of the path kernel
	uint32_t wpb_count = 0, wpb_mat = 0;                      // waves per workgroup (4, or fewer when the LDS queues are wide)
	uint32_t flat_wpb = 0;                                    // flat pipelines: waves per workgroup of the flat pool kernel
	bool flat_emit = false;                                   // ... and every join is a perfect table: emitting runs take it too
	// launch scratch (grown on demand)
	DevRound *rounds_dev = nullptr;
	uint64_t *prefix_dev = nullptr;
	uint32_t *unit_sizes_dev = nullptr;
	uint32_t rounds_cap = 0;
	unsigned long long *counts_dev = nullptr;
	uint64_t counts_cap = 0;
	unsigned long long *shards_dev = nullptr; // [rounds][POLR_NSHARD][k] scratch of the path kernel
	uint64_t shards_cap = 0;
};

struct polr_out {
	polr_pipeline *pipe = nullptr;
	polr_ctx *ctx = nullptr; // (kept so that destroying the object never has to go through the pipeline)
	DevOut dev;
	uint64_t *chunk_base = nullptr; // [max_chunks] exclusive prefix, refreshed by stats
	uint64_t *total_dev = nullptr;
	uint64_t n_rows = 0;
	uint32_t n_chunks = 0;
	bool stats_valid = false;
	// a fused GROUP BY sink (polr_out_fuse_grouped): the device descriptor, its cell tables, and what the result needs
	FusedSink *fused_dev = nullptr;
	unsigned long long *fused_cells = nullptr, *fused_dropped = nullptr;
	uint32_t fused_tables = 0, fused_groups = 0, fused_aggs = 0;
	uint32_t fused_fn[8] = {};
	bool fused_has_valid[8] = {};
};

#define POLR_FAIL(ctx_, code_, ...)                                                                                    \
	do {                                                                                                               \
		char buf_[512];                                                                                                \
		snprintf(buf_, sizeof(buf_), __VA_ARGS__);                                                                     \
		(ctx_)->err = buf_;                                                                                            \
		return (code_);                                                                                                \
	} while (0)

#define HIPCHK(ctx_, call_)                                                                                            \
	do {                                                                                                               \
		hipError_t e_ = (call_);                                                                                       \
		if (e_ != hipSuccess) {                                                                                        \
			POLR_FAIL(ctx_, POLR_E_HIP, "%s failed: %s (%s:%d)", #call_, hipGetErrorString(e_), __FILE__, __LINE__);   \
		}                                                                                                              \
	} while (0)

// Diagnostic (POLR_DEBUG_HIP_ERRORS=1 in the environment): every C-ABI entry point reports a HIP error that an
// EARLIER call left in the thread's last-error slot, naming the entry point that ran before -- how a swallowed
// HIP status is tracked down.  Off: one predictable branch.
void polr_trace_stale(const char *where);
#define POLR_ENTRY() polr_trace_stale(__func__)

static inline hipStream_t polr_stream(polr_ctx *ctx, void *stream) {
	return stream ? (hipStream_t)stream : ctx->stream;
}

// kernels / launchers implemented in polr_build.hip and polr_probe.hip
size_t polr_path_lds_bytes(uint32_t k, uint32_t W, uint32_t waves_per_block);
int polr_path_occupancy(uint32_t k, uint32_t W, uint32_t waves_per_block);
hipError_t polr_launch_path_kernel(uint32_t W, uint32_t k, uint32_t n_blocks, uint32_t waves_per_block,
                                   hipStream_t stream, const DevPipeline *pipe, const DevRound *rounds,
                                   const uint64_t *unit_prefix, uint32_t n_rounds, const uint32_t *unit_sizes,
                                   DevOut out, unsigned long long *counts, SelfRoute sr);
// the whole run in one launch (polr_pool.hip): routers + a pool of probe waves
struct PoolRun;
uint32_t polr_pool_waves_per_block(uint32_t k, uint32_t W); // 0: does not fit at all
size_t polr_pool_lds_bytes(uint32_t k, uint32_t W);
int polr_pool_occupancy(uint32_t k, uint32_t W, bool ext);
hipError_t polr_launch_pool_kernel(uint32_t W, uint32_t k, uint32_t n_blocks, hipStream_t stream, const DevPipeline *pipe,
                                   const ResidentExec *execs, PoolRun *run, DevOut out, bool ext);
size_t polr_pool_flat_lds_bytes(uint32_t k, uint32_t waves_per_block, uint32_t table_dwords);
size_t polr_pool_flat_wave_bytes(uint32_t k);
int polr_pool_flat_occupancy(uint32_t k, uint32_t waves_per_block, uint32_t table_dwords, bool emit);
hipError_t polr_launch_pool_flat_kernel(uint32_t k, uint32_t n_blocks, uint32_t waves_per_block, uint32_t table_dwords,
                                        hipStream_t stream, const DevPipeline *pipe, const ResidentExec *execs,
                                        PoolRun *run, DevOut out, bool emit, uint32_t fused_words);
void polr_launch_gather(hipStream_t stream, DevOut out, const uint64_t *chunk_base, uint32_t n_chunks, uint32_t slot,
                        DevCol src, uint8_t *dst_data, uint8_t *dst_valid);
void polr_launch_compact_ids(hipStream_t stream, DevOut out, const uint64_t *chunk_base, uint32_t n_chunks,
                             uint32_t *dst);
void polr_launch_reduce_counts(hipStream_t stream, const unsigned long long *src, uint64_t n_rounds, uint32_t k,
                               unsigned long long *dst);
void polr_launch_deserialize_col(hipStream_t st, const uint8_t *rows, uint64_t n_rows, uint32_t row_width, uint32_t col,
                                 uint32_t offset, uint32_t width, uint8_t *dst, uint8_t *dst_valid);
void polr_launch_key_minmax(hipStream_t st, const DevCol *keys_dev, uint32_t n_keys, uint64_t n_rows, uint32_t null_eq,
                            long long *out);
void polr_launch_s16_build(hipStream_t st, const DevCol *keys_dev, uint32_t n_keys, const KeyPack &pack, uint64_t n_rows,
                           uint4 *slots,
                           uint64_t capacity, uint32_t *slot_of_row, uint32_t *cursor, uint32_t *rowids,
                           uint32_t *block_sums, uint32_t *scalars, unsigned long long *n_valid);
void polr_launch_s16_scatter(hipStream_t st, uint64_t n_rows, const uint4 *slots, const uint32_t *slot_of_row,
                             uint32_t *cursor, uint32_t *rowids, uint32_t sentinel_start, uint32_t *sentinel_cursor);
void polr_launch_s16_to_s8(hipStream_t st, const uint4 *slots, uint64_t capacity, const uint32_t *rowids, uint2 *s8);
void polr_launch_pht_mark(hipStream_t st, const DevCol *keys_dev, uint64_t n_rows, int64_t min_value, uint64_t range,
                          uint32_t is_signed, uint32_t *bits, uint32_t *idx_row, uint32_t *flags,
                          unsigned long long *unique_keys);
void polr_launch_pht_gather(hipStream_t st, const uint32_t *bits, const uint32_t *idx_row, uint64_t size, DevCol src,
                            uint8_t *dst, uint8_t *dst_valid);
void polr_launch_pack_bitmap(hipStream_t st, const uint8_t *bytes, uint64_t size, uint32_t *bits);
void polr_launch_chunk_prefix(hipStream_t st, const uint32_t *chunk_count, uint32_t n_chunks, uint64_t *chunk_base,
                              uint64_t *total);

// shared between capi and mpx
int polr_plan_launch(polr_pipeline *p, bool materialize, uint64_t total_tuples, uint32_t *unit_size,
                     uint32_t *n_blocks_max);
uint32_t polr_resident_waves(polr_pipeline *p, bool materialize);
uint32_t polr_waves_per_block(polr_pipeline *p, bool materialize); // 0: does not fit at all
