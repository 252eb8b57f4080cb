// duckdb-polr_amd/csrc/polr_capi.hip -- implementation of the C ABI in include/polr_hip.h
// (contexts, build-side residency, pipelines, output chunks, probe launches).
// No CPU fallback anywhere in this file: every entry point either runs on the device or fails.
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include <stdlib.h>

#include "polr_internal.h"

void polr_trace_stale(const char *where) {
	static const int enabled = [] {
		const char *v = getenv("POLR_DEBUG_HIP_ERRORS");
		return (v && v[0] && v[0] != '0') ? 1 : 0;
	}();
	static thread_local const char *previous = "(none)";
	if (enabled) {
		const hipError_t e = hipGetLastError(); // (reads and clears)
		if (e != hipSuccess) {
			fprintf(stderr, "[polr] HIP error '%s' was pending on entry of %s; previous entry point: %s\n",
			        hipGetErrorString(e), where, previous);
		}
		previous = where;
	}
}

extern "C" {

int polr_abi_version(void) {
	return POLR_ABI_VERSION;
}

int polr_ctx_create(int device_id, polr_ctx **out) {
	POLR_ENTRY();
	if (!out) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) {
		return POLR_E_NO_DEVICE;
	}
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
		return POLR_E_NO_DEVICE;
	}
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		// the code objects in this library are gfx950 only
		return POLR_E_NO_DEVICE;
	}
	if (hipSetDevice(device_id) != hipSuccess) {
		return POLR_E_NO_DEVICE;
	}
	polr_ctx *ctx = new polr_ctx();
	ctx->device = device_id;
	ctx->n_cus = prop.multiProcessorCount;
	if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
		delete ctx;
		return POLR_E_HIP;
	}
	*out = ctx;
	return POLR_OK;
}

void polr_ctx_destroy(polr_ctx *ctx) {
	POLR_ENTRY();
	if (!ctx) {
		return;
	}
	if (ctx->closed) {
		return; // (destroyed twice: the object only lives on for its children)
	}
	hipSetDevice(ctx->device);
	hipStreamSynchronize(ctx->stream);
	for (auto &f : ctx->pool_launches) {
		hipEventDestroy(f.done);
	}
	for (auto &e : ctx->pool_events_free) {
		hipEventDestroy(e);
	}
	ctx->pool_launches.clear();
	ctx->pool_events_free.clear();
	hipStreamDestroy(ctx->stream);
	// objects created on this context may outlive it (they hold references): what they still do -- free their
	// device memory -- needs the device ordinal only; the default stream stands in for the destroyed one
	ctx->stream = nullptr;
	ctx->closed = true;
	polr_ctx_release(ctx);
}

const char *polr_last_error(const polr_ctx *ctx) {
	return ctx ? ctx->err.c_str() : "no context (is a gfx950 device visible?)";
}

int polr_ctx_sync(polr_ctx *ctx, void *stream) {
	POLR_ENTRY();
	if (!ctx) {
		return POLR_E_INVALID;
	}
	HIPCHK(ctx, hipStreamSynchronize(polr_stream(ctx, stream)));
	return POLR_OK;
}

int polr_ctx_get_stream(polr_ctx *ctx, void **stream) {
	POLR_ENTRY();
	if (!ctx || !stream) {
		return POLR_E_INVALID;
	}
	*stream = (void *)ctx->stream;
	return POLR_OK;
}

int polr_ctx_set_pool_tuning(polr_ctx *ctx, const polr_pool_tuning *t) {
	POLR_ENTRY();
	if (!ctx) {
		return POLR_E_INVALID;
	}
	if (!t) {
		ctx->tuning = polr_pool_tuning {};
		return POLR_OK;
	}
	if (t->device_share > 16) {
		POLR_FAIL(ctx, POLR_E_INVALID, "device share 1/%u: at most 16 runs side by side", t->device_share);
	}
	if (t->units_x > 4) {
		POLR_FAIL(ctx, POLR_E_INVALID, "units_x %u: 1..4 (ring capacities are sized for 4)", t->units_x);
	}
	if (t->hi_unit && (t->hi_unit < 64 || t->hi_unit > 1024 || t->hi_unit % 64)) {
		POLR_FAIL(ctx, POLR_E_INVALID, "hi_unit %u: 64..1024 in multiples of 64", t->hi_unit);
	}
	if (t->hi_lottery & (t->hi_lottery - 1)) {
		POLR_FAIL(ctx, POLR_E_INVALID, "hi_lottery %u: a power of two", t->hi_lottery);
	}
	if (t->idle_sleep && t->idle_sleep != 16 && t->idle_sleep != 64) {
		POLR_FAIL(ctx, POLR_E_INVALID, "idle_sleep %u: 16 or 64", t->idle_sleep);
	}
	if (t->share_after && t->share_after != 0xFFFFFFFFu && (t->share_after < 16 || t->share_after > 65535)) {
		POLR_FAIL(ctx, POLR_E_INVALID, "share_after %u: 16..65535 steps, or 0xFFFFFFFF for never", t->share_after);
	}
	ctx->tuning = *t;
	return POLR_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------
static bool valid_width(uint32_t w) {
	return w == 1 || w == 2 || w == 4 || w == 8 || w == 16;
}

static int dev_alloc(polr_ctx *ctx, void **p, uint64_t bytes, uint64_t *acct) {
	*p = nullptr;
	if (bytes == 0) {
		bytes = 16;
	}
	HIPCHK(ctx, hipMalloc(p, bytes));
	if (acct) {
		*acct += bytes;
	}
	return POLR_OK;
}

// copy (or alias) a caller column to the device
static int ingest_col(polr_ctx *ctx, const polr_col *src, uint64_t n_rows, OwnedCol *dst, uint64_t *acct,
                      hipStream_t st) {
	if (!valid_width(src->width)) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "column width %u not supported (1,2,4,8,16)", src->width);
	}
	if (!src->data && n_rows) {
		POLR_FAIL(ctx, POLR_E_INVALID, "column data pointer is NULL");
	}
	dst->width = src->width;
	dst->flags = src->flags & POLR_COL_SIGNED;
	if (src->flags & POLR_COL_DEVICE) {
		dst->data = (uint8_t *)src->data;
		dst->valid = (uint8_t *)src->valid;
		dst->owned = false;
		return POLR_OK;
	}
	dst->owned = true;
	int rc = dev_alloc(ctx, (void **)&dst->data, n_rows * src->width, acct);
	if (rc) {
		return rc;
	}
	if (n_rows) {
		HIPCHK(ctx, hipMemcpyAsync(dst->data, src->data, n_rows * src->width, hipMemcpyHostToDevice, st));
	}
	if (src->valid) {
		rc = dev_alloc(ctx, (void **)&dst->valid, n_rows, acct);
		if (rc) {
			return rc;
		}
		if (n_rows) {
			HIPCHK(ctx, hipMemcpyAsync(dst->valid, src->valid, n_rows, hipMemcpyHostToDevice, st));
		}
	}
	return POLR_OK;
}

static void free_col(OwnedCol &c) {
	if (c.owned) {
		if (c.data) {
			hipFree(c.data);
		}
		if (c.valid) {
			hipFree(c.valid);
		}
	}
	c.data = c.valid = nullptr;
}

static int upload_devcols(polr_ctx *ctx, const std::vector<OwnedCol> &cols, DevCol **dst, hipStream_t st) {
	std::vector<DevCol> h(cols.size() ? cols.size() : 1);
	for (size_t i = 0; i < cols.size(); i++) {
		h[i].data = cols[i].data;
		h[i].valid = cols[i].valid;
		h[i].width = cols[i].width;
		h[i].flags = cols[i].flags;
	}
	if (!*dst) {
		HIPCHK(ctx, hipMalloc((void **)dst, h.size() * sizeof(DevCol)));
	}
	HIPCHK(ctx, hipMemcpyAsync(*dst, h.data(), h.size() * sizeof(DevCol), hipMemcpyHostToDevice, st));
	HIPCHK(ctx, hipStreamSynchronize(st)); // h is a stack-lifetime staging buffer
	return POLR_OK;
}

static uint64_t next_pow2_u64(uint64_t v) {
	uint64_t p = 1;
	while (p < v) {
		p <<= 1;
	}
	return p;
}

static int check_key_shape(polr_ctx *ctx, const std::vector<OwnedCol> &keys) {
	if (keys.empty() || keys.size() > POLR_MAX_KEYS) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "%zu equality keys per join not supported (1..%d)", keys.size(),
		          POLR_MAX_KEYS);
	}
	for (auto &k : keys) {
		if (k.width > 8) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "VARCHAR join keys are outside this path");
		}
		if (k.width != 1 && k.width != 2 && k.width != 4 && k.width != 8) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "join key of %u bytes", k.width);
		}
	}
	return POLR_OK;
}

// composite keys outside the plain {key0 | key1 << 32} form are packed exactly (KeyPack)
static bool needs_key_pack(const polr_ht *ht) {
	if (ht->n_keys >= 3) {
		return true;
	}
	for (uint32_t c = 0; c < ht->n_keys; c++) {
		if (ht->key_flags[c]) {
			return true; // compared by value (a CAST on one side) or NULL = NULL: the packed form does both exactly
		}
	}
	return ht->n_keys == 2 && (ht->keys[0].width > 4 || ht->keys[1].width > 4);
}

extern "C" {

// ---------------------------------------------------------------------------------------------------
// Build sides
// ---------------------------------------------------------------------------------------------------
int polr_ht_upload_columns(polr_ctx *ctx, const polr_col *keys, uint32_t n_keys, const polr_col *payload,
                           uint32_t n_payload, uint64_t n_rows, polr_ht **out) {
	POLR_ENTRY();
	if (!ctx || !out || !keys) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	if (n_rows >= 0xFFFFFFF0ull) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "build side of %llu rows exceeds the 32-bit row-id space",
		          (unsigned long long)n_rows);
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	polr_ht *ht = new polr_ht();
	ht->ctx = polr_ctx_retain(ctx);
	ht->n_keys = n_keys;
	ht->n_payload = n_payload;
	ht->n_rows_in = n_rows;
	ht->keys.resize(n_keys);
	ht->payload.resize(n_payload);
	int rc = POLR_OK;
	for (uint32_t i = 0; i < n_keys && !rc; i++) {
		rc = ingest_col(ctx, &keys[i], n_rows, &ht->keys[i], &ht->device_bytes, ctx->stream);
	}
	for (uint32_t i = 0; i < n_payload && !rc; i++) {
		rc = ingest_col(ctx, &payload[i], n_rows, &ht->payload[i], &ht->device_bytes, ctx->stream);
	}
	if (!rc) {
		rc = check_key_shape(ctx, ht->keys);
	}
	if (!rc) {
		ht->key_signed = ht->keys[0].flags & POLR_COL_SIGNED;
		rc = upload_devcols(ctx, ht->keys, &ht->keys_dev, ctx->stream);
	}
	if (rc) {
		polr_ht_destroy(ht);
		return rc;
	}
	*out = ht;
	return POLR_OK;
}

int polr_ht_upload_rows(polr_ctx *ctx, const void *rows, uint64_t n_rows, uint32_t row_width,
                        const uint32_t *col_offset, const uint32_t *col_width, const uint32_t *col_flags,
                        uint32_t n_keys, uint32_t n_payload, polr_ht **out) {
	POLR_ENTRY();
	if (!ctx || !out || (!rows && n_rows) || !col_offset || !col_width) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	if (n_rows >= 0xFFFFFFF0ull) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "build side of %llu rows exceeds the 32-bit row-id space",
		          (unsigned long long)n_rows);
	}
	const uint32_t ncols = n_keys + n_payload;
	for (uint32_t c = 0; c < ncols; c++) {
		if (!valid_width(col_width[c]) || (uint64_t)col_offset[c] + col_width[c] > row_width) {
			POLR_FAIL(ctx, POLR_E_INVALID, "row layout: column %u (offset %u, width %u) does not fit row width %u", c,
			          col_offset[c], col_width[c], row_width);
		}
	}
	if ((ncols + 1 + 7) / 8 > row_width) {
		POLR_FAIL(ctx, POLR_E_INVALID, "row layout: validity bytes exceed row width");
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	polr_ht *ht = new polr_ht();
	ht->ctx = polr_ctx_retain(ctx);
	ht->n_keys = n_keys;
	ht->n_payload = n_payload;
	ht->n_rows_in = n_rows;
	ht->keys.resize(n_keys);
	ht->payload.resize(n_payload);
	uint8_t *blob = nullptr;
	int rc = dev_alloc(ctx, (void **)&blob, n_rows * row_width, nullptr);
	if (!rc && n_rows) {
		hipError_t e = hipMemcpyAsync(blob, rows, n_rows * row_width, hipMemcpyHostToDevice, ctx->stream);
		if (e != hipSuccess) {
			ctx->err = std::string("hipMemcpyAsync(rows) failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	for (uint32_t c = 0; c < ncols && !rc; c++) {
		OwnedCol &col = c < n_keys ? ht->keys[c] : ht->payload[c - n_keys];
		col.width = col_width[c];
		col.flags = col_flags ? (col_flags[c] & POLR_COL_SIGNED) : 0;
		col.owned = true;
		rc = dev_alloc(ctx, (void **)&col.data, n_rows * col.width, &ht->device_bytes);
		if (!rc) {
			// (key columns too: a table whose condition is IS NOT DISTINCT FROM keeps its NULL-key rows, join_hashtable.cpp:182)
			rc = dev_alloc(ctx, (void **)&col.valid, n_rows, &ht->device_bytes);
		}
		if (!rc) {
			polr_launch_deserialize_col(ctx->stream, blob, n_rows, row_width, c, col_offset[c], col.width, col.data,
			                            col.valid);
		}
	}
	if (!rc) {
		hipError_t e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) {
			ctx->err = std::string("row de-serialisation failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	if (blob) {
		hipFree(blob);
	}
	if (!rc) {
		rc = check_key_shape(ctx, ht->keys);
	}
	if (!rc) {
		ht->key_signed = ht->keys[0].flags & POLR_COL_SIGNED;
		rc = upload_devcols(ctx, ht->keys, &ht->keys_dev, ctx->stream);
	}
	if (rc) {
		polr_ht_destroy(ht);
		return rc;
	}
	*out = ht;
	return POLR_OK;
}

int polr_ht_set_key_flags(polr_ht *ht, uint32_t key_col, uint32_t flags) {
	POLR_ENTRY();
	if (!ht) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = ht->ctx;
	if (ht->kind != KIND_NONE) {
		POLR_FAIL(ctx, POLR_E_INVALID, "key flags are set before the table is finalized");
	}
	if (key_col >= ht->n_keys || (flags & ~(POLR_KEY_BY_VALUE | POLR_KEY_NULL_EQUAL))) {
		POLR_FAIL(ctx, POLR_E_INVALID, "key column %u of %u, flags 0x%x", key_col, ht->n_keys, flags);
	}
	ht->key_flags[key_col] = flags;
	return POLR_OK;
}

int polr_ht_finalize_hash(polr_ht *ht, void *stream) {
	POLR_ENTRY();
	if (!ht) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = ht->ctx;
	if (ht->kind != KIND_NONE) {
		POLR_FAIL(ctx, POLR_E_INVALID, "table already finalized");
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	const uint64_t n = ht->n_rows_in;
	ht->pack = KeyPack();
	if (needs_key_pack(ht)) {
		// per-column [min, max] of the build keys (rows with a NULL key never match and do not count)
		long long h_mm[2 * POLR_NKEYS];
		for (uint32_t c = 0; c < POLR_NKEYS; c++) {
			h_mm[2 * c] = 0x7FFFFFFFFFFFFFFFll;
			h_mm[2 * c + 1] = -0x7FFFFFFFFFFFFFFFll - 1;
		}
		long long *mm = nullptr;
		int rc0 = dev_alloc(ctx, (void **)&mm, sizeof(h_mm), nullptr);
		if (rc0) {
			return rc0;
		}
		hipError_t e = hipMemcpyAsync(mm, h_mm, sizeof(h_mm), hipMemcpyHostToDevice, st);
		if (e == hipSuccess) {
			uint32_t null_eq = 0;
			for (uint32_t c = 0; c < ht->n_keys; c++) {
				null_eq |= (ht->key_flags[c] & POLR_KEY_NULL_EQUAL) ? (1u << c) : 0u;
			}
			ht->pack.null_eq = null_eq;
			polr_launch_key_minmax(st, ht->keys_dev, ht->n_keys, n, null_eq, mm);
			e = hipMemcpyAsync(h_mm, mm, sizeof(h_mm), hipMemcpyDeviceToHost, st);
		}
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
		hipFree(mm);
		if (e != hipSuccess) {
			POLR_FAIL(ctx, POLR_E_HIP, "key range scan failed: %s", hipGetErrorString(e));
		}
		uint32_t shift = 0;
		const uint32_t null_eq_mask = ht->pack.null_eq;
		ht->pack.packed = 1;
		for (uint32_t c = 0; c < ht->n_keys; c++) {
			const bool empty = h_mm[2 * c] > h_mm[2 * c + 1]; // no row with valid keys at all
			const int64_t lo = empty ? 0 : h_mm[2 * c], hi = empty ? 0 : h_mm[2 * c + 1];
			const uint64_t range = (uint64_t)hi - (uint64_t)lo;
			// (a NULL = NULL column has one more code, range + 1: NULL)
			const bool null_eq_col = ((null_eq_mask >> c) & 1u) != 0;
			const uint64_t top = range + (null_eq_col ? 1u : 0u);
			uint32_t bits = 0;
			while (bits < 64 && (top >> bits) != 0) {
				bits++;
			}
			if (shift + bits > 64 || (null_eq_col && top == 0)) {
				ht->pack = KeyPack();
				POLR_FAIL(ctx, POLR_E_UNSUPPORTED,
				          "composite key of %u columns needs more than 64 bits (column %u: range %llu after %u bits)",
				          ht->n_keys, c, (unsigned long long)range, shift);
			}
			ht->pack.shift[c] = shift;
			ht->pack.sx[c] = (ht->keys[c].flags & 1u) ? 1u : 0u;
			ht->pack.min[c] = lo;
			ht->pack.range[c] = range;
			shift += bits;
		}
	}
	// load factor <= 0.5 like PointerTableCapacity (join_hashtable.hpp:265-267), floor 1024 slots
	const uint64_t capacity = next_pow2_u64(std::max<uint64_t>(2 * n, 1024));
	if (capacity > (1ull << 31)) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "hash table of %llu slots exceeds the 32-bit slot space",
		          (unsigned long long)capacity);
	}
	uint4 *slots = nullptr;
	uint32_t *slot_of_row = nullptr, *cursor = nullptr, *rowids = nullptr, *block_sums = nullptr, *scalars = nullptr;
	unsigned long long *n_valid = nullptr;
	uint64_t acct = 0;
	int rc = dev_alloc(ctx, (void **)&slots, capacity * sizeof(uint4), &acct);
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&slot_of_row, n * 4, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&cursor, capacity * 4, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&rowids, n * 4, &acct);
	}
	const uint64_t n_blocks = (capacity + 1023) / 1024;
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&block_sums, n_blocks * 4, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&scalars, 4 * 4, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&n_valid, 8, nullptr);
	}
	uint32_t h_scalars[4] = {0, 0, 0, 0};
	unsigned long long h_valid = 0;
	if (!rc) {
		hipError_t e = hipMemsetAsync(cursor, 0, capacity * 4, st);
		e = e == hipSuccess ? hipMemsetAsync(scalars, 0, 16, st) : e;
		e = e == hipSuccess ? hipMemsetAsync(n_valid, 0, 8, st) : e;
		if (e == hipSuccess) {
			polr_launch_s16_build(st, ht->keys_dev, ht->n_keys, ht->pack, n, slots, capacity, slot_of_row, cursor, rowids,
			                      block_sums, scalars, n_valid);
			e = hipMemcpyAsync(h_scalars, scalars, 16, hipMemcpyDeviceToHost, st);
			e = e == hipSuccess ? hipMemcpyAsync(&h_valid, n_valid, 8, hipMemcpyDeviceToHost, st) : e;
			e = e == hipSuccess ? hipStreamSynchronize(st) : e;
		}
		if (e == hipSuccess) {
			// h_scalars: [0] sentinel rows, [1] longest run, [2] rows with a regular key
			polr_launch_s16_scatter(st, n, slots, slot_of_row, cursor, rowids, h_scalars[2], &scalars[3]);
			e = hipStreamSynchronize(st);
		}
		if (e != hipSuccess) {
			ctx->err = std::string("hash table build failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	if (!rc) {
		ht->n_rows = h_valid;
		ht->has_null = h_valid < n;
		ht->capacity = capacity;
		ht->sentinel_start = h_scalars[2];
		ht->sentinel_count = h_scalars[0];
		ht->max_run = std::max<uint64_t>(h_scalars[1], h_scalars[0]);
		if (ht->max_run >= (1ull << 25)) {
			ctx->err = "a build key repeats more than 2^25 times: outside the expansion counter range";
			rc = POLR_E_UNSUPPORTED;
		}
	}
	if (!rc) {
		const bool unique32 = ht->max_run <= 1 && ht->n_keys == 1 && ht->keys[0].width == 4 && !ht->pack.packed;
		if (unique32) {
			uint2 *s8 = nullptr;
			rc = dev_alloc(ctx, (void **)&s8, capacity * sizeof(uint2), &ht->device_bytes);
			if (!rc) {
				polr_launch_s16_to_s8(st, slots, capacity, rowids, s8);
				hipError_t e = hipStreamSynchronize(st);
				if (e != hipSuccess) {
					ctx->err = std::string("s8 conversion failed: ") + hipGetErrorString(e);
					rc = POLR_E_HIP;
				}
			}
			if (!rc) {
				ht->table = s8;
				ht->kind = KIND_S8;
				hipFree(slots);
				hipFree(rowids);
				slots = nullptr;
				rowids = nullptr;
			}
		} else {
			ht->table = slots;
			ht->rowids = rowids;
			ht->kind = KIND_S16;
			ht->device_bytes += acct;
			slots = nullptr;
			rowids = nullptr;
		}
	}
	if (!rc) {
		rc = upload_devcols(ctx, ht->payload, &ht->payload_dev, st);
	}
	if (slots) {
		hipFree(slots);
	}
	if (rowids) {
		hipFree(rowids);
	}
	hipFree(slot_of_row);
	hipFree(cursor);
	hipFree(block_sums);
	hipFree(scalars);
	hipFree(n_valid);
	return rc;
}

static int alloc_perfect_cols(polr_ht *ht, uint64_t size) {
	polr_ctx *ctx = ht->ctx;
	ht->pcols.resize(ht->n_payload);
	for (uint32_t i = 0; i < ht->n_payload; i++) {
		OwnedCol &c = ht->pcols[i];
		c.width = ht->payload[i].width;
		c.flags = ht->payload[i].flags;
		c.owned = true;
		int rc = dev_alloc(ctx, (void **)&c.data, size * c.width, &ht->device_bytes);
		if (!rc) {
			rc = dev_alloc(ctx, (void **)&c.valid, size, &ht->device_bytes);
		}
		if (rc) {
			return rc;
		}
	}
	return POLR_OK;
}

int polr_ht_finalize_perfect(polr_ht *ht, int64_t min_value, int64_t max_value, void *stream) {
	POLR_ENTRY();
	if (!ht) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = ht->ctx;
	if (ht->kind != KIND_NONE) {
		POLR_FAIL(ctx, POLR_E_INVALID, "table already finalized");
	}
	if (ht->n_keys != 1) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "perfect hash join needs exactly one key (plan_comparison_join.cpp:63-133)");
	}
	if (ht->key_flags[0]) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "a key compared by value (CAST) or with NULL = NULL takes a hash table "
		                                   "(the reference plans perfect hash joins for plain equalities only)");
	}
	const bool is_signed = ht->key_signed != 0;
	const uint64_t range = is_signed ? (uint64_t)(max_value - min_value) : (uint64_t)max_value - (uint64_t)min_value;
	if ((is_signed && max_value < min_value) || (!is_signed && (uint64_t)max_value < (uint64_t)min_value) ||
	    range >= (1ull << 31)) {
		POLR_FAIL(ctx, POLR_E_INVALID, "perfect hash range [%lld, %lld] invalid", (long long)min_value,
		          (long long)max_value);
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	const uint64_t size = range + 1;
	const uint64_t words = (size + 31) / 32;
	uint32_t *bits = nullptr, *idx_row = nullptr, *flags = nullptr;
	unsigned long long *unique = nullptr;
	uint64_t acct = 0;
	int rc = dev_alloc(ctx, (void **)&bits, words * 4, &acct);
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&idx_row, size * 4, &acct);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&flags, 8, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&unique, 8, nullptr);
	}
	uint32_t h_flags[2] = {0, 0};
	unsigned long long h_unique = 0;
	if (!rc) {
		hipError_t e = hipMemsetAsync(bits, 0, words * 4, st);
		e = e == hipSuccess ? hipMemsetAsync(idx_row, 0xFF, size * 4, st) : e;
		e = e == hipSuccess ? hipMemsetAsync(flags, 0, 8, st) : e;
		e = e == hipSuccess ? hipMemsetAsync(unique, 0, 8, st) : e;
		if (e == hipSuccess) {
			polr_launch_pht_mark(st, ht->keys_dev, ht->n_rows_in, min_value, range, is_signed ? 1 : 0, bits, idx_row,
			                     flags, unique);
			e = hipMemcpyAsync(h_flags, flags, 8, hipMemcpyDeviceToHost, st);
			e = e == hipSuccess ? hipMemcpyAsync(&h_unique, unique, 8, hipMemcpyDeviceToHost, st) : e;
			e = e == hipSuccess ? hipStreamSynchronize(st) : e;
		}
		if (e != hipSuccess) {
			ctx->err = std::string("perfect table build failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	if (!rc && h_flags[0]) {
		ctx->err = "duplicate build key inside the perfect-hash range";
		rc = POLR_E_DUPLICATE;
	}
	if (!rc) {
		ht->bits = bits;
		ht->idx_row = idx_row;
		ht->device_bytes += acct;
		bits = nullptr;
		idx_row = nullptr;
		ht->min_value = min_value;
		ht->max_value = max_value;
		ht->range = range;
		ht->has_null = h_flags[1];
		ht->n_rows = h_unique;
		ht->capacity = size;
		ht->max_run = h_unique ? 1 : 0;
		ht->is_dense = (h_unique == size && !h_flags[1]) ? 1 : 0;
		rc = alloc_perfect_cols(ht, size);
	}
	if (!rc) {
		for (uint32_t i = 0; i < ht->n_payload; i++) {
			DevCol src;
			src.data = ht->payload[i].data;
			src.valid = ht->payload[i].valid;
			src.width = ht->payload[i].width;
			src.flags = ht->payload[i].flags;
			polr_launch_pht_gather(st, ht->bits, ht->idx_row, size, src, ht->pcols[i].data, ht->pcols[i].valid);
		}
		hipError_t e = hipStreamSynchronize(st);
		if (e != hipSuccess) {
			ctx->err = std::string("perfect column gather failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	if (!rc) {
		ht->table = ht->bits;
		ht->kind = KIND_PERFECT;
		rc = upload_devcols(ctx, ht->pcols, &ht->payload_dev, st);
	}
	if (bits) {
		hipFree(bits);
	}
	if (idx_row) {
		hipFree(idx_row);
	}
	hipFree(flags);
	hipFree(unique);
	return rc;
}

int polr_ht_finalize_auto(polr_ht *ht, int64_t min_value, int64_t max_value, void *stream, uint32_t *kind_out) {
	POLR_ENTRY();
	if (!ht) {
		return POLR_E_INVALID;
	}
	int rc = POLR_E_DUPLICATE;
	const bool is_signed = ht->key_signed != 0;
	const bool ordered = is_signed ? max_value >= min_value : (uint64_t)max_value >= (uint64_t)min_value;
	if (ht->kind == KIND_NONE && ht->n_keys == 1 && ordered && ht->n_rows_in > 0 && !ht->key_flags[0]) {
		const uint64_t range = is_signed ? (uint64_t)(max_value - min_value) : (uint64_t)max_value - (uint64_t)min_value;
		// dense keys of any range, and -- like the reference's planner, plan_comparison_join.cpp:118,125 -- any key
		// whose range is at most 1 M values (a 125 KB bit table, however few of its bits are set: a filtered dimension)
		if (range < (1ull << 31) && (range / POLR_DENSE_FACTOR <= ht->n_rows_in || range <= 1000000ull)) {
			rc = polr_ht_finalize_perfect(ht, min_value, max_value, stream);
			if (rc != POLR_OK && rc != POLR_E_DUPLICATE) {
				return rc;
			}
		}
	}
	if (rc == POLR_E_DUPLICATE) {
		rc = polr_ht_finalize_hash(ht, stream);
	}
	if (!rc && kind_out) {
		*kind_out = ht->kind;
	}
	return rc;
}

int polr_pht_upload(polr_ctx *ctx, uint32_t key_width, uint32_t key_flags, int64_t min_value, int64_t max_value,
                    const uint8_t *bitmap, const polr_col *payload, uint32_t n_payload, polr_ht **out) {
	POLR_ENTRY();
	if (!ctx || !out || !bitmap) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	const bool is_signed = (key_flags & POLR_COL_SIGNED) != 0;
	const uint64_t range = is_signed ? (uint64_t)(max_value - min_value) : (uint64_t)max_value - (uint64_t)min_value;
	if (range >= (1ull << 31) || key_width > 8 || !valid_width(key_width)) {
		POLR_FAIL(ctx, POLR_E_INVALID, "perfect table shape invalid");
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const uint64_t size = range + 1;
	polr_ht *ht = new polr_ht();
	ht->ctx = polr_ctx_retain(ctx);
	ht->n_keys = 1;
	ht->n_payload = n_payload;
	ht->key_signed = is_signed ? POLR_COL_SIGNED : 0;
	ht->keys.resize(1);
	ht->keys[0].width = key_width;
	ht->keys[0].flags = ht->key_signed;
	ht->keys[0].owned = false;
	ht->pcols.resize(n_payload);
	ht->payload.resize(n_payload);
	int rc = POLR_OK;
	for (uint32_t i = 0; i < n_payload && !rc; i++) {
		rc = ingest_col(ctx, &payload[i], size, &ht->pcols[i], &ht->device_bytes, ctx->stream);
		ht->payload[i].width = ht->pcols[i].width;
		ht->payload[i].owned = false;
	}
	uint8_t *bytes = nullptr;
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&bytes, size, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&ht->bits, ((size + 31) / 32) * 4, &ht->device_bytes);
	}
	if (!rc) {
		hipError_t e = hipMemcpyAsync(bytes, bitmap, size, hipMemcpyHostToDevice, ctx->stream);
		if (e == hipSuccess) {
			polr_launch_pack_bitmap(ctx->stream, bytes, size, ht->bits);
			e = hipStreamSynchronize(ctx->stream);
		}
		if (e != hipSuccess) {
			ctx->err = std::string("bitmap upload failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	if (bytes) {
		hipFree(bytes);
	}
	if (!rc) {
		uint64_t set = 0;
		for (uint64_t i = 0; i < size; i++) {
			set += bitmap[i] ? 1 : 0;
		}
		ht->n_rows = ht->n_rows_in = set;
		ht->min_value = min_value;
		ht->max_value = max_value;
		ht->range = range;
		ht->capacity = size;
		ht->max_run = set ? 1 : 0;
		ht->is_dense = set == size;
		ht->table = ht->bits;
		ht->kind = KIND_PERFECT;
		rc = upload_devcols(ctx, ht->pcols, &ht->payload_dev, ctx->stream);
	}
	if (rc) {
		polr_ht_destroy(ht);
		return rc;
	}
	*out = ht;
	return POLR_OK;
}

void polr_ht_destroy(polr_ht *ht) {
	POLR_ENTRY();
	if (!ht) {
		return;
	}
	hipSetDevice(ht->ctx->device);
	for (auto &c : ht->keys) {
		free_col(c);
	}
	for (auto &c : ht->payload) {
		free_col(c);
	}
	for (auto &c : ht->pcols) {
		free_col(c);
	}
	for (void *h : ht->heaps) {
		hipFree(h);
	}
	if (ht->kind == KIND_PERFECT) {
		if (ht->bits) {
			hipFree(ht->bits);
		}
	} else if (ht->table) {
		hipFree(ht->table);
	}
	if (ht->rowids) {
		hipFree(ht->rowids);
	}
	if (ht->idx_row) {
		hipFree(ht->idx_row);
	}
	if (ht->keys_dev) {
		hipFree(ht->keys_dev);
	}
	if (ht->payload_dev) {
		hipFree(ht->payload_dev);
	}
	polr_ctx *ctx_ = ht->ctx;
	delete ht;
	polr_ctx_release(ctx_);
}

int polr_ht_get_info(const polr_ht *ht, polr_ht_info *info) {
	POLR_ENTRY();
	if (!ht || !info) {
		return POLR_E_INVALID;
	}
	info->kind = ht->kind;
	info->n_keys = ht->n_keys;
	info->n_rows = ht->n_rows;
	info->capacity = ht->capacity;
	info->max_run = ht->max_run;
	info->device_bytes = ht->device_bytes;
	info->is_dense = ht->is_dense;
	info->has_null = ht->has_null;
	return POLR_OK;
}

// ---- export / alloc_like: the buffers a build-side broadcast has to move --------------------------
struct HtMeta {
	uint32_t magic, kind, n_keys, n_payload, key_signed, is_dense, has_null, sentinel_start, sentinel_count, pad;
	uint64_t n_rows_in, n_rows, capacity, max_run, range;
	int64_t min_value, max_value;
	uint32_t key_width[POLR_MAX_KEYS];
	uint32_t key_flags[POLR_MAX_KEYS];
	uint32_t key_sem[POLR_MAX_KEYS]; // POLR_KEY_* (polr_ht_set_key_flags)
	KeyPack pack;
	uint32_t payload_width[62];
	uint32_t payload_flags[62];
	uint8_t payload_has_valid[62];
};

static void ht_buffers(const polr_ht *ht, std::vector<void *> &ptrs, std::vector<uint64_t> &bytes) {
	const uint64_t rows = ht->kind == KIND_PERFECT ? ht->capacity : ht->n_rows_in;
	if (ht->kind == KIND_PERFECT) {
		ptrs.push_back(ht->bits);
		bytes.push_back(((ht->capacity + 31) / 32) * 4);
	} else if (ht->kind == KIND_S8) {
		ptrs.push_back(ht->table);
		bytes.push_back(ht->capacity * sizeof(uint2));
	} else {
		ptrs.push_back(ht->table);
		bytes.push_back(ht->capacity * sizeof(uint4));
		ptrs.push_back(ht->rowids);
		bytes.push_back(std::max<uint64_t>(ht->n_rows_in * 4, 16));
	}
	const std::vector<OwnedCol> &cols = ht->kind == KIND_PERFECT ? ht->pcols : ht->payload;
	for (auto &c : cols) {
		ptrs.push_back(c.data);
		bytes.push_back(std::max<uint64_t>(rows * c.width, 16));
		if (c.valid) {
			ptrs.push_back(c.valid);
			bytes.push_back(std::max<uint64_t>(rows, 16));
		}
	}
}

int polr_ht_export(const polr_ht *ht, void *meta, uint64_t *meta_bytes, void **dev_ptrs, uint64_t *dev_bytes,
                   uint32_t *n_buffers) {
	POLR_ENTRY();
	if (!ht || !meta_bytes || !n_buffers) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = ht->ctx;
	if (ht->kind == KIND_NONE) {
		POLR_FAIL(ctx, POLR_E_INVALID, "table not finalized");
	}
	if (ht->n_payload > 62) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "more than 62 payload columns");
	}
	std::vector<void *> ptrs;
	std::vector<uint64_t> bytes;
	ht_buffers(ht, ptrs, bytes);
	const uint32_t cap_buffers = *n_buffers;
	const uint64_t cap_meta = *meta_bytes;
	*n_buffers = (uint32_t)ptrs.size();
	*meta_bytes = sizeof(HtMeta);
	if (!meta || !dev_ptrs || !dev_bytes) {
		return POLR_OK; // size query
	}
	if (cap_buffers < ptrs.size() || cap_meta < sizeof(HtMeta)) {
		POLR_FAIL(ctx, POLR_E_INVALID, "export buffers too small");
	}
	HtMeta m;
	memset(&m, 0, sizeof(m));
	m.magic = 0x504F4C52u;
	m.kind = ht->kind;
	m.n_keys = ht->n_keys;
	m.n_payload = ht->n_payload;
	m.key_signed = ht->key_signed;
	m.is_dense = ht->is_dense;
	m.has_null = ht->has_null;
	m.sentinel_start = ht->sentinel_start;
	m.sentinel_count = ht->sentinel_count;
	m.n_rows_in = ht->n_rows_in;
	m.n_rows = ht->n_rows;
	m.capacity = ht->capacity;
	m.max_run = ht->max_run;
	m.range = ht->range;
	m.min_value = ht->min_value;
	m.max_value = ht->max_value;
	for (uint32_t i = 0; i < ht->n_keys; i++) {
		m.key_width[i] = ht->keys[i].width;
		m.key_flags[i] = ht->keys[i].flags;
		m.key_sem[i] = ht->key_flags[i];
	}
	m.pack = ht->pack;
	const std::vector<OwnedCol> &cols = ht->kind == KIND_PERFECT ? ht->pcols : ht->payload;
	for (uint32_t i = 0; i < ht->n_payload; i++) {
		m.payload_width[i] = cols[i].width;
		m.payload_flags[i] = cols[i].flags;
		m.payload_has_valid[i] = cols[i].valid ? 1 : 0;
	}
	memcpy(meta, &m, sizeof(m));
	for (size_t i = 0; i < ptrs.size(); i++) {
		dev_ptrs[i] = ptrs[i];
		dev_bytes[i] = bytes[i];
	}
	return POLR_OK;
}

int polr_ht_alloc_like(polr_ctx *ctx, const void *meta, uint64_t meta_bytes, polr_ht **out) {
	POLR_ENTRY();
	if (!ctx || !meta || !out || meta_bytes < sizeof(HtMeta)) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	HtMeta m;
	memcpy(&m, meta, sizeof(m));
	if (m.magic != 0x504F4C52u || m.n_payload > 62 || m.n_keys > POLR_MAX_KEYS) {
		POLR_FAIL(ctx, POLR_E_INVALID, "bad table metadata");
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	polr_ht *ht = new polr_ht();
	ht->ctx = polr_ctx_retain(ctx);
	ht->kind = m.kind;
	ht->n_keys = m.n_keys;
	ht->n_payload = m.n_payload;
	ht->key_signed = m.key_signed;
	ht->is_dense = m.is_dense;
	ht->has_null = m.has_null;
	ht->sentinel_start = m.sentinel_start;
	ht->sentinel_count = m.sentinel_count;
	ht->n_rows_in = m.n_rows_in;
	ht->n_rows = m.n_rows;
	ht->capacity = m.capacity;
	ht->max_run = m.max_run;
	ht->range = m.range;
	ht->min_value = m.min_value;
	ht->max_value = m.max_value;
	ht->keys.resize(m.n_keys);
	for (uint32_t i = 0; i < m.n_keys; i++) {
		ht->keys[i].width = m.key_width[i];
		ht->keys[i].flags = m.n_keys == 1 ? m.key_signed : m.key_flags[i];
		ht->keys[i].owned = false;
		ht->key_flags[i] = m.key_sem[i];
	}
	ht->pack = m.pack;
	int rc = POLR_OK;
	const uint64_t rows = m.kind == KIND_PERFECT ? m.capacity : m.n_rows_in;
	if (m.kind == KIND_PERFECT) {
		rc = dev_alloc(ctx, (void **)&ht->bits, ((m.capacity + 31) / 32) * 4, &ht->device_bytes);
		ht->table = ht->bits;
	} else if (m.kind == KIND_S8) {
		rc = dev_alloc(ctx, &ht->table, m.capacity * sizeof(uint2), &ht->device_bytes);
	} else if (m.kind == KIND_S16) {
		rc = dev_alloc(ctx, &ht->table, m.capacity * sizeof(uint4), &ht->device_bytes);
		if (!rc) {
			rc = dev_alloc(ctx, (void **)&ht->rowids, std::max<uint64_t>(m.n_rows_in * 4, 16), &ht->device_bytes);
		}
	} else {
		ctx->err = "bad table kind in metadata";
		rc = POLR_E_INVALID;
	}
	std::vector<OwnedCol> &cols = m.kind == KIND_PERFECT ? ht->pcols : ht->payload;
	cols.resize(m.n_payload);
	if (m.kind == KIND_PERFECT) {
		ht->payload.resize(m.n_payload);
	}
	for (uint32_t i = 0; i < m.n_payload && !rc; i++) {
		cols[i].width = m.payload_width[i];
		cols[i].flags = m.payload_flags[i];
		cols[i].owned = true;
		rc = dev_alloc(ctx, (void **)&cols[i].data, std::max<uint64_t>(rows * cols[i].width, 16), &ht->device_bytes);
		if (!rc && m.payload_has_valid[i]) {
			rc = dev_alloc(ctx, (void **)&cols[i].valid, std::max<uint64_t>(rows, 16), &ht->device_bytes);
		}
		if (m.kind == KIND_PERFECT) {
			ht->payload[i].width = cols[i].width;
			ht->payload[i].owned = false;
		}
	}
	if (!rc) {
		rc = upload_devcols(ctx, cols, &ht->payload_dev, ctx->stream);
	}
	if (rc) {
		polr_ht_destroy(ht);
		return rc;
	}
	*out = ht;
	return POLR_OK;
}

// ---------------------------------------------------------------------------------------------------
// Pipeline
// ---------------------------------------------------------------------------------------------------
static void fill_dev_join(DevJoin *dj, const polr_join_desc *jd, const polr_ht *ht) {
	memset(dj, 0, sizeof(*dj));
	dj->kind = ht->kind;
	dj->n_keys = ht->n_keys;
	for (uint32_t c = 0; c < ht->n_keys; c++) {
		dj->key_width[c] = ht->keys[c].width;
		dj->key_src_join[c] = jd->key_src_join[c];
		dj->key_src_col[c] = jd->key_src_col[c];
	}
	dj->key_signed = ht->key_signed ? 1 : 0;
	dj->n_payload = ht->n_payload;
	dj->mask = ht->capacity ? ht->capacity - 1 : 0;
	dj->min_value = ht->min_value;
	dj->range = ht->range;
	dj->table = ht->table;
	dj->rowids = ht->rowids;
	dj->sentinel_start = ht->sentinel_start;
	dj->sentinel_count = ht->sentinel_count;
	dj->payload = ht->payload_dev;
	dj->n_preds = jd->n_preds;
	for (uint32_t c = 0; c < jd->n_preds && c < POLR_NPREDS; c++) {
		dj->pred_op[c] = jd->pred_op[c];
		dj->pred_src_join[c] = jd->pred_src_join[c];
		dj->pred_src_col[c] = jd->pred_src_col[c];
		dj->pred_build_col[c] = jd->pred_build_col[c];
	}
}

// resolve every (join order, position) of a pipeline variant into a StageDesc (see polr_device.h)
// ext: extension records of the stages that need one (appended; StageDesc::ext holds the INDEX + 1 until the records
// have their device address, see polr_pipeline_create)
static void build_stage_descs(const polr_pipeline *p, const DevPipeline &dp, std::vector<StageDesc> &out,
                              std::vector<StageExt> &ext) {
	out.assign((size_t)dp.n_paths * POLR_KMAX, StageDesc());
	auto source = [&](int32_t sj, int32_t sc, int32_t &slot, const uint8_t *&data, const uint8_t *&valid) {
		if (sj < 0) {
			slot = 0;
			data = p->probe_cols[sc].data;
			valid = p->probe_cols[sc].valid;
		} else {
			const polr_ht *src = p->hts[sj];
			const OwnedCol &col = src->kind == KIND_PERFECT ? src->pcols[sc] : src->payload[sc];
			slot = dp.slot_of_join[sj];
			data = col.data;
			valid = col.valid;
		}
	};
	// width and signedness of the column a key is READ from (a key compared by value may differ from the build column)
	auto source_type = [&](int32_t sj, int32_t sc, uint32_t &width, uint32_t &sx) {
		const OwnedCol &col = sj < 0 ? p->probe_cols[sc]
		                             : (p->hts[sj]->kind == KIND_PERFECT ? p->hts[sj]->pcols[sc] : p->hts[sj]->payload[sc]);
		width = col.width;
		sx = (col.flags & 1u) ? 1u : 0u;
	};
	for (uint32_t q = 0; q < dp.n_paths; q++) {
		for (uint32_t pos = 0; pos < dp.k; pos++) {
			const uint32_t j = dp.paths[q].order[pos];
			const DevJoin &dj = dp.joins[j];
			const polr_ht *ht = p->hts[j];
			StageDesc &d = out[(size_t)q * POLR_KMAX + pos];
			memset(&d, 0, sizeof(d));
			d.kind = dj.kind;
			d.n_keys = dj.n_keys;
			d.key_signed = dj.key_signed;
			d.out_slot = dp.slot_of_join[j];
			for (uint32_t c = 0; c < dj.n_keys && c < 2; c++) {
				d.key_width[c] = dj.key_width[c];
				source(dj.key_src_join[c], dj.key_src_col[c], d.key_slot[c], d.key_data[c], d.key_valid[c]);
			}
			d.packed = ht->pack.packed;
			d.n_preds = dj.n_preds;
			if (d.packed || d.n_preds) {
				StageExt x;
				memset(&x, 0, sizeof(x));
				for (uint32_t c = 0; c < dj.n_keys; c++) {
					source_type(dj.key_src_join[c], dj.key_src_col[c], x.key_width[c], x.key_sx[c]);
					source(dj.key_src_join[c], dj.key_src_col[c], x.key_slot[c], x.key_data[c], x.key_valid[c]);
				}
				x.pack = ht->pack;
				x.n_preds = dj.n_preds;
				for (uint32_t c = 0; c < dj.n_preds; c++) {
					const OwnedCol &bcol =
					    ht->kind == KIND_PERFECT ? ht->pcols[dj.pred_build_col[c]] : ht->payload[dj.pred_build_col[c]];
					x.pred_op[c] = dj.pred_op[c];
					x.pred_width[c] = bcol.width;
					x.pred_sx[c] = (bcol.flags & 1u) ? 1u : 0u;
					x.pred_bdata[c] = bcol.data;
					x.pred_bvalid[c] = bcol.valid;
					source(dj.pred_src_join[c], dj.pred_src_col[c], x.pred_slot[c], x.pred_data[c], x.pred_valid[c]);
				}
				ext.push_back(x);
				d.ext = (const StageExt *)(uintptr_t)ext.size(); // index + 1, patched to the device address later
			}
			d.table = ht->table;
			d.rowids = ht->rowids;
			d.mask = dj.mask;
			d.min_value = dj.min_value;
			d.range = dj.range;
			d.sentinel_start = dj.sentinel_start;
			d.sentinel_count = dj.sentinel_count;
			// 1: at most one match per tuple; 2: keys may repeat (wide steps fall back to a narrow one where they do)
			d.unique = (ht->kind == KIND_PERFECT || ht->kind == KIND_S8 || ht->max_run <= 1) ? 1u : 2u;
		}
	}
}

// Flat pipelines (polr_flat_device.h): every join keyed by ONE 4-byte probe column, at most one build row per key
// (perfect bit table or unique-key hash table).  Decides the workgroup shape of the flat pool kernel and which bit
// tables stay in LDS for the whole run (smallest first, while they fit beside the per-wave queues).
static void plan_flat(polr_pipeline *p, std::vector<StageDesc> &sd_count) {
	DevPipeline &c = p->host_count;
	c.flat = 0;
	c.n_lds_tables = 0;
	c.lds_table_dwords = 0;
	if (c.W != 1 || c.k > 6) {
		return; // some join reads its key through a build column / more joins than the sweep holds in registers
	}
	for (uint32_t j = 0; j < c.k; j++) {
		const polr_ht *ht = p->hts[j];
		const DevJoin &dj = c.joins[j];
		if (dj.n_keys != 1 || dj.key_src_join[0] >= 0 || dj.key_width[0] != 4 || dj.n_preds != 0) {
			return;
		}
		if (ht->kind == KIND_PERFECT) {
			// the flat lookup works in 32-bit modular arithmetic: [min, max] must lie inside the key type's domain
			const int64_t lo = dj.key_signed ? -2147483648ll : 0ll;
			const int64_t hi = dj.key_signed ? 2147483647ll : 4294967295ll;
			if (ht->min_value < lo || ht->max_value > hi || ht->range > 0xFFFFFFFFull) {
				return;
			}
		} else if (ht->kind != KIND_S8 || ht->capacity > (1ull << 31)) {
			return;
		}
	}
	const size_t per_wave = polr_pool_flat_wave_bytes(c.k);
	uint32_t wpb = 4;
	for (uint32_t w : {16u, 8u}) {
		if (per_wave * w <= 120u * 1024) {
			wpb = w;
			break;
		}
	}
	p->flat_wpb = wpb;
	// one workgroup of 16 waves per CU leaves the rest of the 160 KB to the tables; smaller workgroups share a CU
	const size_t budget = wpb == 16 ? std::min<size_t>(64u * 1024, 156u * 1024 - per_wave * wpb) : 16u * 1024;
	std::vector<uint32_t> order;
	for (uint32_t j = 0; j < c.k; j++) {
		if (p->hts[j]->kind == KIND_PERFECT) {
			order.push_back(j);
		}
	}
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return p->hts[a]->range < p->hts[b]->range; });
	uint32_t off_of_join[POLR_KMAX];
	for (uint32_t j = 0; j < POLR_KMAX; j++) {
		off_of_join[j] = 0;
	}
	uint32_t used = 0;
	for (uint32_t j : order) {
		const polr_ht *ht = p->hts[j];
		const uint32_t words = (uint32_t)((ht->range + 1 + 31) / 32);
		// the same build side joined twice shares one LDS copy
		bool shared = false;
		for (uint32_t t = 0; t < c.n_lds_tables; t++) {
			if (c.lds_table_src[t] == (const uint32_t *)ht->table) {
				off_of_join[j] = c.lds_table_off[t] + 1;
				shared = true;
			}
		}
		if (shared) {
			continue;
		}
		const uint32_t padded = (words + 3u) & ~3u;
		if (((size_t)used + padded) * 4 > budget || c.n_lds_tables >= POLR_KMAX) {
			break;
		}
		c.lds_table_src[c.n_lds_tables] = (const uint32_t *)ht->table;
		c.lds_table_off[c.n_lds_tables] = used;
		c.lds_table_len[c.n_lds_tables] = words;
		c.n_lds_tables++;
		off_of_join[j] = used + 1;
		used += padded;
	}
	c.lds_table_dwords = used;
	// (an emitting run needs every join's build id: a perfect table's is the key's offset, a hash table's would take a
	// second probe -- such banks emit through the generic pipeline)
	p->flat_emit = order.size() == c.k;
	for (uint32_t q = 0; q < c.n_paths; q++) {
		for (uint32_t pos = 0; pos < c.k; pos++) {
			sd_count[(size_t)q * POLR_KMAX + pos].lds_off1 = off_of_join[c.paths[q].order[pos]];
		}
	}
	c.flat = 1;
}

int polr_pipeline_create(polr_ctx *ctx, const polr_col *probe_cols, uint32_t n_probe_cols, uint64_t n_probe_rows,
                         const polr_join_desc *joins, uint32_t k, const int32_t *paths, uint32_t n_paths,
                         polr_pipeline **out) {
	POLR_ENTRY();
	if (!ctx || !out || !joins || !paths || (!probe_cols && n_probe_cols)) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	if (k < 1 || k > POLR_MAX_JOINS) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "%u multiplexed joins not supported (1..%d)", k, POLR_MAX_JOINS);
	}
	if (n_paths < 1 || n_paths > POLR_MAX_PATHS) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "%u join orders not supported (1..%d)", n_paths, POLR_MAX_PATHS);
	}
	if (n_probe_rows >= 0xFFFFFFF0ull) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "probe side of %llu rows exceeds the 32-bit row-id space per shard",
		          (unsigned long long)n_probe_rows);
	}
	// the descriptors first: the dependency walk below reads n_keys / n_preds entries of every join and shifts by the
	// join index a key or a condition names
	for (uint32_t j = 0; j < k; j++) {
		const polr_ht *ht = joins[j].ht;
		if (!ht || ht->kind == KIND_NONE) {
			POLR_FAIL(ctx, POLR_E_INVALID, "join %u: build side not finalized", j);
		}
		if (joins[j].n_keys != ht->n_keys || joins[j].n_keys < 1 || joins[j].n_keys > POLR_MAX_KEYS) {
			POLR_FAIL(ctx, POLR_E_INVALID, "join %u: %u probe keys for a %u-key table", j, joins[j].n_keys, ht->n_keys);
		}
		if (joins[j].n_preds > POLR_MAX_PREDS) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "join %u: %u non-equality conditions (at most %d)", j, joins[j].n_preds,
			          POLR_MAX_PREDS);
		}
		for (uint32_t c = 0; c < joins[j].n_keys; c++) {
			const int32_t sj = joins[j].key_src_join[c];
			if (sj >= 0 && ((uint32_t)sj >= k || (uint32_t)sj == j)) {
				POLR_FAIL(ctx, POLR_E_INVALID, "join %u key %u: reads a build column of join %d", j, c, sj);
			}
		}
		for (uint32_t c = 0; c < joins[j].n_preds; c++) {
			const int32_t sj = joins[j].pred_src_join[c];
			if (sj >= 0 && ((uint32_t)sj >= k || (uint32_t)sj == j)) {
				POLR_FAIL(ctx, POLR_E_INVALID, "join %u condition %u: reads a build column of join %d", j, c, sj);
			}
		}
	}
	// every path must be a permutation of 0..k-1 that respects the key dependencies
	// (POLARConfig join_prerequisites, polar_config.cpp:72-95)
	for (uint32_t p = 0; p < n_paths; p++) {
		uint32_t seen = 0;
		for (uint32_t j = 0; j < k; j++) {
			const int32_t x = paths[p * k + j];
			if (x < 0 || (uint32_t)x >= k || (seen >> x) & 1) {
				POLR_FAIL(ctx, POLR_E_INVALID, "path %u is not a permutation of the %u joins", p, k);
			}
			for (uint32_t c = 0; c < joins[x].n_keys; c++) {
				const int32_t sj = joins[x].key_src_join[c];
				if (sj >= 0 && !((seen >> sj) & 1)) {
					POLR_FAIL(ctx, POLR_E_INVALID, "path %u probes join %d before join %d that provides its key", p, x,
					          sj);
				}
			}
			for (uint32_t c = 0; c < joins[x].n_preds; c++) {
				const int32_t sj = joins[x].pred_src_join[c];
				if (sj >= 0 && !((seen >> sj) & 1)) {
					POLR_FAIL(ctx, POLR_E_INVALID, "path %u probes join %d before join %d that a condition of it reads", p, x,
					          sj);
				}
			}
			seen |= 1u << x;
		}
	}
	for (uint32_t j = 0; j < k; j++) {
		const polr_ht *ht = joins[j].ht;
		if (!ht || ht->kind == KIND_NONE) {
			POLR_FAIL(ctx, POLR_E_INVALID, "join %u: build side not finalized", j);
		}
		if (ht->ctx->device != ctx->device) {
			POLR_FAIL(ctx, POLR_E_INVALID, "join %u: build side lives on another device", j);
		}
		if (joins[j].n_keys != ht->n_keys) {
			POLR_FAIL(ctx, POLR_E_INVALID, "join %u: %u probe keys for a %u-key table", j, joins[j].n_keys, ht->n_keys);
		}
		for (uint32_t c = 0; c < ht->n_keys; c++) {
			const int32_t sj = joins[j].key_src_join[c];
			const int32_t sc = joins[j].key_src_col[c];
			uint32_t width;
			if (sj < 0) {
				if (sc < 0 || (uint32_t)sc >= n_probe_cols) {
					POLR_FAIL(ctx, POLR_E_INVALID, "join %u key %u: probe column %d out of range", j, c, sc);
				}
				width = probe_cols[sc].width;
			} else {
				if ((uint32_t)sj >= k || (uint32_t)sj == j || sc < 0 || (uint32_t)sc >= joins[sj].ht->n_payload) {
					POLR_FAIL(ctx, POLR_E_INVALID, "join %u key %u: build column (%d,%d) out of range", j, c, sj, sc);
				}
				width = joins[sj].ht->payload[sc].width;
			}
			// JoinHashTable asserts left/right key types equal (join_hashtable.cpp:24): the reference's left side is then
			// CAST(column) (polar_config.cpp:75-82).  An integer cast is a comparison by VALUE, which a table whose key
			// column carries POLR_KEY_BY_VALUE does on the device -- no materialised copy of the probe column
			if (width != ht->keys[c].width && !(ht->key_flags[c] & POLR_KEY_BY_VALUE)) {
				POLR_FAIL(ctx, POLR_E_INVALID,
				          "join %u key %u: probe key is %u bytes, build key %u bytes (a CAST'ed key: polr_ht_set_key_flags(..., "
				          "POLR_KEY_BY_VALUE) before the table is finalized)",
				          j, c, width, ht->keys[c].width);
			}
			if (width != 1 && width != 2 && width != 4 && width != 8) {
				POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "join %u key %u: probe key of %u bytes", j, c, width);
			}
		}
		if (joins[j].n_preds > POLR_MAX_PREDS) {
			POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "join %u: %u non-equality conditions (at most %d)", j, joins[j].n_preds,
			          POLR_MAX_PREDS);
		}
		for (uint32_t c = 0; c < joins[j].n_preds; c++) {
			const int32_t sj = joins[j].pred_src_join[c];
			const int32_t sc = joins[j].pred_src_col[c];
			const uint32_t bc = joins[j].pred_build_col[c];
			const uint32_t op = joins[j].pred_op[c];
			if (op > POLR_CMP_GE && op != POLR_CMP_STR_EQ) {
				POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "join %u condition %u: comparison %u (EQ, NE, LT, GT, LE, GE, STR_EQ)", j, c, op);
			}
			if (bc >= ht->n_payload) {
				POLR_FAIL(ctx, POLR_E_INVALID, "join %u condition %u: build column %u out of range", j, c, bc);
			}
			uint32_t width;
			if (sj < 0) {
				if (sc < 0 || (uint32_t)sc >= n_probe_cols) {
					POLR_FAIL(ctx, POLR_E_INVALID, "join %u condition %u: probe column %d out of range", j, c, sc);
				}
				width = probe_cols[sc].width;
			} else {
				if ((uint32_t)sj >= k || (uint32_t)sj == j || sc < 0 || (uint32_t)sc >= joins[sj].ht->n_payload) {
					POLR_FAIL(ctx, POLR_E_INVALID, "join %u condition %u: build column (%d,%d) out of range", j, c, sj, sc);
				}
				width = joins[sj].ht->payload[sc].width;
			}
			const OwnedCol &bcol = ht->kind == KIND_PERFECT ? ht->pcols[bc] : ht->payload[bc];
			if (op == POLR_CMP_STR_EQ) {
				if (width != 16 || bcol.width != 16) {
					POLR_FAIL(ctx, POLR_E_INVALID, "join %u condition %u: STR_EQ compares two columns of 16-byte string cells "
					                               "(left %u bytes, right %u bytes)", j, c, width, bcol.width);
				}
			} else if (width != bcol.width || (width != 1 && width != 2 && width != 4 && width != 8)) {
				POLR_FAIL(ctx, POLR_E_INVALID, "join %u condition %u: left side is %u bytes, right side %u bytes", j, c, width,
				          bcol.width);
			}
		}
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	polr_pipeline *p = new polr_pipeline();
	p->ctx = polr_ctx_retain(ctx);
	p->k = k;
	p->n_paths = n_paths;
	p->n_probe_cols = n_probe_cols;
	p->n_probe_rows = n_probe_rows;
	p->n_tuples = n_probe_rows;
	p->probe_cols.resize(n_probe_cols);
	int rc = POLR_OK;
	uint64_t acct = 0;
	for (uint32_t i = 0; i < n_probe_cols && !rc; i++) {
		rc = ingest_col(ctx, &probe_cols[i], n_probe_rows, &p->probe_cols[i], &acct, ctx->stream);
	}
	if (!rc) {
		rc = upload_devcols(ctx, p->probe_cols, &p->probe_cols_dev, ctx->stream);
	}
	if (!rc) {
		DevPipeline &m = p->host_mat;
		memset(&m, 0, sizeof(m));
		m.k = k;
		m.n_paths = n_paths;
		m.n_probe_cols = n_probe_cols;
		m.probe_cols = p->probe_cols_dev;
		m.sel = nullptr;
		m.n_tuples = n_probe_rows;
		for (uint32_t j = 0; j < k; j++) {
			p->hts.push_back(joins[j].ht);
			fill_dev_join(&m.joins[j], &joins[j], joins[j].ht);
		}
		for (uint32_t q = 0; q < n_paths; q++) {
			for (uint32_t j = 0; j < k; j++) {
				m.paths[q].order[j] = (uint32_t)paths[q * k + j];
			}
		}
		p->host_count = m;
		// materialising variant: slot 1+j = join j (the adaptive union's column order)
		m.materialize = 1;
		m.W = 1 + k;
		for (uint32_t j = 0; j < POLR_KMAX; j++) {
			m.slot_of_join[j] = j < k ? (int32_t)(1 + j) : -1;
		}
		// counting variant: carry only the build ids some later join reads its key through
		DevPipeline &c = p->host_count;
		c.materialize = 0;
		uint32_t w = 1;
		for (uint32_t j = 0; j < POLR_KMAX; j++) {
			c.slot_of_join[j] = -1;
		}
		for (uint32_t j = 0; j < k; j++) {
			for (uint32_t cc = 0; cc < joins[j].n_keys; cc++) {
				const int32_t sj = joins[j].key_src_join[cc];
				if (sj >= 0 && c.slot_of_join[sj] < 0) {
					c.slot_of_join[sj] = (int32_t)w++;
				}
			}
			for (uint32_t cc = 0; cc < joins[j].n_preds; cc++) {
				const int32_t sj = joins[j].pred_src_join[cc];
				if (sj >= 0 && c.slot_of_join[sj] < 0) {
					c.slot_of_join[sj] = (int32_t)w++;
				}
			}
		}
		c.W = w;
		// multiplicities (polr_gen_device.h): worth a tuple slot when some join's matches can be folded into them -- its
		// build key may repeat, nobody reads its build rows downstream, it has no non-equality condition
		c.mult = 0;
		for (uint32_t j = 0; j < k; j++) {
			const polr_ht *ht = joins[j].ht;
			const bool repeats = !(ht->kind == KIND_PERFECT || ht->kind == KIND_S8 || ht->max_run <= 1);
			if (repeats && c.slot_of_join[j] < 0 && joins[j].n_preds == 0) {
				c.mult = 1;
			}
		}
		p->host_mat.mult = 0;
		std::vector<StageDesc> sd_mat, sd_count;
		std::vector<StageExt> sd_ext;
		build_stage_descs(p, p->host_mat, sd_mat, sd_ext);
		build_stage_descs(p, p->host_count, sd_count, sd_ext);
		plan_flat(p, sd_count);
		p->host_mat.ext = p->host_count.ext = sd_ext.empty() ? 0u : 1u;
		hipError_t e = hipSuccess;
		if (!sd_ext.empty()) {
			e = hipMalloc((void **)&p->stage_ext, sd_ext.size() * sizeof(StageExt));
			e = e == hipSuccess ? hipMemcpy(p->stage_ext, sd_ext.data(), sd_ext.size() * sizeof(StageExt), hipMemcpyHostToDevice)
			                    : e;
			for (auto *sd : {&sd_mat, &sd_count}) {
				for (auto &d : *sd) {
					if (d.ext) {
						d.ext = p->stage_ext + ((uintptr_t)d.ext - 1);
					}
				}
			}
		}
		e = e == hipSuccess ? hipMalloc((void **)&p->stages_mat, sd_mat.size() * sizeof(StageDesc)) : e;
		e = e == hipSuccess ? hipMalloc((void **)&p->stages_count, sd_count.size() * sizeof(StageDesc)) : e;
		e = e == hipSuccess ? hipMemcpy(p->stages_mat, sd_mat.data(), sd_mat.size() * sizeof(StageDesc),
		                                hipMemcpyHostToDevice)
		                    : e;
		e = e == hipSuccess ? hipMemcpy(p->stages_count, sd_count.data(), sd_count.size() * sizeof(StageDesc),
		                                hipMemcpyHostToDevice)
		                    : e;
		p->host_mat.stages = p->stages_mat;
		p->host_count.stages = p->stages_count;
		e = e == hipSuccess ? hipMalloc((void **)&p->dev_mat, sizeof(DevPipeline)) : e;
		e = e == hipSuccess ? hipMalloc((void **)&p->dev_count, sizeof(DevPipeline)) : e;
		e = e == hipSuccess ? hipMemcpy(p->dev_mat, &p->host_mat, sizeof(DevPipeline), hipMemcpyHostToDevice) : e;
		e = e == hipSuccess ? hipMemcpy(p->dev_count, &p->host_count, sizeof(DevPipeline), hipMemcpyHostToDevice) : e;
		if (e != hipSuccess) {
			ctx->err = std::string("pipeline upload failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	if (rc) {
		polr_pipeline_destroy(p);
		return rc;
	}
	*out = p;
	return POLR_OK;
}

int polr_pipeline_set_selection(polr_pipeline *p, const uint32_t *sel, uint64_t n_sel, uint32_t flags) {
	POLR_ENTRY();
	if (!p) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (p->sel_dev && p->sel_owned) {
		hipFree(p->sel_dev);
	}
	p->sel_dev = nullptr;
	p->sel_owned = false;
	p->scan_n_chunks = 0; // a caller-given selection replaces a scan result (its buffers stay for the next scan)
	p->scan_valid = false;
	if (!sel) {
		p->n_tuples = p->n_probe_rows;
	} else {
		if (flags & POLR_COL_DEVICE) {
			p->sel_dev = (uint32_t *)sel;
		} else {
			int rc = dev_alloc(ctx, (void **)&p->sel_dev, n_sel * 4, nullptr);
			if (rc) {
				return rc;
			}
			p->sel_owned = true;
			if (n_sel) {
				HIPCHK(ctx, hipMemcpy(p->sel_dev, sel, n_sel * 4, hipMemcpyHostToDevice));
			}
		}
		p->n_tuples = n_sel;
	}
	p->host_mat.sel = p->sel_dev;
	p->host_mat.n_tuples = p->n_tuples;
	p->host_count.sel = p->sel_dev;
	p->host_count.n_tuples = p->n_tuples;
	HIPCHK(ctx, hipMemcpy(p->dev_mat, &p->host_mat, sizeof(DevPipeline), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(p->dev_count, &p->host_count, sizeof(DevPipeline), hipMemcpyHostToDevice));
	return POLR_OK;
}

int polr_pipeline_update_probe(polr_pipeline *p, uint32_t col, const void *data, const uint8_t *valid,
                               uint64_t n_rows) {
	POLR_ENTRY();
	if (!p || (!data && n_rows)) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	if (col >= p->n_probe_cols || n_rows > p->n_probe_rows) {
		POLR_FAIL(ctx, POLR_E_INVALID, "update of probe column %u with %llu rows does not fit (%u columns, %llu rows)",
		          col, (unsigned long long)n_rows, p->n_probe_cols, (unsigned long long)p->n_probe_rows);
	}
	OwnedCol &c = p->probe_cols[col];
	if (!c.owned) {
		POLR_FAIL(ctx, POLR_E_INVALID, "probe column %u is caller-owned device memory", col);
	}
	if (valid && !c.valid) {
		POLR_FAIL(ctx, POLR_E_INVALID, "probe column %u was created without a validity array", col);
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (n_rows) {
		HIPCHK(ctx, hipMemcpyAsync(c.data, data, n_rows * c.width, hipMemcpyHostToDevice, ctx->stream));
		if (c.valid) {
			if (valid) {
				HIPCHK(ctx, hipMemcpyAsync(c.valid, valid, n_rows, hipMemcpyHostToDevice, ctx->stream));
			} else {
				HIPCHK(ctx, hipMemsetAsync(c.valid, 1, n_rows, ctx->stream));
			}
		}
	}
	if (!p->sel_dev) {
		p->n_tuples = n_rows;
		p->host_mat.n_tuples = n_rows;
		p->host_count.n_tuples = n_rows;
	}
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return POLR_OK;
}

void polr_pipeline_destroy(polr_pipeline *p) {
	POLR_ENTRY();
	if (!p) {
		return;
	}
	hipSetDevice(p->ctx->device);
	for (auto &c : p->probe_cols) {
		free_col(c);
	}
	for (void *h : p->heaps) {
		hipFree(h);
	}
	if (p->probe_cols_dev) {
		hipFree(p->probe_cols_dev);
	}
	if (p->sel_dev && p->sel_owned) {
		hipFree(p->sel_dev);
	}
	if (p->scan_offsets_dev) {
		hipFree(p->scan_offsets_dev);
	}
	if (p->scan_sel) {
		hipFree(p->scan_sel);
	}
	if (p->scan_packed) {
		hipFree(p->scan_packed);
		hipFree(p->scan_sums);
		hipFree(p->scan_totals);
	}
	if (p->dev_mat) {
		hipFree(p->dev_mat);
	}
	if (p->dev_count) {
		hipFree(p->dev_count);
	}
	if (p->stage_ext) {
		hipFree(p->stage_ext);
	}
	if (p->stages_mat) {
		hipFree(p->stages_mat);
	}
	if (p->stages_count) {
		hipFree(p->stages_count);
	}
	if (p->rounds_dev) {
		hipFree(p->rounds_dev);
	}
	if (p->prefix_dev) {
		hipFree(p->prefix_dev);
	}
	if (p->unit_sizes_dev) {
		hipFree(p->unit_sizes_dev);
	}
	if (p->counts_dev) {
		hipFree(p->counts_dev);
	}
	if (p->shards_dev) {
		hipFree(p->shards_dev);
	}
	polr_ctx *ctx_ = p->ctx;
	delete p;
	polr_ctx_release(ctx_);
}

// ---------------------------------------------------------------------------------------------------
// Output chunks
// ---------------------------------------------------------------------------------------------------
int polr_out_create(polr_pipeline *p, uint32_t chunk_capacity, uint64_t max_chunks, polr_out **out) {
	POLR_ENTRY();
	if (!p || !out) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	*out = nullptr;
	if (chunk_capacity < 64 || chunk_capacity > 65536 || max_chunks < 1 || max_chunks > 0x7FFFFFFFull) {
		POLR_FAIL(ctx, POLR_E_INVALID, "output chunk capacity %u / max_chunks %llu out of range", chunk_capacity,
		          (unsigned long long)max_chunks);
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	polr_out *o = new polr_out();
	o->pipe = p;
	o->ctx = polr_ctx_retain(p->ctx);
	memset(&o->dev, 0, sizeof(o->dev));
	o->dev.chunk_capacity = chunk_capacity;
	o->dev.max_chunks = (uint32_t)max_chunks;
	o->dev.W_out = 1 + p->k;
	o->dev.slot_stride = max_chunks * chunk_capacity;
	int rc = dev_alloc(ctx, (void **)&o->dev.ids, o->dev.slot_stride * o->dev.W_out * 4, nullptr);
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&o->dev.chunk_count, max_chunks * 4, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&o->dev.cursor, 8, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&o->chunk_base, max_chunks * 8, nullptr);
	}
	if (!rc) {
		rc = dev_alloc(ctx, (void **)&o->total_dev, 8, nullptr);
	}
	if (!rc) {
		hipError_t e = hipMemset(o->dev.chunk_count, 0, max_chunks * 4);
		e = e == hipSuccess ? hipMemset(o->dev.cursor, 0, 8) : e;
		if (e != hipSuccess) {
			ctx->err = std::string("output reset failed: ") + hipGetErrorString(e);
			rc = POLR_E_HIP;
		}
	}
	if (rc) {
		polr_out_destroy(o);
		return rc;
	}
	*out = o;
	return POLR_OK;
}

int polr_out_reset(polr_out *o, void *stream) {
	POLR_ENTRY();
	if (!o) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = o->pipe->ctx;
	hipStream_t st = polr_stream(ctx, stream);
	HIPCHK(ctx, hipMemsetAsync(o->dev.chunk_count, 0, (uint64_t)o->dev.max_chunks * 4, st));
	HIPCHK(ctx, hipMemsetAsync(o->dev.cursor, 0, 8, st));
	if (o->fused_cells) { // (a fused GROUP BY sink: its cells start from zero, too)
		HIPCHK(ctx, hipMemsetAsync(o->fused_cells, 0,
		                           (size_t)o->fused_tables * o->fused_groups * (1u + 2u * o->fused_aggs) * 8u, st));
		HIPCHK(ctx, hipMemsetAsync(o->fused_dropped, 0, 8, st));
	}
	o->stats_valid = false;
	return POLR_OK;
}

int polr_out_stats(polr_out *o, void *stream, uint64_t *n_rows, uint64_t *n_chunks, uint32_t *overflowed) {
	POLR_ENTRY();
	if (!o) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = o->pipe->ctx;
	hipStream_t st = polr_stream(ctx, stream);
	uint32_t cur[2] = {0, 0};
	HIPCHK(ctx, hipMemcpyAsync(cur, o->dev.cursor, 8, hipMemcpyDeviceToHost, st));
	HIPCHK(ctx, hipStreamSynchronize(st));
	const uint32_t nc = std::min<uint32_t>(cur[0], o->dev.max_chunks);
	uint64_t total = 0;
	if (nc) {
		polr_launch_chunk_prefix(st, o->dev.chunk_count, nc, o->chunk_base, o->total_dev);
		HIPCHK(ctx, hipMemcpyAsync(&total, o->total_dev, 8, hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
	}
	o->n_rows = total;
	o->n_chunks = nc;
	o->stats_valid = true;
	if (n_rows) {
		*n_rows = total;
	}
	if (n_chunks) {
		*n_chunks = nc;
	}
	if (overflowed) {
		*overflowed = cur[1];
	}
	return POLR_OK;
}

int polr_out_fetch_ids(polr_out *o, void *stream, uint32_t *dst, uint64_t dst_rows) {
	POLR_ENTRY();
	if (!o || (!dst && dst_rows)) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = o->pipe->ctx;
	hipStream_t st = polr_stream(ctx, stream);
	if (!o->stats_valid) {
		int rc = polr_out_stats(o, stream, nullptr, nullptr, nullptr);
		if (rc) {
			return rc;
		}
	}
	if (dst_rows < o->n_rows) {
		POLR_FAIL(ctx, POLR_E_INVALID, "destination holds %llu rows, output has %llu", (unsigned long long)dst_rows,
		          (unsigned long long)o->n_rows);
	}
	if (o->n_rows == 0) {
		return POLR_OK;
	}
	uint32_t *tmp = nullptr;
	const uint64_t bytes = o->n_rows * o->dev.W_out * 4;
	HIPCHK(ctx, hipMalloc((void **)&tmp, bytes));
	polr_launch_compact_ids(st, o->dev, o->chunk_base, o->n_chunks, tmp);
	hipError_t e = hipMemcpyAsync(dst, tmp, bytes, hipMemcpyDeviceToHost, st);
	e = e == hipSuccess ? hipStreamSynchronize(st) : e;
	hipFree(tmp);
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "fetch ids failed: %s", hipGetErrorString(e));
	}
	return POLR_OK;
}

int polr_out_materialize(polr_out *o, void *stream, int32_t src_join, uint32_t src_col, void *dst_data,
                         uint8_t *dst_valid, uint64_t dst_rows, uint32_t dst_flags) {
	POLR_ENTRY();
	if (!o || (!dst_data && dst_rows)) {
		return POLR_E_INVALID;
	}
	polr_pipeline *p = o->pipe;
	polr_ctx *ctx = p->ctx;
	hipStream_t st = polr_stream(ctx, stream);
	if (!o->stats_valid) {
		int rc = polr_out_stats(o, stream, nullptr, nullptr, nullptr);
		if (rc) {
			return rc;
		}
	}
	if (dst_rows < o->n_rows) {
		POLR_FAIL(ctx, POLR_E_INVALID, "destination holds %llu rows, output has %llu", (unsigned long long)dst_rows,
		          (unsigned long long)o->n_rows);
	}
	DevCol src;
	uint32_t slot;
	if (src_join < 0) {
		if (src_col >= p->n_probe_cols) {
			POLR_FAIL(ctx, POLR_E_INVALID, "probe column %u out of range", src_col);
		}
		const OwnedCol &c = p->probe_cols[src_col];
		src.data = c.data;
		src.valid = c.valid;
		src.width = c.width;
		src.flags = c.flags;
		slot = 0;
	} else {
		if ((uint32_t)src_join >= p->k || src_col >= p->hts[src_join]->n_payload) {
			POLR_FAIL(ctx, POLR_E_INVALID, "build column (%d,%u) out of range", src_join, src_col);
		}
		const polr_ht *ht = p->hts[src_join];
		const OwnedCol &c = ht->kind == KIND_PERFECT ? ht->pcols[src_col] : ht->payload[src_col];
		src.data = c.data;
		src.valid = c.valid;
		src.width = c.width;
		src.flags = c.flags;
		slot = 1 + (uint32_t)src_join;
	}
	if (o->n_rows == 0) {
		return POLR_OK;
	}
	const bool to_device = (dst_flags & POLR_COL_DEVICE) != 0;
	uint8_t *d_data = (uint8_t *)dst_data, *d_valid = dst_valid;
	uint8_t *tmp_data = nullptr, *tmp_valid = nullptr;
	if (!to_device) {
		HIPCHK(ctx, hipMalloc((void **)&tmp_data, o->n_rows * src.width));
		d_data = tmp_data;
		if (dst_valid) {
			hipError_t e = hipMalloc((void **)&tmp_valid, o->n_rows);
			if (e != hipSuccess) {
				hipFree(tmp_data);
				POLR_FAIL(ctx, POLR_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
			}
			d_valid = tmp_valid;
		}
	}
	polr_launch_gather(st, o->dev, o->chunk_base, o->n_chunks, slot, src, d_data, d_valid);
	hipError_t e = hipSuccess;
	if (!to_device) {
		e = hipMemcpyAsync(dst_data, tmp_data, o->n_rows * src.width, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess && dst_valid) {
			e = hipMemcpyAsync(dst_valid, tmp_valid, o->n_rows, hipMemcpyDeviceToHost, st);
		}
		e = e == hipSuccess ? hipStreamSynchronize(st) : e;
		hipFree(tmp_data);
		if (tmp_valid) {
			hipFree(tmp_valid);
		}
	}
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "materialize failed: %s", hipGetErrorString(e));
	}
	return POLR_OK;
}

void polr_out_destroy(polr_out *o) {
	POLR_ENTRY();
	if (!o) {
		return;
	}
	hipSetDevice(o->ctx->device);
	if (o->dev.ids) {
		hipFree(o->dev.ids);
	}
	if (o->dev.chunk_count) {
		hipFree(o->dev.chunk_count);
	}
	if (o->dev.cursor) {
		hipFree(o->dev.cursor);
	}
	if (o->chunk_base) {
		hipFree(o->chunk_base);
	}
	if (o->total_dev) {
		hipFree(o->total_dev);
	}
	if (o->fused_dev) {
		hipFree(o->fused_dev);
	}
	if (o->fused_cells) {
		hipFree(o->fused_cells);
	}
	if (o->fused_dropped) {
		hipFree(o->fused_dropped);
	}
	polr_ctx *ctx_ = o->ctx;
	delete o;
	polr_ctx_release(ctx_);
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------
// Probe launches
// ---------------------------------------------------------------------------------------------------
// Launch geometry: 256-thread workgroups (4 independent waves); the grid never exceeds what is
// resident at once (occupancy of the instantiation x CUs), waves grid-stride over units of 64..1024
// tuples sized so that a small routing round still spreads over the chip while a table-sized round
// gives every resident wave a few units.
uint32_t polr_waves_per_block(polr_pipeline *p, bool materialize) {
	const DevPipeline &dp = materialize ? p->host_mat : p->host_count;
	uint32_t &cached = materialize ? p->wpb_mat : p->wpb_count;
	if (cached == 0) {
		for (uint32_t w : {4u, 2u, 1u}) {
			// leave room for at least two workgroups per CU when possible
			if (polr_path_lds_bytes(dp.k, dp.W, w) <= (w == 1 ? 160u * 1024 : 80u * 1024)) {
				cached = w;
				break;
			}
		}
	}
	return cached;
}

uint32_t polr_resident_waves(polr_pipeline *p, bool materialize) {
	const DevPipeline &dp = materialize ? p->host_mat : p->host_count;
	const uint32_t wpb = polr_waves_per_block(p, materialize);
	int &cached = materialize ? p->blocks_per_cu_mat : p->blocks_per_cu_count;
	if (cached == 0) {
		cached = polr_path_occupancy(dp.k, dp.W, wpb);
		if (cached > 8) {
			cached = 8;
		}
	}
	return (uint32_t)p->ctx->n_cus * (uint32_t)cached * wpb;
}

int polr_plan_launch(polr_pipeline *p, bool materialize, uint64_t total_tuples, uint32_t *unit_size,
                     uint32_t *n_blocks_max) {
	const uint32_t wpb = polr_waves_per_block(p, materialize);
	if (wpb == 0) {
		p->ctx->err = "per-wave LDS queues exceed 160 KB (too many joins x carried ids)";
		return POLR_E_UNSUPPORTED;
	}
	const uint64_t waves = polr_resident_waves(p, materialize);
	uint64_t us = (total_tuples + waves - 1) / waves;
	us = ((us + 63) / 64) * 64;
	us = std::min<uint64_t>(std::max<uint64_t>(us, 64), 2048);
	*unit_size = (uint32_t)us;
	*n_blocks_max = (uint32_t)(waves / wpb);
	return POLR_OK;
}

extern "C" {

int polr_probe_rounds_async(polr_pipeline *p, void *stream, const polr_round *rounds, uint32_t n_rounds,
                            polr_out *out, uint64_t *counts_dev) {
	POLR_ENTRY();
	if (!p || !rounds || !counts_dev || n_rounds == 0) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	if (out && out->pipe != p) {
		POLR_FAIL(ctx, POLR_E_INVALID, "output object belongs to another pipeline");
	}
	uint64_t total = 0;
	for (uint32_t r = 0; r < n_rounds; r++) {
		if (rounds[r].path >= p->n_paths) {
			POLR_FAIL(ctx, POLR_E_INVALID, "round %u: path %u out of range", r, rounds[r].path);
		}
		if (rounds[r].begin > p->n_tuples || rounds[r].count > p->n_tuples - rounds[r].begin) {
			POLR_FAIL(ctx, POLR_E_INVALID, "round %u: tuples [%llu, +%llu) outside the %llu tuples of the source", r,
			          (unsigned long long)rounds[r].begin, (unsigned long long)rounds[r].count,
			          (unsigned long long)p->n_tuples);
		}
		total += rounds[r].count;
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	if (total == 0) {
		HIPCHK(ctx, hipMemsetAsync(counts_dev, 0, (uint64_t)n_rounds * p->k * 8, st));
		return POLR_OK;
	}
	const uint64_t need_shards = (uint64_t)n_rounds * POLR_NSHARD * p->k;
	if (need_shards > p->shards_cap) {
		if (p->shards_dev) {
			HIPCHK(ctx, hipStreamSynchronize(st));
			hipFree(p->shards_dev);
			p->shards_dev = nullptr;
		}
		const uint64_t cap = std::max<uint64_t>(need_shards, 64 * POLR_NSHARD * POLR_KMAX);
		HIPCHK(ctx, hipMalloc((void **)&p->shards_dev, cap * 8));
		p->shards_cap = cap;
	}
	HIPCHK(ctx, hipMemsetAsync(p->shards_dev, 0, need_shards * 8, st));
	const bool materialize = out != nullptr;
	uint32_t unit_size, max_blocks;
	int rc = polr_plan_launch(p, materialize, total, &unit_size, &max_blocks);
	if (rc) {
		return rc;
	}
	if (n_rounds > p->rounds_cap) {
		if (p->rounds_dev) {
			HIPCHK(ctx, hipStreamSynchronize(st));
			hipFree(p->rounds_dev);
			hipFree(p->prefix_dev);
			hipFree(p->unit_sizes_dev);
			p->rounds_dev = nullptr;
			p->prefix_dev = nullptr;
			p->unit_sizes_dev = nullptr;
		}
		const uint32_t cap = std::max<uint32_t>(n_rounds, 64);
		HIPCHK(ctx, hipMalloc((void **)&p->rounds_dev, (uint64_t)cap * sizeof(DevRound)));
		HIPCHK(ctx, hipMalloc((void **)&p->prefix_dev, ((uint64_t)cap + 1) * 8));
		HIPCHK(ctx, hipMalloc((void **)&p->unit_sizes_dev, (uint64_t)cap * 4));
		p->rounds_cap = cap;
	}
	std::vector<uint64_t> prefix(n_rounds + 1);
	std::vector<uint32_t> usizes(n_rounds, unit_size);
	prefix[0] = 0;
	for (uint32_t r = 0; r < n_rounds; r++) {
		prefix[r + 1] = prefix[r] + (rounds[r].count + unit_size - 1) / unit_size;
	}
	static_assert(sizeof(DevRound) == sizeof(polr_round), "round layout");
	// pageable source: the runtime stages it before returning, so the caller's array may be reused
	HIPCHK(ctx, hipMemcpyAsync(p->rounds_dev, rounds, (uint64_t)n_rounds * sizeof(DevRound), hipMemcpyHostToDevice, st));
	HIPCHK(ctx, hipMemcpyAsync(p->prefix_dev, prefix.data(), ((uint64_t)n_rounds + 1) * 8, hipMemcpyHostToDevice, st));
	HIPCHK(ctx, hipMemcpyAsync(p->unit_sizes_dev, usizes.data(), (uint64_t)n_rounds * 4, hipMemcpyHostToDevice, st));
	HIPCHK(ctx, hipStreamSynchronize(st)); // prefix is a local vector
	const uint64_t total_units = prefix[n_rounds];
	const uint32_t wpb = polr_waves_per_block(p, materialize);
	const uint32_t n_blocks = (uint32_t)std::min<uint64_t>(max_blocks, (total_units + wpb - 1) / wpb);
	DevOut dout;
	memset(&dout, 0, sizeof(dout));
	if (out) {
		dout = out->dev;
		out->stats_valid = false;
	}
	const DevPipeline &dp = materialize ? p->host_mat : p->host_count;
	hipError_t e = polr_launch_path_kernel(dp.W, dp.k, n_blocks, wpb, st, materialize ? p->dev_mat : p->dev_count,
	                                       p->rounds_dev, p->prefix_dev, n_rounds, p->unit_sizes_dev, dout,
	                                       p->shards_dev, SelfRoute {});
	if (e != hipSuccess) {
		POLR_FAIL(ctx, POLR_E_HIP, "path kernel launch failed: %s", hipGetErrorString(e));
	}
	polr_launch_reduce_counts(st, p->shards_dev, n_rounds, p->k, (unsigned long long *)counts_dev);
	return POLR_OK;
}

int polr_probe_rounds(polr_pipeline *p, void *stream, const polr_round *rounds, uint32_t n_rounds, polr_out *out,
                      uint64_t *counts) {
	POLR_ENTRY();
	if (!p || !counts) {
		return POLR_E_INVALID;
	}
	polr_ctx *ctx = p->ctx;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	const uint64_t need = (uint64_t)n_rounds * p->k;
	if (need > p->counts_cap) {
		if (p->counts_dev) {
			hipFree(p->counts_dev);
			p->counts_dev = nullptr;
		}
		const uint64_t cap = std::max<uint64_t>(need, 256);
		HIPCHK(ctx, hipMalloc((void **)&p->counts_dev, cap * 8));
		p->counts_cap = cap;
	}
	int rc = polr_probe_rounds_async(p, stream, rounds, n_rounds, out, (uint64_t *)p->counts_dev);
	if (rc) {
		return rc;
	}
	HIPCHK(ctx, hipMemcpyAsync(counts, p->counts_dev, need * 8, hipMemcpyDeviceToHost, st));
	HIPCHK(ctx, hipStreamSynchronize(st));
	if (out) {
		uint32_t cur[2];
		HIPCHK(ctx, hipMemcpy(cur, out->dev.cursor, 8, hipMemcpyDeviceToHost));
		if (cur[1]) {
			POLR_FAIL(ctx, POLR_E_OVERFLOW, "output needs more than %u chunks of %u rows (counters are exact)",
			          out->dev.max_chunks, out->dev.chunk_capacity);
		}
	}
	return POLR_OK;
}

} // extern "C"
