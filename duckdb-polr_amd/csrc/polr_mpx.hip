// duckdb-polr_amd/csrc/polr_mpx.hip -- device-resident multiplexer (router kernel) -- placeholder
// until the router lands; the entry points exist so the ABI is complete and fail loudly.
#include "polr_internal.h"

extern "C" {
int polr_mpx_create(polr_pipeline *p, const polr_mpx_config *, polr_mpx **out) {
	if (!p || !out) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	POLR_FAIL(p->ctx, POLR_E_UNSUPPORTED, "device-resident multiplexer not built yet");
}
int polr_mpx_run(polr_mpx *, void *, uint64_t, uint64_t, polr_out *) {
	return POLR_E_INVALID;
}
int polr_mpx_set_chunk_offsets(polr_mpx *, const uint64_t *, uint64_t) {
	return POLR_E_INVALID;
}
int polr_mpx_finish(polr_mpx *, void *, polr_mpx_stats *) {
	return POLR_E_INVALID;
}
int polr_mpx_fetch_log(polr_mpx *, void *, uint32_t *, uint64_t *, uint64_t *, uint64_t, uint64_t *) {
	return POLR_E_INVALID;
}
void polr_mpx_destroy(polr_mpx *) {
}
}
