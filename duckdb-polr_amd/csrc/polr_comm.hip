// duckdb-polr_amd/csrc/polr_comm.hip -- the one exchange step of the path: broadcasting a finalized build side.
//
// The reference has no multi-process code at all; one POLAR pipeline per GPU with its own multiplexer
// state is the counterpart of one PipelineExecutor per worker thread (src/parallel/pipeline.cpp:145-174), and the only
// data every pipeline needs and only one has is the build side JoinHashTable::Finalize left behind
// (src/execution/join_hashtable.cpp:324-377).  polr_bcast_build ships it -- metadata blob, then every device buffer
// in place -- with ncclBroadcast over xGMI (RCCL), once per query before probing starts; nothing else of the path
// communicates.  librccl is loaded on first use (dlopen): the library has no link-time dependency on it, and a process
// that never creates a communicator never touches it.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "polr_internal.h"

namespace {

// the six RCCL entry points used (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260,
// ncclGetErrorString :339, ncclBroadcast :591, ncclAllReduce :611)
typedef struct {
	char internal[POLR_COMM_ID_BYTES];
} rccl_unique_id;
typedef void *rccl_comm_t;
typedef int (*fn_get_unique_id)(rccl_unique_id *);
typedef int (*fn_comm_init_rank)(rccl_comm_t *, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(rccl_comm_t);
typedef const char *(*fn_get_error_string)(int);
typedef int (*fn_broadcast)(const void *, void *, size_t, int /* ncclDataType_t */, int, rccl_comm_t, hipStream_t);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int /* ncclDataType_t */, int /* ncclRedOp_t */, rccl_comm_t,
                             hipStream_t);
typedef int (*fn_comm_abort)(rccl_comm_t);
enum { RCCL_UINT8 = 1, RCCL_INT32 = 2 }; // rccl.h:459-462: ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2
enum { RCCL_MIN = 3 };                   // rccl.h:448-451: ncclSum 0, ncclProd 1, ncclMax 2, ncclMin 3

struct Rccl {
	void *lib = nullptr;
	fn_get_unique_id get_unique_id = nullptr;
	fn_comm_init_rank comm_init_rank = nullptr;
	fn_comm_destroy comm_destroy = nullptr;
	fn_get_error_string get_error_string = nullptr;
	fn_broadcast broadcast = nullptr;
	fn_all_reduce all_reduce = nullptr;
	fn_comm_abort comm_abort = nullptr; // (optional)
	std::string err;
};

Rccl &rccl() {
	static Rccl r = [] {
		Rccl x;
		// The ROCm installation's RCCL first, by path and RTLD_LOCAL: a host process may already carry ANOTHER copy of RCCL
		// (PyTorch bundles one, linked against its own bundled HIP runtime) -- a dlopen by soname would hand that copy
		// back, and the streams of this library belong to the HIP runtime THIS library links
		for (const char *name : {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"}) {
			x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (x.lib) {
				break;
			}
		}
		if (!x.lib) {
			x.err = std::string("librccl.so not found: ") + dlerror();
			return x;
		}
		x.get_unique_id = (fn_get_unique_id)dlsym(x.lib, "ncclGetUniqueId");
		x.comm_init_rank = (fn_comm_init_rank)dlsym(x.lib, "ncclCommInitRank");
		x.comm_destroy = (fn_comm_destroy)dlsym(x.lib, "ncclCommDestroy");
		x.get_error_string = (fn_get_error_string)dlsym(x.lib, "ncclGetErrorString");
		x.broadcast = (fn_broadcast)dlsym(x.lib, "ncclBroadcast");
		x.all_reduce = (fn_all_reduce)dlsym(x.lib, "ncclAllReduce");
		x.comm_abort = (fn_comm_abort)dlsym(x.lib, "ncclCommAbort");
		if (!x.get_unique_id || !x.comm_init_rank || !x.comm_destroy || !x.get_error_string || !x.broadcast ||
		    !x.all_reduce) {
			x.err = "librccl.so lacks a symbol of the NCCL API";
		}
		return x;
	}();
	return r;
}

} // namespace

struct polr_comm {
	polr_ctx *ctx = nullptr;
	rccl_comm_t comm = nullptr;
	int world = 1, rank = 0;
	unsigned long long *scratch = nullptr; // device: [0] = size of the metadata blob; then the blob; behind it a status word
	uint64_t bytes_broadcast = 0;
};
#define POLR_COMM_SCRATCH 8192

#define RCCLCHK(ctx_, call_)                                                                                           \
	do {                                                                                                               \
		int r_ = (call_);                                                                                              \
		if (r_ != 0) {                                                                                                 \
			POLR_FAIL(ctx_, POLR_E_HIP, "%s failed: %s", #call_, rccl().get_error_string(r_));                         \
		}                                                                                                              \
	} while (0)

extern "C" {

int polr_comm_get_unique_id(void *id) {
	if (!id) {
		return POLR_E_INVALID;
	}
	Rccl &r = rccl();
	if (!r.err.empty()) {
		return POLR_E_UNSUPPORTED;
	}
	rccl_unique_id u;
	if (r.get_unique_id(&u) != 0) {
		return POLR_E_HIP;
	}
	memcpy(id, u.internal, POLR_COMM_ID_BYTES);
	return POLR_OK;
}

int polr_comm_create(polr_ctx *ctx, const void *id, int world_size, int rank, polr_comm **out) {
	POLR_ENTRY();
	if (!ctx || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) {
		return POLR_E_INVALID;
	}
	*out = nullptr;
	Rccl &r = rccl();
	if (!r.err.empty()) {
		POLR_FAIL(ctx, POLR_E_UNSUPPORTED, "%s", r.err.c_str());
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	polr_comm *c = new polr_comm();
	c->ctx = polr_ctx_retain(ctx);
	c->world = world_size;
	c->rank = rank;
	rccl_unique_id u;
	memcpy(u.internal, id, POLR_COMM_ID_BYTES);
	int rc = r.comm_init_rank(&c->comm, world_size, u, rank);
	if (rc != 0) {
		polr_ctx_release(c->ctx);
		delete c;
		POLR_FAIL(ctx, POLR_E_HIP, "ncclCommInitRank failed: %s", r.get_error_string(rc));
	}
	hipError_t e = hipMalloc((void **)&c->scratch, POLR_COMM_SCRATCH + 64); // (+ the status word of polr_bcast_build)
	if (e != hipSuccess) {
		r.comm_destroy(c->comm);
		polr_ctx_release(c->ctx);
		delete c;
		POLR_FAIL(ctx, POLR_E_HIP, "communicator scratch: %s", hipGetErrorString(e));
	}
	*out = c;
	return POLR_OK;
}

// root: *ht is the finalized table to send (unchanged).  Every other rank: *ht receives a new table of the same shape,
// owned by the caller (polr_ht_destroy).  Collective: every rank of the communicator calls it, in the same order.
//
// Failures are collective too: whatever goes wrong on ONE rank before the table buffers travel -- the root has no
// finalized table, its export fails or its metadata does not fit the scratch block; a receiver cannot allocate the
// table or reads metadata that make no sense -- every rank still runs the same sequence of collectives
//   (1) broadcast of the metadata block   (size 0 = "the root has nothing to send")
//   (2) all-reduce (min) of one status word per rank, after every rank has its table and its buffer list
//   (3) one broadcast per table buffer, only when the status of every rank is 1
// and every rank returns an error from the same step; nobody is left blocked inside a collective the others never
// enter.  (A failing RCCL or HIP call itself is not recoverable this way: the communicator is then unusable.)
int polr_bcast_build(polr_comm *comm, polr_ht **ht, int root, void *stream) {
	POLR_ENTRY();
	if (!comm || !ht || root < 0 || root >= comm->world) {
		return POLR_E_INVALID; // (a caller bug, the same on every rank: arguments are checked before anything travels)
	}
	polr_ctx *ctx = comm->ctx;
	Rccl &r = rccl();
	HIPCHK(ctx, hipSetDevice(ctx->device));
	hipStream_t st = polr_stream(ctx, stream);
	const bool is_root = comm->rank == root;
	int local_rc = POLR_OK;
	std::string local_msg;
	auto fail_local = [&](int rc, const std::string &msg) {
		if (local_rc == POLR_OK) {
			local_rc = rc;
			local_msg = msg;
		}
	};
	// (1) the metadata blob: its size, then the bytes
	std::vector<uint8_t> meta;
	unsigned long long meta_bytes = 0;
	if (is_root) {
		if (!*ht || (*ht)->kind == KIND_NONE) {
			fail_local(POLR_E_INVALID, "broadcast root has no finalized build side");
		} else if (!(*ht)->heaps.empty()) {
			// (the cells of a VARCHAR column hold addresses of the root's heap)
			fail_local(POLR_E_UNSUPPORTED, "a build side with VARCHAR columns (string heaps) is not broadcast: upload it on every rank");
		} else {
			uint64_t mb = 0;
			uint32_t nb = 0;
			int rc = polr_ht_export(*ht, nullptr, &mb, nullptr, nullptr, &nb);
			if (!rc) {
				meta.resize(mb);
				std::vector<void *> ptrs0(nb);
				std::vector<uint64_t> sizes0(nb);
				rc = polr_ht_export(*ht, meta.data(), &mb, ptrs0.data(), sizes0.data(), &nb);
			}
			if (rc) {
				fail_local(rc, std::string("export of the build side failed: ") + polr_last_error(ctx));
			} else if (mb == 0 || mb + 8 > POLR_COMM_SCRATCH) {
				fail_local(POLR_E_UNSUPPORTED, "build-side metadata of " + std::to_string(mb) + " bytes");
			} else {
				meta_bytes = mb;
			}
		}
		HIPCHK(ctx, hipMemcpyAsync(comm->scratch, &meta_bytes, 8, hipMemcpyHostToDevice, st));
		if (meta_bytes) {
			HIPCHK(ctx, hipMemcpyAsync(comm->scratch + 1, meta.data(), meta_bytes, hipMemcpyHostToDevice, st));
		}
	}
	RCCLCHK(ctx, r.broadcast(comm->scratch, comm->scratch, POLR_COMM_SCRATCH, RCCL_UINT8, root, comm->comm, st));
	polr_ht *fresh = nullptr;
	if (!is_root) {
		HIPCHK(ctx, hipMemcpyAsync(&meta_bytes, comm->scratch, 8, hipMemcpyDeviceToHost, st));
		HIPCHK(ctx, hipStreamSynchronize(st));
	}
	if (meta_bytes == 0) {
		// the root had nothing to send: every rank leaves here, after the one collective all of them have entered
		if (is_root) {
			POLR_FAIL(ctx, local_rc, "%s", local_msg.c_str());
		}
		POLR_FAIL(ctx, POLR_E_INVALID, "the broadcast root (rank %d) has no build side to send", root);
	}
	if (!is_root) {
		if (meta_bytes + 8 > POLR_COMM_SCRATCH) {
			fail_local(POLR_E_HIP, "broadcast metadata corrupt (" + std::to_string(meta_bytes) + " bytes)");
		} else {
			meta.resize(meta_bytes);
			HIPCHK(ctx, hipMemcpy(meta.data(), comm->scratch + 1, meta_bytes, hipMemcpyDeviceToHost));
			int rc = polr_ht_alloc_like(ctx, meta.data(), meta_bytes, &fresh);
			if (rc) {
				fresh = nullptr;
				fail_local(rc, std::string("receiving table: ") + polr_last_error(ctx));
			}
		}
	}
	// the buffer list of this rank's table (the root's own, a receiver's fresh one)
	polr_ht *mine = is_root ? *ht : fresh;
	uint64_t mb = 0;
	uint32_t nb = 0;
	std::vector<void *> ptrs;
	std::vector<uint64_t> sizes;
	if (mine) {
		int rc = polr_ht_export(mine, nullptr, &mb, nullptr, nullptr, &nb);
		if (!rc) {
			std::vector<uint8_t> meta2(mb);
			ptrs.resize(nb);
			sizes.resize(nb);
			rc = polr_ht_export(mine, meta2.data(), &mb, ptrs.data(), sizes.data(), &nb);
		}
		if (rc) {
			fail_local(rc, std::string("buffer list of the table: ") + polr_last_error(ctx));
		}
	}
	// (2) agree: 1 only if every rank is ready for the buffers
	int status = local_rc == POLR_OK ? 1 : 0;
	int *status_dev = (int *)((uint8_t *)comm->scratch + POLR_COMM_SCRATCH);
	HIPCHK(ctx, hipMemcpyAsync(status_dev, &status, sizeof(int), hipMemcpyHostToDevice, st));
	RCCLCHK(ctx, r.all_reduce(status_dev, status_dev, 1, RCCL_INT32, RCCL_MIN, comm->comm, st));
	int agreed = 0;
	HIPCHK(ctx, hipMemcpyAsync(&agreed, status_dev, sizeof(int), hipMemcpyDeviceToHost, st));
	HIPCHK(ctx, hipStreamSynchronize(st));
	if (agreed != 1) {
		if (fresh) {
			polr_ht_destroy(fresh);
		}
		if (local_rc != POLR_OK) {
			POLR_FAIL(ctx, local_rc, "%s", local_msg.c_str());
		}
		POLR_FAIL(ctx, POLR_E_HIP, "another rank could not take part in the broadcast of this build side");
	}
	if (!is_root) {
		*ht = fresh;
	}
	// (3) every device buffer of the table, in place
	for (uint32_t i = 0; i < nb; i++) {
		if (sizes[i] == 0) {
			continue;
		}
		RCCLCHK(ctx, r.broadcast(ptrs[i], ptrs[i], sizes[i], RCCL_UINT8, root, comm->comm, st));
		comm->bytes_broadcast += sizes[i];
	}
	HIPCHK(ctx, hipStreamSynchronize(st));
	return POLR_OK;
}

int polr_comm_bytes_broadcast(const polr_comm *comm, uint64_t *bytes) {
	if (!comm || !bytes) {
		return POLR_E_INVALID;
	}
	*bytes = comm->bytes_broadcast;
	return POLR_OK;
}

void polr_comm_destroy(polr_comm *comm) {
	POLR_ENTRY();
	if (!comm) {
		return;
	}
	hipSetDevice(comm->ctx->device);
	if (comm->comm) {
		rccl().comm_destroy(comm->comm);
	}
	if (comm->scratch) {
		hipFree(comm->scratch);
	}
	polr_ctx *ctx_ = comm->ctx;
	delete comm;
	polr_ctx_release(ctx_);
}

} // extern "C"
