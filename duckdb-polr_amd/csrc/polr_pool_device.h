// duckdb-polr_amd/csrc/polr_pool_device.h -- "the whole run in one launch": device-side protocol between the
// routers (one wave per executor, multiplexer state in LDS for the whole run) and a POOL of probe waves that
// serve the rounds of ALL executors.
//
// Reference counterpart: one POLARPipelineExecutor per worker thread, each with its own MultiplexerState,
// pulling source chunks and calling RunPath per routed slice (src/parallel/polar_pipeline_executor.cpp:255-538,
// src/parallel/pipeline.cpp:145-174).  Here an *executor* is only the routing half of that object -- the
// multiplexer state, its share of the source chunks, the reward feedback -- and owns no compute: every round it
// routes is cut into units and queued; whichever probe wave is free takes the next unit.  The exploration rounds
// of one executor therefore run beside the table-sized rounds of the others instead of idling a fixed slice of
// the grid, and a skewed source range only makes its executor finish later, not its share of the device.
//
//   rings     POLR_POOL_RINGS (64) x {hi, mid, lo}: FIFO of 16-byte unit entries.  Probe wave g of the pool uses ring
//             g % 64 (every ring the same number of waves).  64 rings keep the pollers of one control line few: with
//             8 rings, 512 idle waves per ring answered every small round with a storm of compare-and-swaps on one word
//             -- measured 8 us per round, serialised over all executors.  Routers deal the units of a round round-robin
//             over the rings.  hi = rounds of <= POLR_POOL_HI_TUPLES tuples (exploration slices: latency-critical,
//             taken first), mid = bigger rounds whose counters their executor waits for, lo = terminal rounds
//             (PoolRoundOut).
//   entry     two 8-byte granules, each carrying the lap tag of its ticket, written and read with relaxed
//             agent-scope 8-byte atomics (self-validating, no fences):
//               g0 = tag:16 | kind:2 | slot:2 | emit:1 | path:5 | 0:1 | level:5 | begin:32
//               g1 = tag:16 | exec:16 | count:32
//   tickets   mid, lo: a wave takes the next ticket with one returning atomic add on the queue's head and looks for the
//             entry of that ticket (tickets past the tail are simply not written yet; it keeps the ticket while it
//             serves other queues); hi: never waited on -- taken with a CAS on hi_head only while hi_head < hi_tail,
//             and only by the waves whose number matches the ticket's in the low `lottery` bits.  heads/tails are
//             monotonic across runs (tags never repeat within 65535 laps).
//   slots     POLR_SLOTS (4) rounds of one executor can be in flight: the front one decided, the others rehearsed
//             ahead on a shadow of the state while their decisions cannot depend on intermediates still outstanding
//             (polr_can_speculate); each slot has its own counter bank and arrival counters.
//   arrival   a wave adds its k stage counters to the executor's bank [slot][ring & 7][k] with returning atomics, then
//             the unit's tokens (POLR_POOL_TOKENS) to arrived[slot][ring & 7]; the router waits for the sum of the 8 shards
//             to reach the tokens of the units it has published in that slot, exchanges the counters with 0 and routes.
//   exit      the router that finishes last publishes one EXIT entry per worker wave of every ring (lo).
//   watchdog  every wait is bounded (POLR_RES_TIMEOUT_TICKS); a timeout raises `abort` in the run header, which every
//             waiter polls: a lost wave is an error (POLR_E_HIP from polr_mpx_finish), never a hang.
#pragma once

#include "polr_mpx_device.h"

#ifdef POLR_DIAG_TIMELINE
// diagnostic build: router time by phase (ticks of the 100 MHz clock, summed over all routers and steps):
// [0] waiting for the front round, [1] absorbing its counters, [2] the real routing step, [3] publishing,
// [4] rehearsals (route + publish), [5] steps, [6] entry .. first publish, [7] routers,
// [8] entry .. state in LDS, [9] .. state initialised, [10] .. arrival targets known (= first step begins)
// (summed in registers, one set of atomics per router when it leaves: per-step atomics on eleven words slowed every
// atomic of the run down)
#define RT_T(v_) const unsigned long long v_ = wall_clock64();
#define RT_DECL unsigned long long rt_acc[11] = {};
#define RT_ADD(i_, d_) rt_acc[i_] += (unsigned long long)(d_);
#define RT_FLUSH                                                                                                       \
	if (lane == 0) {                                                                                                   \
		_Pragma("unroll") for (int i_ = 0; i_ < 11; i_++) {                                                            \
			atomicAdd(&polr_diag_router[i_], rt_acc[i_]);                                                              \
		}                                                                                                              \
	}
#else
#define RT_T(v_)
#define RT_DECL
#define RT_ADD(i_, d_)
#define RT_FLUSH
#endif

#define POLR_POOL_RINGS 64 // unit queues; counters and arrivals are sharded 8 ways (ring & 7)
#define POLR_POOL_SHARDS 8
#define POLR_POOL_HI_TUPLES 4096u // rounds up to this many tuples are latency-critical (exploration slices)
#define POLR_POOL_HI_UNIT 64u // smallest unit of a small round (ring capacities are sized for it)
#ifndef POLR_POOL_LOTTERY_PATIENCE
#define POLR_POOL_LOTTERY_PATIENCE 64u // idle polls (> 100 us) after which a wave tries for ANY hi ticket, every 16th poll
#endif
#define POLR_POOL_KIND_WORK 1u
#define POLR_POOL_KIND_EXIT 2u
#define POLR_POOL_KIND_CONT 3u // shared work of the generic pipeline: tuples that wait in front of a stage (polr_poolg.hip)
// Arrivals are counted in TOKENS: a unit published by a router is worth POLR_POOL_TOKENS; a probe wave that gives part
// of its unit to the pool (work sharing, polr_poolg.hip) halves what its own arrival is worth and hands the other half
// on with the piece, `level` halvings deep (entry bits 32..36) -- the router waits for the tokens of the units it
// published, however they were cut up on the way.
#define POLR_POOL_TOKENS 65536ull
#define POLR_POOL_MAX_LEVEL 16u

struct PoolRingCtl { // one 128-byte line each
	unsigned long long lo_head, pad0[15];
	unsigned long long lo_tail, pad1[15];
	unsigned long long hi_head, hi_tail, pad2[14];   // (read together by the pollers)
	unsigned long long mid_head, mid_tail, pad3[14];
};

struct PoolEntry {
	unsigned long long g0, g1;
};

// Persistent object of a set of executors that run together (owned by the first multiplexer of the set).
struct PoolSync {
	PoolRingCtl ctl[POLR_POOL_RINGS];
	uint32_t lo_cap, hi_cap; // entries per ring (powers of two)
	uint32_t pad[30];
	// followed by: PoolEntry lo[POLR_POOL_RINGS][lo_cap], mid[POLR_POOL_RINGS][lo_cap], hi[POLR_POOL_RINGS][hi_cap]
};

__device__ __forceinline__ PoolEntry *polr_pool_lo(PoolSync *s, uint32_t ring, uint32_t lo_cap) {
	return (PoolEntry *)(s + 1) + (size_t)ring * lo_cap;
}
__device__ __forceinline__ PoolEntry *polr_pool_mid(PoolSync *s, uint32_t ring, uint32_t lo_cap) {
	return (PoolEntry *)(s + 1) + (size_t)POLR_POOL_RINGS * lo_cap + (size_t)ring * lo_cap;
}
__device__ __forceinline__ PoolEntry *polr_pool_hi(PoolSync *s, uint32_t ring, uint32_t lo_cap, uint32_t hi_cap) {
	return (PoolEntry *)(s + 1) + (size_t)2 * POLR_POOL_RINGS * lo_cap + (size_t)ring * hi_cap;
}

// Per-run header, rewritten by the host with every launch (copied with the executor descriptors).
struct PoolRun {
	PoolSync *sync;
	uint32_t n_exec;
	uint32_t n_router_blocks;
	uint32_t pool_waves;                    // all probe waves
	uint32_t lo_cap, hi_cap;                // entries per ring (powers of two), as in *sync
	uint32_t hi_tuples;                     // rounds of up to this many tuples go to the hi queues
	uint32_t units_x;                       // a big round is cut into about units_x * pool_waves / (executors routing) units
	uint32_t hi_unit;                       // tuples per unit of a small round
	uint32_t hi_lottery;                    // power of two <= the probe waves of a ring: wave w of a ring tries for hi ticket t
	                                        // only if w % hi_lottery == t % hi_lottery (bounds the compare-and-swap storm
	                                        // a published unit sets off among the pollers of its ring)
	uint32_t idle_sleep;                    // longest s_sleep argument of an idle probe wave's back-off (16 .. 127)
	uint32_t n_rings;                       // rings in use: a power of two <= min(POLR_POOL_RINGS, probe workgroups), so that
	                                        // every ring has waves that serve it
	uint32_t routers_done;                  // device: routers that have finished
	uint32_t abort;                         // device: a watchdog fired
	volatile uint32_t *host_words;          // pinned words of the run's FIRST multiplexer (it owns the rings): [2] = 1 when
	                                        // the run was given up, whichever router saw it
	unsigned long long timeout_ticks;       // watchdog: longest wait, ticks of the 100 MHz wall clock
	uint32_t *share_recs;                   // work sharing (generic pipeline): one record of share_stride dwords per probe wave,
	uint32_t *share_flags;                  // and its "full" flag; nullptr: off
	uint32_t share_stride, share_after;     // a wave shares after share_after steps on one unit, and again after as many
	uint32_t worker_waves[POLR_POOL_RINGS]; // probe waves that poll ring r (EXIT entries to publish); read in place
};

__device__ __forceinline__ uint32_t polr_pool_tag(unsigned long long ticket, uint32_t cap) {
	return (uint32_t)((ticket / cap) % 65535ull) + 1u;
}

__device__ __forceinline__ unsigned long long polr_pool_g0(uint32_t tag, uint32_t kind, uint32_t slot, uint32_t emit,
                                                           uint32_t path, uint32_t begin, uint32_t level = 0) {
	return ((unsigned long long)tag << 48) | ((unsigned long long)(kind & 3u) << 46) |
	       ((unsigned long long)(slot & 3u) << 44) | ((unsigned long long)(emit & 1u) << 43) |
	       ((unsigned long long)(path & 31u) << 38) | ((unsigned long long)(level & 31u) << 32) | begin;
}
__device__ __forceinline__ unsigned long long polr_pool_g1(uint32_t tag, uint32_t exec, uint32_t count) {
	return ((unsigned long long)tag << 48) | ((unsigned long long)(exec & 0xFFFFu) << 32) | count;
}

// a unit as a probe wave sees it
struct PoolUnit {
	uint32_t kind, slot, emit, path, exec, begin, count, level;
};

// what a router has decided for one round, as the publisher needs it
// cls: which queue a round goes to.  0 = hi: a small round (an exploration slice) -- latency-critical, taken first;
// 1 = mid: a bigger round whose counters the executor's next decision waits for -- taken before ...
// 2 = lo: a TERMINAL round, after which its executor routes nothing any more (DEFAULT_PATH, INIT_ONCE after its init
// phase, ADAPTIVE_REINIT below its resistance tolerance: num_cache_flushing_skips = "never again").  Terminal rounds are
// the bulk of a table-sized run and nobody waits for them: FIFO behind them, an executor in the middle of its
// explore / exploit windows would sit out the whole backlog at every window.
struct PoolRoundOut {
	uint32_t begin, count, path, emit, unit, n_units, cls;
};

// unit size of a round: small rounds are cut into units of hi_unit tuples (PoolRun: 1 024 for the flat pipeline, 256 for
// the generic one -- a unit costs its wave the same chain of dependent round trips whatever its size, and with hundreds
// of executors exploring that wave time is what the pool runs out of); big rounds are cut so that one executor's
// round gives every probe wave of its share of the pool a few units, in multiples of `gran` tuples
__device__ __forceinline__ void polr_pool_size_units(uint64_t tuples64, uint32_t pool_waves, uint32_t n_exec, uint32_t gran,
                                                     uint32_t hi_tuples, uint32_t units_x, uint32_t hi_unit, bool terminal,
                                                     PoolRoundOut &r) {
	// (32-bit arithmetic: a round is at most 2^32 - 1 tuples, and the device has no integer divider -- a 64-bit
	// division is a ~150-instruction sequence on the router's single lane; gran is a power of two)
	const uint32_t tuples = (uint32_t)tuples64;
	if (tuples <= hi_tuples) {
		r.cls = 0;
		r.unit = hi_unit;
	} else {
		r.cls = terminal ? 2u : 1u;
		uint32_t target = units_x * pool_waves / (n_exec ? n_exec : 1u);
		target = target < 16u ? 16u : target;
		uint32_t us = tuples / target + (tuples % target ? 1u : 0u);
		us = (us + gran - 1u) & ~(gran - 1u);
		// (never more than 65 536 tuples: the flat pipeline queues 16-bit positions inside the unit)
		r.unit = us < gran ? gran : (us > 65536u ? 65536u : us);
	}
	r.n_units = tuples / r.unit + (tuples % r.unit ? 1u : 0u);
}

// sum of the 8 arrival shards of a slot (full wave; the same value in every lane)
__device__ __forceinline__ unsigned long long polr_pool_arrived(ResidentSync *sync, uint32_t slot, uint32_t lane) {
	unsigned long long v = 0;
	if (lane < POLR_POOL_SHARDS) {
		v = __hip_atomic_load(&sync->arrived[slot][lane].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	for (int d = 4; d > 0; d >>= 1) {
		v += __shfl_xor(v, d, 64);
	}
	return __shfl(v, 0, 64);
}

// absorb the counter bank of a slot: lane = j * 8 + shard exchanges counter j of shard `shard` with 0 (k <= 8:
// one instruction for the whole bank).  Returns the sum of all k counters in lane 0.
__device__ __forceinline__ uint64_t polr_pool_absorb(DevMpx *m, DevMpx *mg, unsigned long long *bank, uint32_t k,
                                                     uint32_t lane, bool discard) {
	const uint32_t j = lane >> 3, shard = lane & 7u;
	unsigned long long v = 0;
	if (j < k) {
		v = atomicExch(&bank[(uint64_t)shard * POLR_KMAX + j], 0ull);
	}
	for (int d = 4; d > 0; d >>= 1) {
		v += __shfl_xor(v, d, 64);
	}
	// lanes j*8 now hold counter j
	uint64_t s = 0;
	const uint32_t last_path = (uint32_t)((volatile DevMpx *)m)->last_path;
	for (uint32_t jj = 0; jj < k; jj++) {
		const unsigned long long c = __shfl(v, jj * 8, 64);
		if (lane == 0 && !discard && c) {
			s += c;
			atomicAdd((unsigned long long *)&mg->stage_out[last_path][jj], c);
		}
	}
	return s;
}

// publish one round: deal its units over the rings (starting at ring `rot`), lane-parallel
__device__ __forceinline__ void polr_pool_publish(const PoolRun &run, PoolSync *sync, uint32_t exec, uint32_t slot,
                                                  const PoolRoundOut &r, uint32_t rot, uint32_t lane) {
	const uint32_t lo_cap = run.lo_cap, hi_cap = run.hi_cap;
	// ring (rot + u) % R gets unit u: lane r reserves the n_r tickets of ring r (one instruction for all rings)
	static_assert(POLR_POOL_RINGS == 64, "one lane per ring");
	const uint32_t R = run.n_rings;
	unsigned long long base = 0;
	if (lane < R) {
		const uint32_t first = (lane + R - (rot & (R - 1u))) & (R - 1u); // smallest u dealt to ring `lane`
		const uint32_t n_r = r.n_units > first ? (r.n_units - first + R - 1u) / R : 0u;
		if (n_r) {
			base = atomicAdd(r.cls == 0 ? &sync->ctl[lane].hi_tail
			                            : (r.cls == 1 ? &sync->ctl[lane].mid_tail : &sync->ctl[lane].lo_tail),
			                 (unsigned long long)n_r);
		}
	}
	for (uint32_t u0 = 0; u0 < r.n_units; u0 += 64) {
		const uint32_t u = u0 + lane;
		const uint32_t ring = (rot + u) & (R - 1u);
		const unsigned long long ring_base = __shfl(base, ring, 64);
		if (u < r.n_units) {
			const unsigned long long ticket = ring_base + u / R;
			const uint32_t cap = r.cls == 0 ? hi_cap : lo_cap;
			PoolEntry *e = (r.cls == 0 ? polr_pool_hi(sync, ring, lo_cap, hi_cap)
			                           : (r.cls == 1 ? polr_pool_mid(sync, ring, lo_cap) : polr_pool_lo(sync, ring, lo_cap))) +
			               (ticket & (cap - 1u));
			const uint32_t tag = polr_pool_tag(ticket, cap);
			const uint32_t ub = r.begin + u * r.unit;
			const uint32_t left = r.count - u * r.unit;
			const uint32_t uc = left < r.unit ? left : r.unit;
			__hip_atomic_store(&e->g1, polr_pool_g1(tag, exec, uc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&e->g0, polr_pool_g0(tag, POLR_POOL_KIND_WORK, slot, r.emit, r.path, ub), __ATOMIC_RELAXED,
			                   __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// one entry on ring `ring`'s hi queue, by one lane of a PROBE wave that shares its work (the hi queues are the ones idle
// waves look at first, and nobody blocks on them): the wave's own slot, path and executor, `level` halvings deep
__device__ __forceinline__ void polr_pool_publish_shared(PoolSync *sync, uint32_t ring, uint32_t lo_cap, uint32_t hi_cap,
                                                         uint32_t kind, const PoolUnit &of, uint32_t begin, uint32_t count,
                                                         uint32_t level) {
	const unsigned long long ticket = atomicAdd(&sync->ctl[ring].hi_tail, 1ull);
	PoolEntry *e = polr_pool_hi(sync, ring, lo_cap, hi_cap) + (ticket & (hi_cap - 1u));
	const uint32_t tag = polr_pool_tag(ticket, hi_cap);
	__hip_atomic_store(&e->g1, polr_pool_g1(tag, of.exec, count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_store(&e->g0, polr_pool_g0(tag, kind, of.slot, of.emit, of.path, begin, level), __ATOMIC_RELAXED,
	                   __HIP_MEMORY_SCOPE_AGENT);
}

// one EXIT entry per probe wave of every ring (the router that finishes last); run: the header in global memory
__device__ __forceinline__ void polr_pool_publish_exit(const PoolRun *run, const PoolRun &rh, PoolSync *sync, uint32_t lane) {
	const uint32_t lo_cap = rh.lo_cap;
	const uint32_t n_mine = lane < rh.n_rings ? run->worker_waves[lane] : 0u; // lane r: ring r
	unsigned long long base_mine = 0;
	if (n_mine) {
		base_mine = atomicAdd(&sync->ctl[lane].lo_tail, (unsigned long long)n_mine);
		// every probe wave leaves holding one mid ticket that no unit will be written for: the next run starts behind them
		(void)atomicAdd(&sync->ctl[lane].mid_tail, (unsigned long long)n_mine);
	}
	for (uint32_t ring = 0; ring < rh.n_rings; ring++) {
		const uint32_t n = __shfl(n_mine, ring, 64);
		const unsigned long long base = __shfl(base_mine, ring, 64);
		for (uint32_t i = lane; i < n; i += 64) {
			const unsigned long long ticket = base + i;
			PoolEntry *e = polr_pool_lo(sync, ring, lo_cap) + (ticket & (lo_cap - 1u));
			const uint32_t tag = polr_pool_tag(ticket, lo_cap);
			__hip_atomic_store(&e->g1, polr_pool_g1(tag, 0, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&e->g0, polr_pool_g0(tag, POLR_POOL_KIND_EXIT, 0, 0, 0, 0), __ATOMIC_RELAXED,
			                   __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// ---- worker side ---------------------------------------------------------------------------------
__device__ __forceinline__ bool polr_pool_decode(unsigned long long g0, unsigned long long g1, uint32_t tag, PoolUnit &u) {
	if ((uint32_t)(g0 >> 48) != tag || (uint32_t)(g1 >> 48) != tag) {
		return false;
	}
	u.kind = (uint32_t)(g0 >> 46) & 3u;
	u.slot = (uint32_t)(g0 >> 44) & 3u;
	u.emit = (uint32_t)(g0 >> 43) & 1u;
	u.path = (uint32_t)(g0 >> 38) & 31u;
	u.level = (uint32_t)(g0 >> 32) & 31u;
	u.begin = (uint32_t)g0;
	u.exec = (uint32_t)(g1 >> 32) & 0xFFFFu;
	u.count = (uint32_t)g1;
	return true;
}

// one lane: take a ticket of a non-blocking queue if it has one (compare-and-swap on its head while head < tail) and
// read its entry (being written by the router that reserved it)
__device__ __forceinline__ bool polr_pool_try_claim(POLR_GLOBAL unsigned long long *head, POLR_GLOBAL unsigned long long *tail,
                                                    POLR_GLOBAL PoolEntry *entries, uint32_t cap, uint32_t wave_in_ring,
                                                    uint32_t lottery, bool anybody, unsigned long long timeout_ticks,
                                                    unsigned long long &g0, unsigned long long &g1, uint32_t &tag) {
	const unsigned long long hh = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const unsigned long long ht = __hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	// the lottery keeps the compare-and-swap storm of a published unit small; `anybody` lifts it for a wave that has been
	// idle for a while: the waves a ticket is meant for may not be on the device (another launch is holding their
	// slots -- two full-size launches side by side), and a ticket nobody may take blocks every entry behind it
	if (hh >= ht || (!anybody && (((uint32_t)hh ^ wave_in_ring) & (lottery - 1u)) != 0)) {
		return false;
	}
	unsigned long long expect = hh;
	if (!__hip_atomic_compare_exchange_strong(head, &expect, hh + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
	                                          __HIP_MEMORY_SCOPE_AGENT)) {
		return false;
	}
	POLR_GLOBAL PoolEntry *e = entries + (hh & (cap - 1u));
	tag = polr_pool_tag(hh, cap);
	const unsigned long long t1 = wall_clock64();
	while (true) {
		g0 = __hip_atomic_load(&e->g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		g1 = __hip_atomic_load(&e->g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (((uint32_t)(g0 >> 48) == tag && (uint32_t)(g1 >> 48) == tag) || wall_clock64() - t1 > timeout_ticks) {
			break;
		}
		__builtin_amdgcn_s_sleep(1);
	}
	return true;
}

// What a probe wave knows about its ring between polls: the queue bases and the tickets it holds (a wave that holds a
// not-yet-written mid / lo ticket serves hi units meanwhile, so the tickets are kept across calls).
// (all accesses name the global address space: a probe wave polls between LDS-heavy units, see polr_device.h)
struct PoolPoller {
	POLR_GLOBAL PoolRun *run;
	POLR_GLOBAL PoolRingCtl *ctl;
	POLR_GLOBAL PoolEntry *hi_q, *mid_q, *lo_q;
	uint32_t lo_cap, hi_cap, wave_in_ring, lottery, idle_sleep;
	uint32_t idle_polls;                      // polls since this wave last had a unit
	unsigned long long lo_ticket, mid_ticket; // ~0ull: none
	unsigned long long timeout_ticks;
};

__device__ __forceinline__ void polr_pool_poller_init(PoolPoller &pp, PoolRun *run_generic, PoolSync *sync_generic,
                                                      uint32_t ring, uint32_t lo_cap, uint32_t hi_cap,
                                                      uint32_t wave_in_ring, uint32_t lottery, uint32_t idle_sleep,
                                                      unsigned long long timeout_ticks) {
	pp.run = as_global(run_generic);
	pp.ctl = as_global(&sync_generic->ctl[ring]);
	pp.hi_q = as_global(polr_pool_hi(sync_generic, ring, lo_cap, hi_cap));
	pp.mid_q = as_global(polr_pool_mid(sync_generic, ring, lo_cap));
	pp.lo_q = as_global(polr_pool_lo(sync_generic, ring, lo_cap));
	pp.lo_cap = lo_cap;
	pp.hi_cap = hi_cap;
	pp.wave_in_ring = wave_in_ring;
	pp.lottery = lottery;
	pp.idle_sleep = idle_sleep;
	pp.timeout_ticks = timeout_ticks;
	pp.lo_ticket = pp.mid_ticket = ~0ull;
	pp.idle_polls = 0;
}

// ONE look at the three queues (all lanes get the same answer): POLR_POLL_NONE, _WORK (u is the unit) or _LEAVE (an
// EXIT entry, or an entry that never arrived: the wave has to go)
#define POLR_POLL_NONE 0u
#define POLR_POLL_WORK 1u
#define POLR_POLL_LEAVE 2u
__device__ __forceinline__ uint32_t polr_pool_poll(PoolPoller &pp, PoolUnit &u, uint32_t lane) {
	unsigned long long g0 = 0, g1 = 0;
	uint32_t tag = 0, got = 0;
	if (lane == 0) {
		// (1) units somebody waits for, small rounds first: only while a queue is not empty, never blocking
		got = polr_pool_try_claim(&pp.ctl->hi_head, &pp.ctl->hi_tail, pp.hi_q, pp.hi_cap, pp.wave_in_ring, pp.lottery,
		                          pp.idle_polls >= POLR_POOL_LOTTERY_PATIENCE && (pp.idle_polls & 15u) == 15u, pp.timeout_ticks, g0, g1,
		                          tag)
		          ? 1u
		          : 0u;
		if (!got) {
			// (2) the mid queue: hold one ticket, look whether its entry has been written.  (Blocking tickets, like lo:
			// compare-and-swap claims of the hundreds of units of an exploit round were measured slower than FIFO
			// blocking behind them.  A mid unit waits at most for the unit its ticket holder is busy with.)
			if (pp.mid_ticket == ~0ull) {
				pp.mid_ticket = __hip_atomic_fetch_add(&pp.ctl->mid_head, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			POLR_GLOBAL PoolEntry *e = pp.mid_q + (pp.mid_ticket & (pp.lo_cap - 1u));
			tag = polr_pool_tag(pp.mid_ticket, pp.lo_cap);
			g0 = __hip_atomic_load(&e->g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			g1 = __hip_atomic_load(&e->g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((uint32_t)(g0 >> 48) == tag && (uint32_t)(g1 >> 48) == tag) {
				got = 3;
				pp.mid_ticket = ~0ull;
			}
		}
		if (!got) {
			// (3) the lo queue: hold one ticket, look whether its entry has been written
			if (pp.lo_ticket == ~0ull) {
				pp.lo_ticket = __hip_atomic_fetch_add(&pp.ctl->lo_head, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			POLR_GLOBAL PoolEntry *e = pp.lo_q + (pp.lo_ticket & (pp.lo_cap - 1u));
			tag = polr_pool_tag(pp.lo_ticket, pp.lo_cap);
			g0 = __hip_atomic_load(&e->g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			g1 = __hip_atomic_load(&e->g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((uint32_t)(g0 >> 48) == tag && (uint32_t)(g1 >> 48) == tag) {
				got = 2;
				pp.lo_ticket = ~0ull;
			}
		}
	}
	got = __builtin_amdgcn_readfirstlane(got);
	if (!got) {
		pp.idle_polls++;
		return POLR_POLL_NONE;
	}
	pp.idle_polls = 0;
	g0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(g0 >> 32)) << 32) |
	     __builtin_amdgcn_readfirstlane((uint32_t)g0);
	g1 = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(g1 >> 32)) << 32) |
	     __builtin_amdgcn_readfirstlane((uint32_t)g1);
	tag = __builtin_amdgcn_readfirstlane(tag);
	if (!polr_pool_decode(g0, g1, tag, u)) {
		return POLR_POLL_LEAVE; // (a hi entry that never arrived: watchdog)
	}
	return u.kind == POLR_POOL_KIND_EXIT ? POLR_POLL_LEAVE : POLR_POLL_WORK;
}

// Take the next unit for this wave, waiting for one.  Returns false when the wave has to leave (EXIT entry, abort or
// watchdog).
__device__ __forceinline__ bool polr_pool_next_unit(PoolPoller &pp, PoolUnit &u, uint32_t lane) {
	uint32_t spins = 0;
	while (true) {
		const uint32_t got = polr_pool_poll(pp, u, lane);
		if (got != POLR_POLL_NONE) {
			return got == POLR_POLL_WORK;
		}
		// nothing yet: back off (longer the longer nothing comes), give up when the run was given up
		spins++;
		if (spins < 8) {
			__builtin_amdgcn_s_sleep(2);
		} else if (spins < 32 || pp.idle_sleep < 64) {
			__builtin_amdgcn_s_sleep(16);
		} else {
			__builtin_amdgcn_s_sleep(64);
		}
		if ((spins & 15u) == 0) {
			// (no clock of its own: an idle wave may wait as long as the run takes; the routers' waits are bounded,
			// they raise `abort`, and the last of them always publishes the EXIT entries)
			uint32_t ab = 0;
			if (lane == 0) {
				ab = __hip_atomic_load(&pp.run->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (__builtin_amdgcn_readfirstlane(ab)) {
				return false;
			}
		}
	}
}

// ---- the router of one executor: ONE full wave, for the whole run ----------------------------------
// lds: POLR_RES_ROUTER_DWORDS dwords of state + scratch_lds: POLR_RES_HOT_DWORDS dwords (saved state of a rehearsal);
// cache_lds / cache_cap: window of the chunk-offset array in 8-byte entries.
__device__ __forceinline__ void polr_pool_router(const ResidentExec &x, PoolRun *run, const PoolRun &rh, uint32_t exec,
                                                 uint32_t k, uint32_t gran, uint32_t lane, uint32_t *lds,
                                                 uint64_t *cache_lds, uint32_t cache_cap, uint32_t *scratch_lds) {
	RT_DECL
	RT_T(rt_entry)
	DevMpx *mg = x.mpx;
	PoolSync *sync = rh.sync;
	const bool reset = (x.flags & POLR_RUN_RESET) != 0;
	const size_t bank_stride = (size_t)POLR_NSHARD * POLR_KMAX;
	if (reset) {
		// drop whatever bank 0 still holds (fire and forget; performed long before the first round is absorbed)
		const uint32_t j = lane >> 3;
		if (j < k) {
			(void)atomicExch(&x.counts[(uint64_t)(lane & 7u) * POLR_KMAX + j], 0ull);
		}
	}
	{
		const uint32_t *src = (const uint32_t *)mg;
		constexpr uint32_t kPer = (POLR_RES_HOT_DWORDS + 63) / 64;
		uint32_t v[kPer];
#pragma unroll
		for (uint32_t j = 0; j < kPer; j++) {
			const uint32_t i = j * 64 + lane;
			v[j] = i < POLR_RES_HOT_DWORDS ? src[i] : 0u;
		}
#pragma unroll
		for (uint32_t j = 0; j < kPer; j++) {
			const uint32_t i = j * 64 + lane;
			if (i < POLR_RES_HOT_DWORDS) {
				lds[i] = v[j];
			}
		}
	}
	DevMpx *m = (DevMpx *)lds;
	volatile uint32_t *const host_words = mg->progress;
	__builtin_amdgcn_wave_barrier();
	RT_T(rt_e1)
	RT_ADD(8, rt_e1 - rt_entry)
	if (lane == 0) {
		m->progress = nullptr; // nobody on the host follows the steps of a one-launch run
	}
	DevRound *round = (DevRound *)(lds + POLR_RES_HOT_DWORDS); // 24 bytes
	uint64_t *prefix = (uint64_t *)(lds + POLR_RES_HOT_DWORDS + 8);
	uint32_t *us = lds + POLR_RES_HOT_DWORDS + 12;
	OffsCache oc;
	oc.base = 0;
	oc.n = 0;
	oc.data = cache_lds;
	if (cache_cap > POLR_OFFS_CACHE) {
		cache_cap = POLR_OFFS_CACHE;
	}
	if (reset) {
		for (uint32_t i = lane; i < POLR_MAX_PATHS * POLR_MAX_JOINS; i += 64) {
			mg->stage_out[i / POLR_MAX_JOINS][i % POLR_MAX_JOINS] = 0;
		}
		{
			static_assert(sizeof(polr::MultiplexerCore) % 4 == 0, "cleared in dwords");
			uint32_t *cw = (uint32_t *)&m->core;
			for (uint32_t i = lane; i < sizeof(polr::MultiplexerCore) / 4; i += 64) {
				cw[i] = 0;
			}
		}
		__builtin_amdgcn_wave_barrier();
		if (lane == 0) {
			const polr_mpx_config cfg = m->cfg;
			m->core.InitAfterZero(cfg.routing, m->n_paths, cfg.regret_budget, cfg.init_tuple_count, cfg.atc_multiplier);
			m->num_intermediates_total = 0;
			m->num_rounds = 0;
			m->n_log = 0;
			m->last_path = 0;
		}
	}
	if (lane == 0) {
		m->chunk_idx = x.morsel_cursor ? 0 : x.chunk_begin;
		m->chunk_end = x.morsel_cursor ? 0 : x.chunk_end;
		m->chunk_offsets = x.chunk_offsets;
		m->n_chunks = x.n_chunks;
		m->n_tuples = x.n_tuples;
		m->done = x.chunk_begin >= x.chunk_end ? 1 : 0;
	}
	__builtin_amdgcn_wave_barrier();
	RT_T(rt_e2)
	RT_ADD(9, rt_e2 - rt_entry)
	if (x.chunk_offsets && !x.morsel_cursor && x.chunk_begin < x.chunk_end) {
		// the boundaries of the first chunks, in one cooperative load (the first routing step would otherwise fetch them
		// one dependent global load after the other on its single lane)
		polr_offs_cache_fill(oc, x, x.chunk_begin, cache_cap, lane);
	}
	unsigned long long target[POLR_SLOTS];
	if (((volatile DevMpx *)m)->res_valid) {
#pragma unroll
		for (uint32_t s = 0; s < POLR_SLOTS; s++) {
			target[s] = ((volatile DevMpx *)m)->res_target[s];
		}
	} else {
#pragma unroll
		for (uint32_t s = 0; s < POLR_SLOTS; s++) {
			target[s] = polr_pool_arrived(x.sync, s, lane);
			if (s) {
				polr_pool_absorb(m, mg, x.counts + s * bank_stride, k, lane, true);
			}
		}
	}
	// Rounds in flight, oldest first: the FRONT one is decided (the state has routed it); the ones behind it were
	// rehearsed ahead on a copy of the state (the shadow) and published already -- every real step that follows must
	// decide exactly them, in order.  Round i of the run sits in slot i % POLR_SLOTS, so the front's slot is
	// (n_pub - n_fly) % POLR_SLOTS.
	RT_T(rt_e3)
	RT_ADD(10, rt_e3 - rt_entry)
	uint32_t n_steps = 0;
	uint32_t range_i = 0; // further ranges of this executor taken so far (ResidentExec::more_begin / more_end)
	uint32_t n_pub = 0; // rounds published so far
	uint32_t n_fly = 0; // published, counters not absorbed yet
	PoolRoundOut ahead[POLR_SLOTS - 1] = {}; // the rehearsed rounds behind the front (statically indexed: registers)
	bool shadow_valid = false;               // scratch_lds holds the state as it will be after the last rehearsed round
	bool shadow_ended = false;               // the rehearsal ran off the end of the source: nothing more to publish ahead
	bool failed = false;
	// what this router's own watchdog was waiting for when it fired (reported by polr_mpx_finish)
	bool diag_timed = false;
	unsigned long long diag_want = 0, diag_got = 0;
	uint32_t diag_slot = 0, diag_fly = 0;
	uint32_t rot = exec; // units of consecutive rounds start on different rings
	// one routing step on whatever state sits in `lds` (the real one, or the shadow swapped in): the round it decides
	auto route_here = [&](PoolRoundOut &r) -> bool {
		// (asked for before the routing code runs, used after it: a sizing hint, one global round trip off the path)
		const uint32_t done_now = __hip_atomic_load(&run->routers_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (lane == 0) {
			polr_router_route(m, round, prefix, us, rh.pool_waves, &oc, false);
			if (x.path_plus1) {
				// BACKPRESSURE (src/parallel/pipeline.cpp:147-156, polar_config.cpp:128-147): this executor IS one
				// join order; its multiplexer routes DEFAULT_PATH, the order it stands for replaces path 0
				round->path = x.path_plus1 - 1u;
				m->last_path = x.path_plus1 - 1u;
			}
		}
		__builtin_amdgcn_wave_barrier();
		const bool done = ((volatile DevMpx *)m)->done != 0;
		const volatile DevRound *vr = round;
		r.begin = (uint32_t)vr->begin;
		r.count = (uint32_t)vr->count;
		r.path = vr->path;
		r.emit = vr->emit;
		// a round is cut for the executors that are still routing: the last ones get the whole pool
		// (routers_done counts executors that have finished ROUTING; their terminal rounds may still be queued, which
		// is what the lo queue is for)
		const uint32_t active = done_now < rh.n_exec ? rh.n_exec - done_now : 1u;
		const bool terminal = ((volatile DevMpx *)m)->core.num_cache_flushing_skips == polr::kIdxMax;
		polr_pool_size_units(r.count, rh.pool_waves, active, gran, rh.hi_tuples, rh.units_x, rh.hi_unit, terminal, r);
		return done;
	};
	auto publish = [&](const PoolRoundOut &r) {
		const uint32_t slot = n_pub & (POLR_SLOTS - 1u);
		polr_pool_publish(rh, sync, exec, slot, r, rot, lane);
		rot += r.n_units;
#pragma unroll
		for (uint32_t s = 0; s < POLR_SLOTS; s++) {
			if (s == slot) {
				target[s] += (unsigned long long)r.n_units * POLR_POOL_TOKENS;
			}
		}
		n_pub++;
		n_fly++;
	};
	auto swap_shadow = [&]() {
		__builtin_amdgcn_wave_barrier();
		for (uint32_t i = lane; i < POLR_RES_HOT_DWORDS; i += 64) {
			const uint32_t a = lds[i], b = scratch_lds[i];
			lds[i] = b;
			scratch_lds[i] = a;
		}
		__builtin_amdgcn_wave_barrier();
	};
	while (true) {
		__builtin_amdgcn_wave_barrier();
		const uint32_t front_slot = (n_pub - n_fly) & (POLR_SLOTS - 1u);
		RT_T(rt0)
		// (1) the oldest round in flight has to be complete before its counters can be absorbed
		if (n_fly) {
			unsigned long long want = 0;
#pragma unroll
			for (uint32_t s = 0; s < POLR_SLOTS; s++) {
				want = s == front_slot ? target[s] : want;
			}
			const unsigned long long t0 = wall_clock64();
			uint32_t spins = 0;
			while (polr_pool_arrived(x.sync, front_slot, lane) != want) {
				__builtin_amdgcn_s_sleep(1);
				if ((++spins & 7u) == 0) { // (every spin is a round trip to the arrival counters: a look every ~10 us)
					uint32_t ab = __hip_atomic_load(&run->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (ab || wall_clock64() - t0 > rh.timeout_ticks) {
						failed = true; // a probe wave is missing
						if (!ab) { // (this router's own watchdog: say what it was waiting for, polr_mpx_finish reports it)
							diag_want = want;
							diag_slot = front_slot;
							diag_fly = n_fly;
							diag_timed = true;
						}
						break;
					}
				}
			}
			if (failed) {
				if (diag_timed) {
					diag_got = polr_pool_arrived(x.sync, front_slot, lane);
				}
				break;
			}
		}
		RT_T(rt1)
		RT_ADD(0, rt1 - rt0)
		// (2) the real step
		if (!(reset && n_steps == 0)) { // (a reset run starts on the bank it dropped at entry)
			const uint64_t got = polr_pool_absorb(m, mg, x.counts + (n_fly ? front_slot : 0u) * bank_stride, k, lane, false);
			if (lane == 0) {
				m->core.AddNumIntermediates(got);
				m->num_intermediates_total += got;
			}
		}
		if (!x.morsel_cursor && range_i < x.n_more) {
			// this executor's current range is used up and it owns another one: go on there, same multiplexer state
			const uint64_t ci = ((volatile DevMpx *)m)->chunk_idx, ce = ((volatile DevMpx *)m)->chunk_end;
			if (ci >= ce) {
				uint64_t nb = 0, ne = 0;
#pragma unroll
				for (uint32_t j = 0; j < POLR_MORE_RANGES; j++) {
					if (j == range_i) {
						nb = x.more_begin[j];
						ne = x.more_end[j];
					}
				}
				if (lane == 0) {
					m->chunk_idx = nb;
					m->chunk_end = ne;
					m->done = nb >= ne ? 1 : 0;
				}
				range_i++;
				__builtin_amdgcn_wave_barrier();
			}
		}
		if (lane == 0 && x.morsel_cursor && m->chunk_idx >= m->chunk_end) {
			// this executor's morsel is used up (or it has none yet): pull the next one
			const unsigned long long next = atomicAdd(x.morsel_cursor, (unsigned long long)x.morsel_chunks);
			if (next < x.morsel_end) {
				m->chunk_idx = next;
				m->chunk_end = next + x.morsel_chunks < x.morsel_end ? next + x.morsel_chunks : x.morsel_end;
				m->done = 0;
			}
		}
		n_steps++;
		__builtin_amdgcn_wave_barrier();
		RT_T(rt2)
		RT_ADD(1, rt2 - rt1)
		RT_ADD(5, 1)
		{
			PoolRoundOut r;
			const bool done = route_here(r);
			RT_T(rt3)
			RT_ADD(2, rt3 - rt2)
			if (n_fly > 1) {
				// this round is already out: the real decision must be the rehearsed one, bit for bit
				const PoolRoundOut &sp = ahead[0];
				if (done || r.begin != sp.begin || r.count != sp.count || r.path != sp.path || r.emit != sp.emit) {
					failed = true; // (cannot happen while polr_can_speculate is right; never continue on a wrong round)
					break;
				}
#pragma unroll
				for (uint32_t i = 0; i + 1 < POLR_SLOTS - 1; i++) {
					ahead[i] = ahead[i + 1];
				}
				n_fly--;
			} else {
				n_fly = 0;
				if (done) {
					break;
				}
				publish(r);
				RT_T(rt4)
				RT_ADD(3, rt4 - rt3)
				if (n_steps == 1) {
					RT_ADD(6, rt4 - rt_entry)
					RT_ADD(7, 1)
				}
				shadow_valid = false;
				shadow_ended = false;
			}
		}
		RT_T(rt5)
		// (3) rehearse ahead: while the decision after the last published round cannot depend on intermediates that
		// are still outstanding, decide it on the shadow and publish it in the next slot
		while (n_fly < POLR_SLOTS && !shadow_ended) {
			if (!shadow_valid) {
				if (!polr_can_speculate(((volatile DevMpx *)m)->core)) {
					break;
				}
				__builtin_amdgcn_wave_barrier();
				for (uint32_t i = lane; i < POLR_RES_HOT_DWORDS; i += 64) {
					scratch_lds[i] = lds[i];
				}
				__builtin_amdgcn_wave_barrier();
				if (lane == 0) {
					((DevMpx *)scratch_lds)->log_enabled = 0; // (no trace of a rehearsal)
				}
				shadow_valid = true;
			} else if (!polr_can_speculate(((volatile DevMpx *)scratch_lds)->core)) {
				break;
			}
			// the rehearsal runs IN PLACE (the routing code only ever sees the one LDS object, which keeps its accesses
			// LDS instructions): the shadow is swapped in, routed, and swapped out again
			swap_shadow();
			PoolRoundOut r2;
			const bool done2 = route_here(r2);
			swap_shadow();
			if (done2) {
				shadow_ended = true; // (a rehearsal that runs off the end of the source publishes nothing)
				break;
			}
#pragma unroll
			for (uint32_t i = 0; i < POLR_SLOTS - 1; i++) {
				if (i + 1 == n_fly) {
					ahead[i] = r2;
				}
			}
			publish(r2);
		}
		RT_T(rt6)
		RT_ADD(4, rt6 - rt5)
		// while the pool probes: keep the boundaries of the chunks ahead in LDS
		{
			const uint64_t ci = ((volatile DevMpx *)m)->chunk_idx;
			if (x.chunk_offsets && ci - oc.base >= oc.n / 2) { // (also the first fill: n == 0)
				polr_offs_cache_fill(oc, x, ci, cache_cap, lane);
			}
		}
	}
	if (failed) {
		if (lane == 0) {
			__hip_atomic_store(&run->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (host_words) {
				host_words[2] = 1;
			}
			if (diag_timed && host_words) { // (the pinned words of THIS executor's multiplexer)
				host_words[5] = exec;
				host_words[6] = diag_slot;
				host_words[7] = diag_fly;
				host_words[8] = (uint32_t)diag_want;
				host_words[9] = (uint32_t)(diag_want >> 32);
				host_words[10] = (uint32_t)diag_got;
				host_words[11] = (uint32_t)(diag_got >> 32);
				host_words[4] = 1u;
			}
		}
	}
	__builtin_amdgcn_wave_barrier();
	if (x.flags & POLR_RUN_FINISH) {
		if (lane == 0) {
			polr_close_run(m);
		}
		__builtin_amdgcn_wave_barrier();
		polr_write_stats(m, mg, x.stats_out, lane);
	}
	__builtin_amdgcn_wave_barrier();
	if (lane == 0) {
		m->res_valid = failed ? 0u : 1u;
#pragma unroll
		for (uint32_t s = 0; s < POLR_SLOTS; s++) {
			m->res_target[s] = target[s];
		}
		m->progress = host_words;
		if (host_words) { // what a per-round run would have published: the run is over
			host_words[1] = m->done;
			host_words[0] = m->steps_done;
		}
	}
	__builtin_amdgcn_wave_barrier();
	{
		uint32_t *dst = (uint32_t *)mg;
		for (uint32_t i = lane; i < POLR_RES_HOT_DWORDS; i += 64) {
			dst[i] = lds[i];
		}
	}
	RT_FLUSH
	// the router that finishes last lets the pool go
	uint32_t last = 0;
	if (lane == 0) {
		const uint32_t before = atomicAdd(&run->routers_done, 1u);
		last = before + 1u == rh.n_exec ? 1u : 0u;
	}
	if (__builtin_amdgcn_readfirstlane(last)) {
		polr_pool_publish_exit(run, rh, sync, lane);
	}
}
