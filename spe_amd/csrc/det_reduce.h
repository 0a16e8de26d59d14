// Deterministic cross-workgroup reductions of the gradient path (bias / LayerNorm / LayerScale column sums, loss sums): no fp32
// atomics.  Every workgroup that contributes to a reduction ("member" m of n of reduction "set" s) writes its partial vector to
// its own row of a slab in the reduction workspace and takes a ticket (integer atomic - exact, order-free); the member
// that draws the last ticket of its group of 16 adds that group's rows in index order, and the group that finishes last adds
// the group sums in index order and stores the result.  The partials are fixed by the grid decomposition and both orders are
// fixed, so the sum is bitwise reproducible run to run whatever order the workgroups execute in (the fp32 atomics this
// replaces added in arrival order).  The tail costs the last workgroup 16 + n/16 rows instead of n.
//
// Workspace: caller-owned device memory registered once with spe_set_reduce_workspace (include/spe_hip.h): tickets first
// (zero between launches: the last arriver resets what it used), slabs after.  Launches that use it must be stream-ordered
// (the product launches everything on the current stream); a launcher whose reduction does not fit returns -4.
#pragma once
#include "common.h"

struct DetWs {
    float* slab;            // slab_floats floats
    unsigned* tickets;      // ntickets counters, zero when no launch is in flight
    long slab_floats;
    int ntickets;
    float* defer;           // != NULL: DEFERRED reduction - every member only leaves its partial at defer[(set * nmembers + member) * L + c]
                            // (plain stores); the sums are taken later, for many launches at once, by spe_reduce_flush (misc.hip)
};
#define DET_G 16
// host side (misc.hip): the registered workspace, or {nullptr, ...}
DetWs spe_detws();
// ---- deferred reductions (round 4).  The tree of det_reduce costs the LAST workgroup of a launch six dependent round trips to the
// memory-side coherence point: 5-8 us at the end of every LayerNorm / LayerScale backward, conversion with column sums and GEMM epilogue
// with a bias gradient - ~250 launches, ~1.4 ms per step (tools/debug/rowops_time.py).  The destinations of those sums are PARAMETER
// gradients inside the all-reduce buckets: nothing reads them before the bucket is reduced / the optimiser runs.  The owner of those
// buckets registers their address ranges (spe_reduce_defer_ranges) and an arena; a launcher whose destinations ALL lie inside the ranges
// hands its kernel a region of the arena for the partial rows (plain stores, no tickets) and records where the totals belong;
// spe_reduce_flush sums the rows of ALL recorded launches in one kernel - members in index order, so the result is as reproducible as
// the tree's - ordered after the producers by the stream.  Destinations outside the ranges (loss sums, temporaries) keep the tree.
#define DET_DEFER_MAXSEG 40
struct DetDeferSeg { float* dst; int len; };
// host side (misc.hip).  det_defer_try: region for sets * members rows of L floats when deferral applies to these destinations and
// everything fits (flushing first if the table or the arena is full), else nullptr; det_defer_commit: totals o = set * L + c,
// o < sum(len), go to the segments in order (dst += or =).
float* det_defer_try(long sets, long members, long L, int nseg, const DetDeferSeg* segs, hipStream_t st);
void det_defer_commit(float* region, long sets, long members, long L, int nseg, const DetDeferSeg* segs, int accumulate);

// floats / tickets one launch needs: `sets` reductions of `members` partial vectors of L values each
static inline long det_slab_floats(long sets, long members, long L) { return members <= 1 ? 0 : sets * (members + (members + DET_G - 1) / DET_G) * L; }
static inline long det_tickets(long sets, long members) { return members <= 1 ? 0 : sets * ((members + DET_G - 1) / DET_G + 1); }
#define DET_CHECK(ws, sets, members, L) \
    do { if ((members) > 1 && (!(ws).slab || det_slab_floats(sets, members, L) > (ws).slab_floats || det_tickets(sets, members) > (ws).ntickets)) return -4; } while (0)

// All threads of the workgroup call this (tid / nthreads: linear thread index / count).  partial(c) -> this workgroup's partial
// of value c in [0, L) (any thread may ask for any c: LDS or registers behind it must be published by the caller's barrier);
// store(c, total) writes the finished sum (exactly one thread of one workgroup calls it per c).  Workgroup-uniform control flow.
template <class P, class S>
__device__ __forceinline__ void det_reduce(const DetWs& ws, int set, int member, int nmembers, int L, int tid, int nthreads, P partial, S store) {
    if (nmembers <= 1) {
        for (int c = tid; c < L; c += nthreads) store(c, partial(c));
        return;
    }
    if (ws.defer) {         // deferred: the partial row only (see above)
        float* row = ws.defer + ((long)set * nmembers + member) * L;
        for (int c = tid; c < L; c += nthreads) row[c] = partial(c);
        return;
    }
#if defined(SPE_ABLATE) && defined(SPE_DBG_NORED)
    // timing experiment (tools/debug/rowops_time.py): no cross-workgroup sum.  Measured, round 4: the tail below costs 5-8 us per launch
    // (layernorm_bwd 19.8 -> 12.1 us, lsres_bwd16 15.2 -> 10.3, conversion + column sums 11.1 -> 5.5), ~1.4 ms per step over ~250
    // launches: six dependent round trips to the memory-side coherence point for the last workgroup.  A variant that ADDS 64-bit fixed-point
    // partials with returning atomics (exact, order-free: 3 round trips) was slower - same-address contention, 15-19 us for the conversion
    // even with 16 accumulator copies - and is gone; the sums into the gradient buckets are deferred to one flush kernel instead (above).
    if (member == 0) for (int c = tid; c < L; c += nthreads) store(c, partial(c));
    return;
#endif
    __shared__ unsigned det_ticket;
    const int ngroups = (nmembers + DET_G - 1) / DET_G, group = member / DET_G, g0 = group * DET_G;
    const int gsize = min(DET_G, nmembers - g0);
    float* base = ws.slab + (long)set * (nmembers + ngroups) * L;
    unsigned* tk = ws.tickets + (long)set * (ngroups + 1);
    // Slab traffic is made of RETURNING device-scope atomics (swap to write, swap-with-0 to read): they execute at the memory-side
    // coherence point shared by the XCDs, and a thread that has its return values back knows its writes have landed there - the
    // workgroup barrier (which waits for them) therefore orders the partials before the ticket.  Plain accesses would need a
    // device-scope release fence per workgroup, which writes back the whole dirty L2 of the XCD - these kernels have just written
    // megabytes of dx - and cost 60-100 us per launch; write-through (sc1) stores are acknowledged before they reach memory and
    // lost the race against the ticket about once in 10^3 launches.
    unsigned sink = 0u;
    auto put = [&](float* q, float v) {
        sink |= __hip_atomic_exchange(reinterpret_cast<unsigned*>(q), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto get = [](float* q) {          // swap with 0, not or-with-0: the compiler turns an idempotent read-modify-write into a plain atomic load
        return __uint_as_float(__hip_atomic_exchange(reinterpret_cast<unsigned*>(q), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    for (int c = tid; c < L; c += nthreads) put(base + (long)member * L + c, partial(c));
    asm volatile("" :: "v"(sink));          // the swaps return: completion, not just issue
    __syncthreads();
    if (tid == 0) det_ticket = __hip_atomic_fetch_add(tk + 1 + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (det_ticket != (unsigned)(gsize - 1)) return;
    float* gp = base + (long)(nmembers + group) * L;
    for (int c = tid; c < L; c += nthreads) {
        float v[DET_G];                    // all loads in flight together, then the sum in index order
#pragma unroll
        for (int m = 0; m < DET_G; ++m) v[m] = m < gsize ? get(base + (long)(g0 + m) * L + c) : 0.f;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < DET_G; ++m) if (m < gsize) s += v[m];
        if (ngroups == 1) store(c, s); else put(gp + c, s);
    }
    if (tid == 0) __hip_atomic_store(tk + 1 + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
    if (ngroups == 1) return;
    asm volatile("" :: "v"(sink));
    __syncthreads();
    if (tid == 0) det_ticket = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (det_ticket != (unsigned)(ngroups - 1)) return;
    for (int c = tid; c < L; c += nthreads) {
        float s = 0.f;
        for (int gb = 0; gb < ngroups; gb += DET_G) {
            float v[DET_G];
#pragma unroll
            for (int g = 0; g < DET_G; ++g) v[g] = gb + g < ngroups ? get(base + (long)(nmembers + gb + g) * L + c) : 0.f;
#pragma unroll
            for (int g = 0; g < DET_G; ++g) if (gb + g < ngroups) s += v[g];
        }
        store(c, s);
    }
    if (tid == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
