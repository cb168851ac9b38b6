// hb_table.hip.h - a device hash table NodeID (u128) -> 32-bit id, filled by the kernels that use it.
// Shared by the GPU ingest (hb_ingest.hip: endpoint -> provisional id as the record batches arrive) and the AMPC counter
// shard (hb_ampc.hip: key -> counter slot).  Device code only; gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/rocprim.hpp>

// every vector-memory operation this wave has issued is complete (gfx950: s_waitcnt vmcnt(0); a compiler barrier as well)
#ifndef HB_DRAIN_VMEM
#define HB_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

namespace hbt {
using u128 = rocprim::uint128_t;

// ---- the table: NodeID (u128) -> id, open addressing, linear probing ------------------------------------------------------
// slot state lives in the pid array: kEmpty, kBusy (claimed, key not yet published), else the pid.  An insert claims an
// empty slot with ONE compare-and-swap, writes the key, then publishes the pid; a reader that sees a pid may read the key.
// Nothing ever spins INSIDE an iteration: a lane that finds kBusy simply goes round the loop again, by which time the
// claiming lane - which runs straight-line code to the publishing store inside the same iteration - is done even if it sits
// in the same wave (the loop's only exit is wave-uniform, see table_get).
//
// Visibility across the 8 XCDs (their L2s are not coherent with each other, /opt/skills/guides/MI355X_MICROARCH.md
// "inter-workgroup visibility"): EVERY access to a slot - id word and both key halves, readers and writers - is an 8-byte-or-
// smaller agent-scope atomic (`sc1`: write-through stores, L1-bypassing loads), the guide's valid form "8-B agent atomics on
// both sides"; the writer drains its key stores (`s_waitcnt vmcnt(0)`, as inline asm: the compiler may not drop it) before
// the id store, a reader loads the key only after it has seen the id.  No release / acquire FENCE anywhere: the first form of
// this kernel published with a release store (`buffer_wbl2`: a write-back of the XCD's whole L2, dirty with the kernel's own
// output stream, per new id) and read with acquire loads (`buffer_inv` per probe) - at C4 the table kernel, not the host
// link, set the pace of hb_append_edges (profiles/r04j_ingest_C4_trace.txt), and a returned atomic right in front of the
// release is exactly the pattern the guide's "compiler hazard" warns about.
constexpr uint32_t kEmpty = 0xFFFFFFFFu, kBusy = 0xFFFFFFFEu;
constexpr uint64_t kMaxPids = 1ull << 31; // far above the engine's own limit (n_pad < 2^30)

__device__ __forceinline__ uint64_t slot_hash(u128 key)
{
    // both halves through a full-avalanche mixer: ids may be hashes (the reference's xxh3-128) or small consecutive
    // integers (tests) - neither may cluster
    uint64_t x = (uint64_t)key ^ (((uint64_t)(key >> 64)) * 0x9E3779B97F4A7C15ull);
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return x;
}

struct Table {
    u128 *keys;
    uint32_t *pids;
    uint64_t mask; // slots - 1
    unsigned long long *counter; // pids handed out so far
};

// pid of `key`, inserted with the next free pid if absent (want_pid == kEmpty) or with want_pid (rehash).
// Progress does not depend on where the compiler places a block: the loop has ONE exit and its condition is
// wave-uniform (a ballot over the lanes still searching), so no lane leaves early and there is no exit block the publishing
// stores could be sunk into - a lane that claimed a slot publishes it inside the iteration, in straight-line code in front of the
// ballot every lane of the wave takes part in, and a lane of the same wave that saw kBusy finds the pid on its next turn.
// (The earlier form returned from inside the loop; that it worked rested on the publishing block staying in the loop body.)
__device__ __forceinline__ uint32_t table_get(const Table &t, u128 key, uint32_t want_pid)
{
    const unsigned long long klo = (unsigned long long)key, khi = (unsigned long long)(key >> 64);
    unsigned long long *halves = (unsigned long long *)t.keys; // slot s: halves[2 s] = low, halves[2 s + 1] = high 64 bits
    uint64_t slot = slot_hash(key) & t.mask;
    uint32_t result = kEmpty;
    bool searching = true;
    do {
        if (searching) {
            const uint32_t s = __hip_atomic_load(&t.pids[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s == kEmpty) {
                // (a lost race: somebody else claimed it first - read the slot again on the next turn)
                if (atomicCAS(&t.pids[slot], kEmpty, kBusy) == kEmpty) {
                    __hip_atomic_store(&halves[2 * slot], klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&halves[2 * slot + 1], khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t pid = want_pid != kEmpty ? want_pid : (uint32_t)atomicAdd(t.counter, 1ull);
                    HB_DRAIN_VMEM(); // the key has left this CU before the id does
                    __hip_atomic_store(&t.pids[slot], pid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    result = pid;
                    searching = false;
                }
            } else if (s != kBusy) { // (kBusy = being published: the same slot again on the next turn)
                const unsigned long long lo = __hip_atomic_load(&halves[2 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load(&halves[2 * slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lo == klo && hi == khi) {
                    result = s;
                    searching = false;
                } else {
                    slot = (slot + 1) & t.mask;
                }
            }
        }
    } while (__ballot(searching) != 0ull);
    return result;
}

// id of `key` if the table holds it, else kEmpty; never inserts.  Only for kernels that run while NO insert is in flight
// (no kBusy slots: a launch of its own on the table's stream).
__device__ __forceinline__ uint32_t table_find(const Table &t, u128 key)
{
    const unsigned long long klo = (unsigned long long)key, khi = (unsigned long long)(key >> 64);
    const unsigned long long *halves = (const unsigned long long *)t.keys;
    uint64_t slot = slot_hash(key) & t.mask;
    for (;;) {
        const uint32_t s = t.pids[slot];
        if (s >= kBusy) return kEmpty;
        if (halves[2 * slot] == klo && halves[2 * slot + 1] == khi) return s;
        slot = (slot + 1) & t.mask;
    }
}
} // namespace hbt
