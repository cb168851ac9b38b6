/*
 * hb_ampc.h - a GPU-resident shard of the AMPC harmonic-centrality counter table.
 *
 * The reference's distributed variant keeps `counters: DefaultDhtTable<NodeID, HyperLogLog<64>>`
 * (crates/core/src/entrypoint/ampc/harmonic_centrality/mod.rs:47-53) in a raft-replicated key-value store and
 * drives it with three batch operations per shard (mapper.rs:52-118):
 *     batch_set     setup_counters                                   dht/store.rs (insert / overwrite)
 *     batch_get     get_old_counters (edge.from of a batch)
 *     batch_upsert  update_counters with `HyperLogLog64Upsert`       dht/upsert.rs:43-54,67-89, dht/store.rs:159-190
 * batch_upsert applies the pairs IN ORDER: an absent key is inserted (`Inserted`); otherwise old.merge(&new)
 * (register-wise max, hyperloglog.rs:4531-4535) and the action is `Merged` iff the stored value changed, else
 * `NoChange`.  This header is the C ABI a GPU worker would put behind those three calls: the counters of a shard
 * live in HBM as one 64-byte block each, the merge / changed detection runs in a HIP kernel (one quad per key,
 * the key's pairs applied in batch order), the key -> slot index is a device hash table (hb_table.hip.h) since round 5.
 * Raft replication, the network protocol and the other upsert operators are out of scope.
 *
 * extern "C", never unwinds, 0 = ok, negative = HB_ERR_* of hyperball.h; needs a gfx950 device (no CPU fallback).
 */
#ifndef HB_AMPC_H
#define HB_AMPC_H

#include <stdint.h>

#include "hyperball.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hbu_table hbu_table;

#define HBU_NO_CHANGE 0 /* UpsertAction::NoChange  (dht/upsert.rs:24-28) */
#define HBU_MERGED    1 /* UpsertAction::Merged                          */
#define HBU_INSERTED  2 /* UpsertAction::Inserted                        */

/* device < 0: current device.  capacity_hint: expected number of keys (the table grows as needed). */
int hbu_create(int32_t device, uint64_t capacity_hint, hbu_table **out);
void hbu_destroy(hbu_table *t);
const char *hbu_last_error(const hbu_table *t);
int hbu_len(const hbu_table *t, uint64_t *keys);

/* counters: count x 64 bytes (HyperLogLog<64>::registers).  Later pairs of the same key win. */
int hbu_batch_set(hbu_table *t, const hb_u128 *keys, const uint8_t *counters, uint64_t count);
/* found[i] = 0 for an absent key (counters_out[i] is then all zero = HyperLogLog::default()). */
int hbu_batch_get(hbu_table *t, const hb_u128 *keys, uint64_t count, uint8_t *counters_out, uint8_t *found);
/* HyperLogLog64Upsert over the pairs in order; actions[i] = HBU_* for pair i. */
int hbu_batch_upsert(hbu_table *t, const hb_u128 *keys, const uint8_t *counters, uint64_t count, uint8_t *actions);

#ifdef __cplusplus
}
#endif
#endif /* HB_AMPC_H */
