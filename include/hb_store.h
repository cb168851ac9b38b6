/* hb_store.h - native writer of the centrality stores that `store_harmonic` leaves behind.
 *
 * Reference: crates/core/src/webgraph/centrality/mod.rs:72-114 writes two `speedy_kv` databases next to each other,
 *     <output>/harmonic        Db<NodeID, f64>   NodeID -> centrality
 *     <output>/harmonic_rank   Db<NodeID, u64>   NodeID -> rank in (Reverse(total_cmp centrality), NodeID) order
 * each committed and then merged into ONE segment (`commit` + `merge_all_segments`).  What is on disk afterwards, per
 * database (crates/speedy-kv/src):
 *     meta.json              {"segments": ["<uuid>"]}, serde_json pretty form               lib.rs:227-231,292-297
 *     <uuid>.blobs           key bytes, value bytes, key bytes, ... in ascending key-byte order blob_store.rs:100-118
 *     <uuid>.bid             one 32-byte BlobPointer {key.start, key.end, value.start, value.end} (u64 LE) per entry,
 *                            entry i = BlobId i                                                 lib.rs:46-92, blob_index.rs:75-78
 *     <uuid>.ids             fst::Map  key bytes -> BlobId                                      blob_id_index.rs:137-174
 *     <uuid>.blm             bincode of bloom::BytesBloomFilter (bits = f(#entries, 0.01))      segment.rs:56-59,81-84
 * Keys and values are bincode `standard()` encodings (serialized.rs:86-92, crates/common/src/lib.rs:1-3): NodeID = u128 in
 * bincode's variable-length integer form, f64 = 8 bytes little endian, u64 = variable-length integer.
 *
 * FORMAT UNPINNED.  Three of the formats live in crates that are NOT vendored under /root/reference and were restated from
 * their published formats: `fst` 0.4.7 (the map file: version 3, any valid node encoding is readable; this writer emits a
 * prefix tree WITHOUT suffix sharing - for keys that are hashes only the last one or two bytes of a key could share, so the
 * file is somewhat larger than the reference's but answers the same lookups), `bitvec`
 * 1.0.1's serde form of BitVec<usize, Lsb0> inside bincode 2.0.0-rc.3, and xxh3-128 with the secret derived from seed 42
 * (third_party/xxhash, exact).  No store written by the reference exists in this image and no Rust toolchain to read one
 * back; tests/speedy_kv_reader.py is an independent Python reader written against the same descriptions.  Keeping the Rust
 * writer remains the supported route; this is the native alternative asked for in VERDICT r2 #8.
 *
 * hb_store_write / hb_store_harmonic are host-only: no device is touched.  Thread-compatible (no shared state). */
#ifndef HB_STORE_H
#define HB_STORE_H

#include <stddef.h>
#include <stdint.h>

#include "hyperball.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HB_STORE_F64 0 /* Db<NodeID, f64>: values = const double *   */
#define HB_STORE_U64 1 /* Db<NodeID, u64>: values = const uint64_t * */

/* Write ONE speedy_kv database directory `dir` (created if absent) holding `count` entries ids[i] -> values[i] as a single
 * segment.  This writer CREATES databases: a directory whose meta.json already lists segments is refused (HB_ERR_INVALID) -
 * Db::open_or_create would have added a segment to it, and replacing the meta would orphan the old files; an empty
 * `{"segments": []}` is filled.  meta.json is written last, through a temporary file + rename: a database whose meta.json
 * exists is complete, and a meta.json that cannot be written fails the call (HB_ERR_IO).  ids need not be sorted and must
 * be distinct (a duplicate id is an error: Db::insert would have overwritten, which a caller of this function cannot mean).
 * All host cores are used (the cgroup's CPU quota; HB_HOST_THREADS overrides): 11-13 M entries/s per database on the MI355X
 * boxes' 16 CPUs.  Returns HB_OK, or HB_ERR_INVALID / HB_ERR_NOMEM / HB_ERR_IO with a message in err (if err_len > 0). */
int hb_store_write(const char *dir, const hb_u128 *ids, const void *values, int value_kind, uint64_t count, char *err, size_t err_len);

/* store_harmonic (centrality/mod.rs:72-114): `<output>/harmonic` from (ids, centralities) and `<output>/harmonic_rank`
 * from (ids, ranks) - the arrays hb_result_copy and hb_result_ranks return. */
int hb_store_harmonic(const char *output, const hb_u128 *ids, const double *centralities, const uint64_t *ranks, uint64_t count,
                      char *err, size_t err_len);

/* [ABI 5] The same two databases straight from the context that holds the results of hb_run: the (NodeID, f64) list and the ranks
 * are taken as hb_result_copy / hb_result_ranks return them, and the key order of both databases - ascending bincode key BYTES,
 * which is not ascending NodeID order (little-endian integers behind a class byte) - comes from one 136-bit radix sort on the
 * device instead of a comparison sort of 79 M 24-byte records on the host (C4: 1.2 s -> 0.1 s of the 6.7 s emission).  Same
 * files, byte for byte, as hb_store_harmonic on the arrays (tests/test_gpu.py).  Needs the context's device. */
int hb_store_harmonic_results(hb_ctx *ctx, const char *output, char *err, uint64_t err_len);

#ifdef __cplusplus
}
#endif
#endif
