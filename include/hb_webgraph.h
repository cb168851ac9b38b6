/*
 * hb_webgraph.h - native reader of Stract's on-disk webgraph edge store.
 *
 * Replaces, for the harmonic-centrality path only, the per-pass streaming through Rust iterators
 *     Webgraph::host_edges() -> EdgeStore::iter_hosts_small()      crates/core/src/webgraph/mod.rs:192, store.rs:297-314
 *     Webgraph::host_nodes() -> EdgeStore::iter_host_node_ids()    webgraph/mod.rs:157, store.rs:338-357
 * which read three columns of every document of every segment (SmallSegmentEdgesIter, store.rs:360-417):
 * `from_host_id` (u128), `to_host_id` (u128), `rel_flags` (u64)  (webgraph/schema.rs:182-260).
 *
 * On disk (`<webgraph>/edges/`, store.rs:60-74; a tantivy-fork index):
 *   meta.json                      {"segments":[{"segment_id":"<uuid>","max_doc":N,...}, ...], ...}
 *                                   (tantivy/src/index/index_meta.rs:215-225,325-342); segments are iterated in this
 *                                   order, documents in ascending doc id: that IS the stream order the reference's
 *                                   first-occurrence rule (store.rs:313) is defined on
 *   <32 hex uuid>.col              one columnar file per segment (index_meta.rs:134-146):
 *     [columnar body][JSON footer {"version":..,"crc":..}][footer_len u32][1337 u32]     directory/footer.rs:16,33-41,45-102
 *     columnar body = [column data][sstable dictionary][sstable_len u64][num_rows u32][version u32 = 1][02 71 77 42]
 *                                   columnar/columnar/reader/mod.rs:85-103, writer/serializer.rs:58-69, format_version.rs:6-17
 *     dictionary key = column name, 0x00, column type code (U64 = 1, U128 = 6)            writer/serializer.rs:20-31, column_type.rs:13-21
 *     dictionary value = byte range of the column inside [column data]                     sstable/value/range.rs
 *     sstable = blocks [len+1 u32][0 raw | 1 zstd][values block][keys block] ..., [0 u32], optional index,
 *               [fst_len u64][index_offset u64][num_terms u64][version u32 = 3]            sstable/delta.rs:45-88, mod.rs:293-316, dictionary.rs:183-227
 *     column   = [cardinality u8 = 0 (Full)][codec u8][num_rows u32][min][max][num_rows raw little-endian values]
 *                [column_index_num_bytes u32]; codec Raw only: 0 for u128 columns, 3 for u64 columns
 *                                   column/serialize.rs:17-59, column_index/serialize.rs:21-36, u128_based/raw.rs:33-57,94-113,
 *                                   u64_based/raw.rs, u64_based/mod.rs:27-32 (the fork writes nothing else: serialize.rs:23,53)
 *
 * FORMAT STATUS: "format unpinned" for whole files - no file written by the reference is available in this
 * image (no Rust toolchain).  Pinned against reference-held bytes / cases: the sstable block framing
 * (sstable/mod.rs:373-396 test_simple_sstable), the meta.json shape (index_meta.rs:436-440), the directory-footer
 * refusals (directory/footer.rs:169-235) and the vint lengths (sstable/vint.rs:47-60).  tests/ build
 * fixtures with a Python writer that follows the serialisers cited above line by line; tools/ref_golden.rs
 * is the program that writes the same fixture with the reference itself.
 *
 * All functions: extern "C", never unwind, 0 = ok, negative = HB_ERR_* of hyperball.h.  Host only (no GPU).
 */
#ifndef HB_WEBGRAPH_H
#define HB_WEBGRAPH_H

#include <stdint.h>

#include "hyperball.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hbw_reader hbw_reader;

#define HBW_VERIFY_CRC 0x1u /* check every .col file's CRC-32 against its footer (reads the whole file once) */
#define HBW_PAGE_IDS   0x2u /* also locate the page-level `from_id` / `to_id` columns (webgraph/schema.rs:132-180): the records
                               the reference's tail mode queries (harmonic.rs:82-87), see HB_FLAG_REFERENCE_TAIL */

/* Opens `<webgraph>/edges` (the directory holding meta.json).  Maps every segment's .col file and locates the three
 * columns; fails if a column is missing, has another codec than Raw, or row counts disagree with meta.json. */
int hbw_open(const char *edges_dir, uint32_t flags, hbw_reader **out);
void hbw_close(hbw_reader *r);
/* Message of the last failing call (r == NULL: of hbw_open). */
const char *hbw_last_error(const hbw_reader *r);

int hbw_num_segments(const hbw_reader *r, uint64_t *count);
/* uuid: 32 hex chars + NUL. */
int hbw_segment_info(const hbw_reader *r, uint64_t segment, char uuid[33], uint64_t *num_rows);
/* Documents of all segments = records host_edges() would yield before de-duplication. */
int hbw_total_rows(const hbw_reader *r, uint64_t *rows);

/* Copies `count` records starting at stream position `first` (segments in meta.json order, documents ascending:
 * SmallSegmentEdgesIter order) into out[]: {from_host_id, to_host_id, rel_flags} = SmallEdge (edge.rs:31-35). */
int hbw_read_host_edges(const hbw_reader *r, uint64_t first, uint64_t count, hb_edge *out);

/* Same positions, page-level ids: {from_id, to_id, rel_flags} (reader opened with HBW_PAGE_IDS). */
int hbw_read_page_edges(const hbw_reader *r, uint64_t first, uint64_t count, hb_edge *out);

/* Replaces `HarmonicCentrality::calculate(&Webgraph)`'s input side end to end (harmonic.rs:292, :58-72, :116-131):
 * streams the store's records into ctx in slabs (hb_append_edges) and finalizes with the node set derived from all
 * endpoints (= host_nodes()).  Then hb_run() as usual.  With HBW_PAGE_IDS (ctx created with HB_FLAG_REFERENCE_TAIL) the
 * page-level records follow through hb_append_tail_edges, so the run is `stract centrality harmonic` as written. */
int hb_load_webgraph(hb_ctx *ctx, const char *edges_dir, uint32_t flags);

/* ---- test exports ------------------------------------------------------------------------------------------- */
/* Decodes an sstable (dictionary bytes incl. its 20-byte footer).  Keys are written to keys_out as
 * [len u32][bytes]...; value_mode 0 = no values (VoidSSTable), 1 = byte ranges (RangeSSTable): ranges_out gets
 * {start, end} pairs.  Returns the number of entries in *count. */
int hbw_debug_sstable(const uint8_t *bytes, uint64_t len, int value_mode, uint8_t *keys_out, uint64_t keys_cap,
                      uint64_t *ranges_out, uint64_t ranges_cap, uint64_t *count);
/* CRC-32 (IEEE, crc32fast) of a buffer. */
uint32_t hbw_debug_crc32(const uint8_t *bytes, uint64_t len);

#ifdef __cplusplus
}
#endif
#endif /* HB_WEBGRAPH_H */
