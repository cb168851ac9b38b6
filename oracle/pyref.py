"""pyref.py - a SECOND, independent restatement of the reference arithmetic, in plain Python.

TEST INFRASTRUCTURE ONLY.  Written directly from the Rust sources (not from hb_oracle.c) so that the two
restatements can be checked against each other (tests/test_oracle.py): a transcription slip in one of them
shows up as a disagreement.  Small inputs only (pure-Python loops).

Follows, line by line:
  crates/core/src/hyperloglog.rs:4311-4313  FastHasher::hash
  crates/core/src/hyperloglog.rs:4366-4383  am(), b()
  crates/core/src/hyperloglog.rs:4385-4400  add, add_u128
  crates/core/src/hyperloglog.rs:4407-4470  estimate_bias
  crates/core/src/hyperloglog.rs:4472-4480  linear_counting, threshold
  crates/core/src/hyperloglog.rs:4484-4516  size
  crates/core/src/hyperloglog.rs:4531-4535  merge
  crates/core/src/kahan_sum.rs:35-54        KahanSum
  crates/core/src/webgraph/centrality/harmonic.rs:53-287  the HyperBall loop (map-based, host-level)
Parity status: unpinned against the reference binary (no Rust toolchain here), like hb_oracle.c.
"""
import math
import os
import re
import struct

_HERE = os.path.dirname(os.path.abspath(__file__))
MASK64 = (1 << 64) - 1
N = 64
SKIPPED_REL = 0x6FED00  # harmonic.rs:36-49 with the bit values of webpage/html/links.rs:114-141


def _load_tables():
    """RAW_ESTIMATE_DATA_VEC[1] / BIAS_DATA_VEC[1] (the rows N = 64 indexes, hyperloglog.rs:4411,4467) as
    extracted from the reference by oracle/gen_tables.py into hll64_tables.h."""
    text = open(os.path.join(_HERE, "hll64_tables.h")).read()
    out = []
    for name in ("HLL64_RAW_ESTIMATE", "HLL64_BIAS"):
        body = re.search(name + r"\[HLL64_TABLE_LEN\] = \{(.*?)\};", text, re.S).group(1)
        out.append([float(x) for x in body.replace("\n", " ").split(",") if x.strip()])
    assert len(out[0]) == 159 and len(out[1]) == 159
    return out


RAW, BIAS = _load_tables()
THRESHOLD_B6 = 40  # THRESHOLD_DATA_VEC[6 - 4], hyperloglog.rs:27-45


def _load_extra():
    """Rows of the same reference tables for other register counts (oracle/hll_tables_extra.json, extracted from
    hyperloglog.rs by the snippet in oracle/gen_tables.py's header): b - 1 - 4 = 2 (N = 128) and 11 (N = 65 536),
    plus THRESHOLD_DATA_VEC - what the reference's own HyperLogLog tests use (hyperloglog.rs:4553-4611)."""
    import json
    d = json.load(open(os.path.join(_HERE, "hll_tables_extra.json")))
    raw = {int(k): v for k, v in d["raw"].items()}
    bias = {int(k): v for k, v in d["bias"].items()}
    assert raw[1] == RAW and bias[1] == BIAS  # the N = 64 rows agree with hll64_tables.h
    return raw, bias, d["threshold"]


RAW_ROWS, BIAS_ROWS, THRESHOLDS = _load_extra()


def hll_b(n):
    return int(math.log2(n))  # (N as f64).log2() as usize, hyperloglog.rs:4381-4383


def hll_am(m):
    """am(), hyperloglog.rs:4366-4378"""
    if m >= 128:
        return 0.7213 / (1.0 + (1.079 / float(m)))
    if m >= 64:
        return 0.709
    if m >= 32:
        return 0.697
    return 0.673


def fast_hash(item):
    return (item * 11400714819323198549) & MASK64  # wrapping_mul


def leading_zeros64(w):
    return 64 - w.bit_length()


def hll_new(n=N):
    return [0] * n


def hll_add(reg, item):
    b = hll_b(len(reg))  # (N as f64).log2() as usize (6 for the 64-register counters of the centrality path)
    h = fast_hash(item & MASK64)  # add_u128: `item as u64`
    j = h >> (64 - b)
    w = (h << b) & MASK64
    p = leading_zeros64(w) + 1
    reg[j] = max(reg[j], p & 0xFF)


def hll_merge(dst, src):
    for i in range(len(dst)):
        dst[i] = max(dst[i], src[i])


def total_cmp(a, b):
    """f64::total_cmp as -1/0/1."""
    def key(x):
        (bits,) = struct.unpack("<q", struct.pack("<d", x))
        return bits ^ ((bits >> 63) & 0x7FFFFFFFFFFFFFFF)
    ka, kb = key(a), key(b)
    return (ka > kb) - (ka < kb)


def binary_search_by(arr, e):
    """slice::binary_search_by(|v| v.total_cmp(&e)) of Rust >= 1.82 (library/core/src/slice/mod.rs):
    returns ("Ok", i) or ("Err", i)."""
    size = len(arr)
    if size == 0:
        return ("Err", 0)
    base = 0
    while size > 1:
        half = size // 2
        mid = base + half
        cmp = total_cmp(arr[mid], e)
        base = base if cmp > 0 else mid
        size -= half
    cmp = total_cmp(arr[base], e)
    if cmp == 0:
        return ("Ok", base)
    return ("Err", base + (1 if cmp < 0 else 0))


def estimate_bias(e, n=N):
    K = 6
    row = hll_b(n) - 1 - 4  # hyperloglog.rs:4411: RAW_ESTIMATE_DATA_VEC[b - 1 - RAW_ESTIMATE_DATA_OFFSET]
    lookup, bias_row = RAW_ROWS[row], BIAS_ROWS[row]
    kind, i = binary_search_by(lookup, e)
    if kind == "Err" and i == len(lookup):
        idx_left = i - 1
    else:
        idx_left = i
    idx_right = idx_left + 1 if idx_left < len(lookup) - 1 else None
    neighbors = []
    for _ in range(K):
        if idx_left is not None and idx_right is not None:
            delta_left = abs(lookup[idx_left] - e)
            delta_right = abs(lookup[idx_right] - e)
            if delta_right < delta_left:
                right, idx = True, idx_right
            else:
                right, idx = False, idx_left
        elif idx_left is not None:
            right, idx = False, idx_left
        elif idx_right is not None:
            right, idx = True, idx_right
        else:
            raise AssertionError("neighborhood search failed")
        neighbors.append(idx)
        if right:
            idx_right = idx + 1 if idx < len(lookup) - 1 else None
        else:
            idx_left = idx - 1 if idx > 0 else None
    s = 0.0
    for i in neighbors:  # Iterator::sum is a left fold
        s = s + bias_row[i]
    return s / float(K)


def as_usize(x):
    """Rust `f64 as usize`: truncation toward zero, saturating, NaN -> 0."""
    if x != x or x <= 0.0:
        return 0
    if x >= 18446744073709551616.0:
        return MASK64
    return int(x)


def hll_size(reg):
    """HyperLogLog<N>::size for N = len(reg) (hyperloglog.rs:4484-4516)."""
    n = len(reg)
    m = float(n)
    mx = max(reg) if n else 0
    if mx <= 30 and n <= (1 << 22):
        # every partial sum of the left fold is a multiple of 2^-30 below 2^22: exact in f64 whatever the order,
        # so the fold can be done in integers (what makes N = 65 536 affordable in Python)
        s = float(sum(1 << (30 - int(val)) for val in reg)) / float(1 << 30)
    else:
        s = 0.0
        for val in reg:  # ONE_OVER_POWER_OF_TWO[val] == 2^-val exactly (hyperloglog.rs:4043-4300)
            s = s + math.ldexp(1.0, -val)
    z = 1.0 / s
    e = hll_am(n) * (m * m) * z  # am() * m.powi(2) * z, left to right
    if e <= 5.0 * m:
        e_star = e - estimate_bias(e, n)
    else:
        e_star = e
    v = sum(1 for r in reg if r == 0)
    if v != 0:
        h = m * math.log(m / float(v))
    else:
        h = e_star
    if h <= float(THRESHOLDS[hll_b(n) - 4]):  # threshold(): THRESHOLD_DATA_VEC[b - THRESHOLD_DATA_OFFSET]
        return as_usize(h)
    return as_usize(e_star)


def hll_relative_error(n):
    return 1.04 / math.sqrt(float(n))  # hyperloglog.rs:4519-4521


def hll_size_bounds(reg):
    """size_bounds(), hyperloglog.rs:4523-4529"""
    size = hll_size(reg)
    delta = as_usize(hll_relative_error(len(reg)) * 2.0 * float(size))
    return size - delta, size + delta


class Kahan:
    __slots__ = ("sum", "err")

    def __init__(self):
        self.sum = 0.0
        self.err = 0.0

    def add(self, rhs):
        y = rhs - self.err
        t = self.sum + y
        self.err = (t - self.sum) - y
        self.sum = t


def harmonic_centrality(edges):
    """edges: stream of (from_id, to_id, rel_flags) with integer ids (u128).  Returns (dict id -> f64 in
    ascending id order, passes).  Semantics of harmonic.rs:53-287 on host-level edges; the bloom filter /
    exact-set switch is results-inert and replaced by the exact changed set."""
    nodes = sorted({x for e in edges for x in (e[0], e[1])})  # store.rs:338-357: flagged records included
    seen, kept = set(), []
    for e in edges:  # store.rs:313 unique_by first, THEN the rel filter (harmonic.rs:131)
        k = (e[0], e[1])
        if k in seen:
            continue
        seen.add(k)
        if (e[2] if len(e) > 2 else 0) & SKIPPED_REL:
            continue
        kept.append(k)
    old = {}
    for v in nodes:  # initialize, harmonic.rs:53-73
        c = hll_new()
        hll_add(c, v)
        old[v] = c
    new = {v: list(c) for v, c in old.items()}
    cent = {v: Kahan() for v in nodes}
    n = len(nodes)
    changed_nodes = set(nodes)  # harmonic.rs:221-225
    has_changes = True
    t = 0
    while has_changes:  # harmonic.rs:237-280
        has_changes = False
        new_changed = set()
        for (f, to) in kept:  # update_all_counters / update_changed_counters
            if f not in changed_nodes:
                continue
            fc, tc = old[f], new[to]
            if any(a > b for a, b in zip(fc, tc)):
                hll_merge(tc, fc)
                new_changed.add(to)
                has_changes = True
        for v in nodes:  # update_centralities, harmonic.rs:159-176
            sn, so = hll_size(new[v]), hll_size(old[v])
            d = sn - so if sn >= so else 0  # checked_sub().unwrap_or_default()
            cent[v].add(float(d) / float(t + 1))
        old = {v: list(c) for v, c in new.items()}  # Counters::step
        changed_nodes = new_changed
        t += 1
    out = {}
    if n >= 2:
        norm = float(n - 1)
        for v in nodes:  # normalize_centralities, harmonic.rs:178-195
            c = cent[v].sum
            if c > 0.0:
                c = c / norm
                out[v] = c if math.isfinite(c) else 0.0
    return out, t


# ---- the reference's loop with its changed-set machinery spelled out (bloom + sqrt(n) exact tail) --------------------
LARGE_PRIME = 11400714819323198549  # bloom/src/lib.rs:36


def bloom_num_bits(estimated_items, fp=0.05):
    """bloom/src/lib.rs:38-41."""
    return int(math.ceil(float(estimated_items) * math.log(fp) / (-8.0 * math.log(2.0) ** 2)))


class Bloom:
    """U64BloomFilter (bloom/src/lib.rs:60-123): one multiplicative hash of the LOW 64 bits of the id."""

    def __init__(self, estimated_items, fp=0.05):
        self.num_bits = bloom_num_bits(estimated_items, fp)
        self.bits = set()

    def _slot(self, item):
        return (((item & MASK64) * LARGE_PRIME) & MASK64) % self.num_bits  # :85-93 (usize = u64)

    def insert(self, item):
        self.bits.add(self._slot(item))

    def contains(self, item):
        return self._slot(item) in self.bits

    def estimate_card(self):
        """:108-123 - the logarithm is cast to i64 BEFORE the multiplication (truncation toward zero)."""
        ones = len(self.bits)
        if ones == 0 or self.num_bits == 0:
            return 0
        if ones == self.num_bits:
            return (1 << 64) - 1
        v = -self.num_bits * int(math.log(1.0 - float(ones) / float(self.num_bits)))
        return v if 0 <= v < (1 << 64) else 0  # try_into::<u64>().unwrap_or_default()


TERMINATED = object()  # tantivy::TERMINATED (a doc id no document has)
COMPRESSION_BLOCK_SIZE = 128


class _SegmentPostings:
    """tantivy SegmentPostings + BlockSegmentPostings + SkipReader as far as LinksScorer uses them
    (crates/tantivy/src/postings/segment_postings.rs:145-163, block_segment_postings.rs:355-360, skip.rs:119-132,256-283):
    the posting list is cut into full blocks of 128 documents plus one final partial block; `cur` walks the loaded
    block (padded with TERMINATED); the skip reader knows the last document of FULL blocks only."""

    def __init__(self, ndocs):
        self.ndocs = ndocs
        self.block = 0  # index of the loaded block
        self.cur = 0

    def _block_len(self, b):
        return max(0, min(COMPRESSION_BLOCK_SIZE, self.ndocs - b * COMPRESSION_BLOCK_SIZE))

    def doc(self):  # position in the posting list, or TERMINATED
        return self.block * COMPRESSION_BLOCK_SIZE + self.cur if self.cur < self._block_len(self.block) else TERMINATED

    def last_doc_in_block(self):  # skip.rs: remaining_docs >= 128 ? stored last doc : TERMINATED
        if self._block_len(self.block) == COMPRESSION_BLOCK_SIZE:
            return self.block * COMPRESSION_BLOCK_SIZE + COMPRESSION_BLOCK_SIZE - 1
        return TERMINATED

    def block_advance(self):  # BlockSegmentPostings::advance
        self.block += 1

    def reset_cursor_start_block(self):
        self.cur = 0

    def advance(self):  # SegmentPostings::advance
        if self.cur == COMPRESSION_BLOCK_SIZE - 1:
            self.cur = 0
            self.block_advance()
        else:
            self.cur += 1
        return self.doc()


def links_scorer_docs(to_ids, self_id):
    """LinksScorer (crates/core/src/webgraph/query/raw/links.rs:115-232) driven like a collector drives a DocSet
    (`doc = scorer.doc(); while doc != TERMINATED { collect(doc); doc = scorer.advance(); }`): to_ids = the ToId
    column of one segment's posting list of term from_id == self_id, in doc order.  Returns the yielded positions."""
    if not to_ids:
        return []  # read_postings finds no term: EmptyScorer (:88-101)
    postings = _SegmentPostings(len(to_ids))

    def dedup_val(doc):  # :179-185
        return None if doc is TERMINATED else ("v", to_ids[doc])

    # LinksScorer::new, :143-165
    last = dedup_val(postings.doc())
    while postings.doc() is not TERMINATED and last == ("v", self_id):
        postings.advance()
        last = dedup_val(postings.doc()) if postings.doc() is not TERMINATED else None

    def has_seen(doc):  # :186-190
        dv = dedup_val(doc)
        return dv is not None and last == dv

    def skip_self(doc):  # :192-194
        return dedup_val(doc) == ("v", self_id)

    out = []
    doc = postings.doc()
    while doc is not TERMINATED:
        out.append(doc)
        # advance(), :199-229
        postings.advance()
        while has_seen(postings.last_doc_in_block()) and postings.doc() is not TERMINATED:
            postings.block_advance()
            postings.reset_cursor_start_block()
        while (has_seen(postings.doc()) or skip_self(postings.doc())) and postings.doc() is not TERMINATED:
            postings.advance()
        dv = dedup_val(postings.doc())
        if dv is not None:
            last = dv
        doc = postings.doc()
    return out


def forwardlinks_result(pages, node_set, segments=None):
    """What `graph.search(ForwardlinksQuery::new(h).with_limit(Unlimited))` + the filter of harmonic.rs:87 leave, for
    every host node h: {h: [to, ...]} over the page-level documents `pages` = [(from_id, to_id, rel_flags)] in doc
    order; segments = list of segment lengths (None: one segment).  One LinksScorer per segment and term."""
    fwd = {}
    if segments is None:
        segments = [len(pages)]
    base = 0
    for cnt in segments:
        lists = {}
        for e in pages[base:base + cnt]:
            if e[0] in node_set:
                lists.setdefault(e[0], []).append(e)
        for h, docs in lists.items():
            for i in links_scorer_docs([d[1] for d in docs], h):
                d = docs[i]
                if (d[2] if len(d) > 2 else 0) & SKIPPED_REL:  # harmonic.rs:87, on the YIELDED document
                    continue
                if d[1] in node_set:  # :91-92
                    fwd.setdefault(h, []).append(d[1])
        base += cnt
    return fwd


def harmonic_centrality_reference(edges, pages=None, segments=None):
    """calculate_centrality (harmonic.rs:215-287) with the bloom filter, the exact-counting switch and the
    sqrt(n) tail as written.  pages: the page-level (from_id, to_id, rel_flags) records ForwardlinksQuery matches
    in the tail, in doc order, `segments` = their segment lengths (None: one segment) - the query's
    LinksScorer de-duplicates neighbouring documents per segment before the rel filter sees them;
    pages None = host-level edges (then the result equals harmonic_centrality()).
    Returns (dict, passes, passes that took update_changed_counters)."""
    nodes = sorted({x for e in edges for x in (e[0], e[1])})
    n = len(nodes)
    if n == 0:
        return {}, 0, 0
    seen, kept = set(), []
    for e in edges:
        k = (e[0], e[1])
        if k in seen:
            continue
        seen.add(k)
        if (e[2] if len(e) > 2 else 0) & SKIPPED_REL:
            continue
        kept.append(k)
    node_set = set(nodes)
    fwd = {}
    if pages is None:
        for (f, to) in kept:
            if f in node_set and to in node_set:  # harmonic.rs:91-92
                fwd.setdefault(f, []).append(to)
    else:
        fwd = forwardlinks_result(list(pages), node_set, segments)
    old = {}
    for v in nodes:
        c = hll_new()
        hll_add(c, v)
        old[v] = c
    new = {v: list(c) for v, c in old.items()}
    cent = {v: Kahan() for v in nodes}
    changed = Bloom(n)  # :221-225
    for v in nodes:
        changed.insert(v)
    threshold = int(round(max(math.sqrt(float(n)), 0.0)))  # :228 (f64::round: half away from zero; sqrt is never x.5)
    exact_counting, has_changes, t, tail_passes = False, True, 0, 0
    exact_changed = set()
    while has_changes:
        new_changed = Bloom(n)
        has_changes = False
        if exact_changed and len(exact_changed) <= threshold:  # :244-252 update_changed_counters
            tail_passes += 1
            nxt = set()
            for u in sorted(exact_changed):
                for to in fwd.get(u, ()):
                    fc, tc = old[u], new[to]
                    if any(a > b for a, b in zip(fc, tc)):
                        hll_merge(tc, fc)
                        new_changed.insert(to)
                        nxt.add(to)
                        has_changes = True
            exact_changed = nxt
        else:  # update_all_counters, :116-157
            track = exact_counting
            if track:
                exact_changed = set()
            for (f, to) in kept:
                if not changed.contains(f):
                    continue
                fc, tc = old[f], new[to]
                if any(a > b for a, b in zip(fc, tc)):
                    hll_merge(tc, fc)
                    new_changed.insert(to)
                    if track:
                        exact_changed.add(to)
                    has_changes = True
        for v in nodes:
            sn, so = hll_size(new[v]), hll_size(old[v])
            cent[v].add(float(sn - so if sn >= so else 0) / float(t + 1))
        old = {v: list(c) for v, c in new.items()}
        changed = new_changed
        t += 1
        if changed.estimate_card() <= threshold:  # :277-279
            exact_counting = True
    out = {}
    if n >= 2:
        norm = float(n - 1)
        for v in nodes:
            c = cent[v].sum
            if c > 0.0:
                c = c / norm
                out[v] = c if math.isfinite(c) else 0.0
    return out, t, tail_passes
