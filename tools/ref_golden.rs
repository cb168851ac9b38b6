//! ref_golden.rs - run the REFERENCE on small graphs and dump what parity needs.
//!
//! Not buildable in this repository's image (no rustc / cargo, ~600 un-vendored crates).  On a machine that builds
//! StractOrg/stract (snapshot 2024-12-20):
//!
//!     cp tools/ref_golden.rs  <stract>/crates/core/examples/ref_golden.rs
//!     printf '\n[[example]]\nname = "ref_golden"\n' >> <stract>/crates/core/Cargo.toml
//!     cd <stract> && cargo run --release -p stract --example ref_golden -- /tmp/ref_out
//!     cp /tmp/ref_out/reference_*.json <this repo>/tests/golden/
//!     cp -r /tmp/ref_out/store_fixture <this repo>/tests/golden/reference_store      # whole edge store, for the column reader
//!
//! Second use - pinning the native centrality-store writer (include/hb_store.h): write a store with this repository
//! (`python -c "from stract_amd import _lib; ..."` or hb_store_harmonic from C), then let the REFERENCE's reader open it:
//!
//!     cargo run --release -p stract --example ref_golden -- --check-store /path/to/output
//!
//! opens <output>/harmonic and <output>/harmonic_rank with speedy_kv::Db::open_or_create (crates/speedy-kv/src/lib.rs:234),
//! walks `iter()`, looks every key up again with `get()` (bloom filter + fst + blob index + blob file) and writes
//! <output>/reference_read_back.txt ("id_hex32 f64_bits_hex16 rank" per line, iteration order) for comparison with the arrays
//! the library returned.  It also writes the reference's OWN stores of graph 2 under <out dir>/reference_centrality_store.
//!
//! and `pytest tests` then pins the oracle, the HIP path and the native column reader to the reference with no
//! other change (tests/test_reference_golden.py).  Every graph goes through `Webgraph::insert`
//! (crates/core/src/webgraph/mod.rs:106), is read back through the same `host_edges()` the algorithm uses
//! (mod.rs:192), and is scored by `HarmonicCentrality::calculate` (centrality/harmonic.rs:292).
//!
//! Output, one JSON file per graph:
//!   { "name": ..., "edges": [[from_hex32, to_hex32, rel_flags_u64], ...]   // host_edges() stream, AFTER its unique_by
//!     "centrality": [[id_hex32, f64_bits_hex16], ...]                       // HarmonicCentrality::iter(), ascending id
//!     "pages": [[from_id_hex32, to_id_hex32, rel_flags_u64], ...] }         // page_edges(): only for the mixed graph (4)
use std::fmt::Write as _;
use std::path::Path;

use stract::webgraph::centrality::harmonic::HarmonicCentrality;
use stract::webgraph::{Edge, Node, NodeID, Webgraph};
use stract::webpage::html::links::RelFlags;

fn build(path: &Path, edges: &[(String, String, RelFlags)], commit_every: usize) -> Webgraph {
    let mut graph = Webgraph::builder(path, 0u64.into()).open().unwrap();
    for (i, (from, to, flags)) in edges.iter().enumerate() {
        graph
            .insert(Edge {
                from: Node::from_str_not_validated(from),
                to: Node::from_str_not_validated(to),
                rel_flags: *flags,
                ..Edge::empty()
            })
            .unwrap();
        if (i + 1) % commit_every == 0 {
            graph.commit().unwrap(); // several segments: exercises the cross-segment first-occurrence rule
        }
    }
    graph.commit().unwrap();
    graph
}

fn dump(name: &str, graph: &Webgraph, out_dir: &Path) {
    dump_with_pages(name, graph, out_dir, false)
}

fn dump_with_pages(name: &str, graph: &Webgraph, out_dir: &Path, pages: bool) {
    let mut s = String::new();
    write!(s, "{{\"name\":\"{name}\",\"edges\":[").unwrap();
    for (i, e) in graph.host_edges().enumerate() {
        if i > 0 {
            s.push(',');
        }
        write!(s, "[\"{:032x}\",\"{:032x}\",{}]", e.from.as_u128(), e.to.as_u128(), e.rel_flags.as_u64()).unwrap();
    }
    s.push_str("],\"centrality\":[");
    let hc = HarmonicCentrality::calculate(graph);
    for (i, (id, c)) in hc.iter().enumerate() {
        if i > 0 {
            s.push(',');
        }
        write!(s, "[\"{:032x}\",\"{:016x}\"]", id.as_u128(), c.to_bits()).unwrap();
    }
    s.push(']');
    if pages {
        // the page-level records update_changed_counters queries in the tail (harmonic.rs:82-87, mod.rs:183)
        s.push_str(",\"pages\":[");
        for (i, e) in graph.page_edges().enumerate() {
            if i > 0 {
                s.push(',');
            }
            write!(s, "[\"{:032x}\",\"{:032x}\",{}]", e.from.as_u128(), e.to.as_u128(), e.rel_flags.as_u64()).unwrap();
        }
        s.push(']');
    }
    s.push('}');
    std::fs::write(out_dir.join(format!("reference_{name}.json")), s).unwrap();
}

/// A directory written by hb_store_harmonic, read with the reference's own reader (see the header).
fn check_store(dir: &Path) {
    let cen: speedy_kv::Db<NodeID, f64> = speedy_kv::Db::open_or_create(dir.join("harmonic")).unwrap();
    let rank: speedy_kv::Db<NodeID, u64> = speedy_kv::Db::open_or_create(dir.join("harmonic_rank")).unwrap();
    let mut s = String::new();
    let mut n = 0usize;
    for (id, c) in cen.iter() {
        let again = cen.get(&id).unwrap().expect("key found by iter() but not by get(): bloom filter or fst map");
        assert_eq!(c.to_bits(), again.to_bits());
        let r = rank.get(&id).unwrap().expect("key missing from harmonic_rank");
        writeln!(s, "{:032x} {:016x} {}", id.as_u128(), c.to_bits(), r).unwrap();
        n += 1;
    }
    assert_eq!(n, cen.len());
    assert_eq!(n, rank.len());
    std::fs::write(dir.join("reference_read_back.txt"), s).unwrap();
    println!("store ok: {n} entries readable by speedy_kv");
}

fn main() {
    if std::env::args().nth(1).as_deref() == Some("--check-store") {
        let dir = std::env::args().nth(2).expect("usage: ref_golden --check-store <output dir of hb_store_harmonic>");
        return check_store(Path::new(&dir));
    }
    let out = std::env::args().nth(1).expect("usage: ref_golden <out dir> | ref_golden --check-store <dir>");
    let out = Path::new(&out);
    std::fs::create_dir_all(out).unwrap();
    let none = RelFlags::default();

    // 1. the fixture of the reference's own tests (harmonic.rs:323-341)
    let fixture: Vec<(String, String, RelFlags)> = [("A", "B"), ("B", "C"), ("A", "C"), ("C", "A"), ("D", "C")]
        .iter()
        .map(|(a, b)| (format!("{a}.com"), format!("{b}.com"), none))
        .collect();
    dump("fixture", &build(&out.join("g_fixture"), &fixture, usize::MAX), out);

    // 2. SURVEY.md Appendix B: n = 200 hosts, 1200 unique non-self edges from a 64-bit LCG, three commits
    let (mut x, n) = (12345u64, 200u64);
    let mut lcg = || {
        x = x.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407);
        (x >> 33) % n + 1
    };
    let mut seen = std::collections::BTreeSet::new();
    while seen.len() < 1200 {
        let (f, t) = (lcg(), lcg());
        if f != t {
            seen.insert((f, t));
        }
    }
    let lcg_edges: Vec<_> = seen.iter().map(|(f, t)| (format!("host{f}.com"), format!("host{t}.com"), none)).collect();
    let g_lcg = build(&out.join("g_lcg"), &lcg_edges, 500);
    dump("lcg200", &g_lcg, out);
    // the reference's own centrality stores of this graph (centrality/mod.rs:72-114): files to compare a native store with
    let hc = HarmonicCentrality::calculate(&g_lcg);
    stract::webgraph::centrality::store_harmonic(hc.iter().map(|(n, c)| (*n, c)), out.join("reference_centrality_store"));

    // 3. ingest semantics: flagged first occurrences, later clean copies, duplicates across commits, self links
    let mut salted = lcg_edges.clone();
    for (i, e) in lcg_edges.iter().enumerate().take(300) {
        if i % 3 == 0 {
            salted.insert(i, (e.0.clone(), e.1.clone(), RelFlags::NOFOLLOW)); // flagged copy BEFORE the clean one: pair is lost
        } else if i % 3 == 1 {
            salted.push((e.0.clone(), e.1.clone(), RelFlags::TAG)); // flagged copy AFTER: ignored
        } else {
            salted.push((e.0.clone(), e.0.clone(), none)); // self link
        }
    }
    let g = build(&out.join("store_fixture"), &salted, 400);
    dump("salted", &g, out);
    // out/store_fixture/edges now holds meta.json + the .col files of this graph: the fixture for the native column reader

    // 4. pages AND hosts: the sqrt(n) tail follows page-level links whose source page id equals
    //    a host id (root pages); a chain of hosts linked alternately from root pages and from sub-pages makes the run
    //    end earlier than the host-level iteration would.  Consumed with HB_FLAG_REFERENCE_TAIL.
    let mut mixed: Vec<(String, String, RelFlags)> = lcg_edges.clone();
    mixed.push(("host1.com".to_string(), "tail0.com".to_string(), none));
    for k in 0..60 {
        let from = if k % 2 == 0 { format!("tail{k}.com") } else { format!("tail{k}.com/links.html") };
        let to = if k % 3 == 0 { format!("tail{}.com/about", k + 1) } else { format!("tail{}.com", k + 1) };
        mixed.push((from, to, none));
    }
    // ONE commit = one segment: what a ForwardlinksQuery returns depends on the documents' order inside a segment (its
    // LinksScorer de-duplicates neighbouring documents per segment, query/raw/links.rs:115-232), and the consumers of this
    // file take `pages` as one segment in doc order (hb_load_tail_edges / hbo.faithful_run(.., pages))
    dump_with_pages("mixed_pages", &build(&out.join("g_mixed"), &mixed, usize::MAX), out, true);
}
