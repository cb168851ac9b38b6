#!/usr/bin/env python3
"""Block coverage of the library's sources - device code included - by the GPU parity tests, measured WITHOUT a GPU: the tests run
against the coverage build of the interpreted library (tests/simt, `make cov`: -fsanitize-coverage=trace-pc-guard,pc-table), every
process leaves a <pid>.cov file, this tool merges them and maps the instrumented blocks to source lines through the DWARF line table (llvm-dwarfdump).

usage: tools/simt_coverage.py [--select PYTEST_K_EXPRESSION] [--out FILE] [--list FILE.hip.h ...]
prints, per source file under stract_amd/csrc, source lines reached / lines with code, and (--list) the lines no test reached."""
import argparse
import collections
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMT = os.path.join(ROOT, "tests", "simt")
LIB = os.path.join(SIMT, "_build_cov", "libhyperball_simt_cov.so")
DWARFDUMP = "/opt/rocm/lib/llvm/bin/llvm-dwarfdump"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--select", default="not test_c2 and not test_caching_allocator_under_memory_pressure")
    ap.add_argument("--out", default="")
    ap.add_argument("--list", nargs="*", default=[])
    ap.add_argument("--extra", nargs="*", default=[], help="more commands to run under the coverage build (each one string)")
    a = ap.parse_args()
    subprocess.check_call(["make", "-s", "-j8", "-C", SIMT, "cov"])
    cov_dir = tempfile.mkdtemp(prefix="hb_cov_")
    env = dict(os.environ, HB_LIB_PATH=LIB, HB_ALLOW_SIMT_INTERPRETER="1", PYTHONPATH=ROOT, HB_SIMT_COV_DIR=cov_dir)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu.py"), "-m", "gpu", "-q", "-k", a.select, "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True)
    summary = (r.stdout.strip().splitlines() or ["?"])[-1]
    for cmd in a.extra:
        subprocess.run(cmd, shell=True, env=env, cwd=ROOT)
    hit = collections.defaultdict(int)
    for path in glob.glob(os.path.join(cov_dir, "*.cov")):
        for line in open(path):
            off, h = line.split()
            hit[off] |= int(h)
    # instrumented blocks (start offsets) -> source lines: every row of the DWARF line table belongs to the block that starts at or
    # before it; a line is REACHED if some row of it lies in a block that ran, MISSED if all its rows lie in blocks that never ran
    # (the block-start line alone would blame a line for its error-return branch)
    pcs = sorted(int(o, 16) for o in hit)
    ran = [hit["%x" % pc] for pc in pcs]
    import bisect
    import re
    dump = subprocess.run([DWARFDUMP, "--debug-line", LIB], capture_output=True, text=True).stdout
    reached = collections.defaultdict(set)
    seen = collections.defaultdict(set)
    files = {}
    row = re.compile(r"^0x([0-9a-f]{16})\s+(\d+)\s+\d+\s+(\d+)\s")
    name_re = re.compile(r'^\s+name: "(.*)"')
    idx_re = re.compile(r"^file_names\[\s*(\d+)\]:")
    cur = None
    for ln in dump.splitlines():
        if ln.startswith("debug_line["):
            files = {}
            continue
        m = idx_re.match(ln)
        if m:
            cur = int(m.group(1))
            continue
        m = name_re.match(ln)
        if m and cur is not None:
            files[cur] = os.path.basename(m.group(1))
            cur = None
            continue
        m = row.match(ln)
        if not m:
            continue
        addr, line, fidx = int(m.group(1), 16), int(m.group(2)), int(m.group(3))
        name = files.get(fidx, "")
        if not line or not (name.startswith("hb_") or name.startswith("hll64")) or name.endswith(".cpp") or name == "hb_threads.h":
            continue  # (the host-only .cpp files are compiled without the callbacks: not part of this measurement)
        k = bisect.bisect_right(pcs, addr) - 1
        if k < 0:
            continue
        seen[name].add(line)
        if ran[k]:
            reached[name].add(line)
    out = ["tests: %s   (%s)" % (summary, a.select),
           "source lines with code, by file: reached by some test / all (a line counts as reached if any instantiation or inlined copy of it ran)",
           "%-24s %8s %8s %7s" % ("file", "reached", "lines", "share")]
    tot = [0, 0]
    for name in sorted(seen):
        h, n = len(reached[name]), len(seen[name])
        tot[0] += h
        tot[1] += n
        out.append("%-24s %8d %8d %6.1f%%" % (name, h, n, 100.0 * h / max(n, 1)))
    out.append("%-24s %8d %8d %6.1f%%" % ("all", tot[0], tot[1], 100.0 * tot[0] / max(tot[1], 1)))
    for name in (a.list or []):
        only_missed = sorted(seen[name] - reached[name])
        out.append("\n%s: lines never reached: %s" % (name, " ".join(map(str, only_missed))))
    text = "\n".join(out)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
