import sys
sys.path.insert(0, '.')
from stract_amd import _lib, synth
g, scale, label = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "C3")
with _lib.Context(flags=int(sys.argv[2], 0) if len(sys.argv) > 2 else 0) as ctx:
    ctx.load_dense(g.ids, g.row_ptr, g.src)
    for rep in range(2):
        ctx.run()
    for ps in ctx.pass_stats():
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in ps.items() if k in ("pass", "changed", "active_edges", "touched", "mode", "ms_gpu", "ms_main", "ms_level1")})
