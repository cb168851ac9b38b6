import time, sys
sys.path.insert(0, '.')
t0=time.perf_counter()
import numpy as np
from stract_amd import _lib, synth
from oracle import hbo
t=time.perf_counter(); print("imports %.3f"%(t-t0))
def T(label, f):
    t=time.perf_counter(); r=f(); print("%-22s %.3f s"%(label, time.perf_counter()-t)); return r
g=T("rmat 13", lambda: synth.RmatGraph(13, 60_000))
o=T("oracle create", lambda: hbo.Dense(g.id_low64(), g.row_ptr, g.src))
T("oracle run", lambda: o.run())
for rep in range(2):
    ctx=T("Context()", lambda: _lib.Context())
    T("load_dense", lambda: ctx.load_dense(g.ids, g.row_ptr, g.src))
    T("begin", lambda: ctx.begin())
    T("step", lambda: ctx.step())
    T("registers", lambda: ctx.registers())
    T("kahan", lambda: ctx.kahan())
    T("run", lambda: ctx.run())
    T("results", lambda: ctx.results())
    T("close", lambda: ctx.close())
