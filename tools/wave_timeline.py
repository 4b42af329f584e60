"""Per-wave timeline of one gradient launch (development aid).  Needs a library built with -DCUDE_WAVE_TIMING
(tools/abl_so/wt.so): every wave records wall_clock64() at its start, after the forward sweep and at its end, plus
its HW_ID / XCC_ID.  Prints the dispatch ramp (spread of start times), the wave lifetimes and the per-XCD picture.

usage: python tools/wave_timeline.py [N=125000] [variant=wt]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cude_oracle as o  # noqa: E402
from cude import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
variant = sys.argv[2] if len(sys.argv) > 2 else "wt"
_lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", variant + ".so")
from cude.engine import Engine  # noqa: E402

arch = (2, 6, 2)
tp, G, cp, age, t2, bt, rng = o.synthetic_cpep_population(N)
eng = Engine("cpep", arch, n_steps=30, n_state=3)
eng.set_population_cpep(tp, G, cp, age, t2)
eng.set_params(o.glorot_params(arch, 1), bt)
eng.adam_init(1e-2)
for _ in range(5):
    eng.adam_step(want_loss=False)
eng.set_kernel_timing(True)
eng.adam_step(want_loss=False)
ms, _ = eng.kernel_time_ms()
nw = (N + 63) // 64
buf = np.zeros((nw, 4), dtype=np.int64)
lib = _lib.load()
lib.cude_debug_wave_timing.restype = C.c_int32
lib.cude_debug_wave_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.cude_debug_wave_timing(eng._h, buf.ctypes.data_as(C.c_void_p), nw) == 0
t0, t1, t2e = (buf[:, k].astype(np.float64) * 0.01 for k in range(3))      # 100 MHz clock -> microseconds
base = t0.min()
start, mid, end = t0 - base, t1 - base, t2e - base
hw = buf[:, 3] & 0xFFFFFFFF
xcc = (buf[:, 3] >> 32) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
print(f"N={N} waves={nw} kernel (events) {ms * 1e3:.1f} us; first start -> last end {end.max():.1f} us")
print(f"start times: median {np.median(start):.1f} us, 90% {np.quantile(start, 0.9):.1f}, max {start.max():.1f}")
life = end - start
print(f"wave lifetime: min {life.min():.1f} median {np.median(life):.1f} max {life.max():.1f} us; "
      f"forward part median {np.median(mid - start):.1f} us")
print(f"end times: min {end.min():.1f} median {np.median(end):.1f} max {end.max():.1f} us")
order = np.argsort(start)
print("start time of the k-th dispatched wave:", {int(k): round(float(start[order[k]]), 1)
                                                    for k in (0, 255, 511, 1023, 1535, nw - 1) if k < nw})
print("block index of the 8 latest starters:", order[-8:].tolist())
import collections  # noqa: E402
simd_key = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
cu_key = [k[:4] for k in simd_key]
per_simd, per_cu = collections.Counter(simd_key), collections.Counter(cu_key)
ns = np.array([per_simd[k] for k in simd_key])
nc = np.array([per_cu[k] for k in cu_key])
print("lifetime quantiles (us) 10/50/90/99%:", np.round(np.quantile(life, [0.1, 0.5, 0.9, 0.99]), 1).tolist(),
      " waves slower than 600 us:", int((life > 600).sum()))
for k in sorted(set(ns.tolist())):
    m = ns == k
    print(f"  {k} wave(s) on the SIMD: {m.sum()} waves, lifetime median {np.median(life[m]):.1f} max {life[m].max():.1f}")
for k in sorted(set(nc.tolist())):
    m = nc == k
    print(f"  {k} wave(s) on the CU: {m.sum()} waves, lifetime median {np.median(life[m]):.1f} max {life[m].max():.1f}")
slow = life > 600
print("slow waves: waves/SIMD", collections.Counter(ns[slow].tolist()), "waves/CU", collections.Counter(nc[slow].tolist()),
      "block idx range", int(np.flatnonzero(slow).min()) if slow.any() else None,
      int(np.flatnonzero(slow).max()) if slow.any() else None)
print("distinct SIMDs used:", len(per_simd), "distinct CUs:", len(per_cu))
for x in range(8):
    m = xcc == x
    if m.any():
        slots = len(set(zip(se[m].tolist(), sh[m].tolist(), cu[m].tolist(), simd[m].tolist())))
        print(f"  XCC {x}: {m.sum()} waves on {slots} SIMDs, start max {start[m].max():.1f}, lifetime median "
              f"{np.median(life[m]):.1f} max {life[m].max():.1f}, end max {end[m].max():.1f}")
