"""Bits of the adaptive c-peptide launches (loss, per-subject SSE, gradients, accepted steps) written to an .npz: run once
with the shipped library and once with an A/B variant (CUDE_ABL=<name> -> tools/abl_so/<name>.so), then compare.
python tools/abl_adaptive_bits.py out.npz [other.npz]"""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
if os.environ.get("CUDE_ABL"):
    from cude import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "abl_so", os.environ["CUDE_ABL"] + ".so")
    _lib.STRICT = False
from cude.engine import Engine  # noqa: E402

out = {}
for arch in ((2, 4, 2), (2, 6, 2), (2, 8, 3), (2, 5, 1)):
    N = 5000
    nn = bench.glorot(arch, 1234)
    eng0, pop = bench.cpep_engine(Engine, arch, 2, N, 777, 0, nn)
    eng0.close()
    eng = Engine("cpep", arch, n_steps=0, n_state=2)
    eng.set_population_cpep(pop["tp"], pop["G"], pop["obs"], pop["age"], pop["t2dm"])
    eng.set_params(nn, pop["beta0"])
    f = eng.forward(want_sse=True, want_traj=True)
    loss, g_nn, g_b = eng.loss_grad()
    key = "x".join(map(str, arch))
    out[key + "_loss"] = np.array([f["loss"], loss])
    out[key + "_sse"], out[key + "_traj"], out[key + "_g_nn"], out[key + "_g_b"] = f["sse"], f["traj"], g_nn, g_b
    out[key + "_dt17"] = eng.adaptive_steps(17)[1]
    eng.close()
for arch in ((4, 3, 5), (4, 3, 3), (4, 6, 2)):
    N = 3000
    tp, data, theta = bench.synthetic_suppression(N, 779)
    nn = bench.glorot(arch, 1234)
    eng = Engine("supp", arch, n_steps=0, lam=0.01)
    eng.set_population_supp(tp, data)
    eng.set_params(nn, theta)
    f = eng.forward(want_sse=True, want_traj=True)
    loss, g_nn, g_b = eng.loss_grad()
    key = "supp" + "x".join(map(str, arch))
    out[key + "_loss"] = np.array([f["loss"], loss])
    out[key + "_sse"], out[key + "_traj"], out[key + "_g_nn"], out[key + "_g_b"] = f["sse"], f["traj"], g_nn, g_b
    out[key + "_dt17"] = eng.adaptive_steps(17)[1]
    eng.close()
np.savez(sys.argv[1], **out)
if len(sys.argv) > 2:
    other = np.load(sys.argv[2])
    for k in out:
        same = np.array_equal(out[k], other[k])
        d = (np.max(np.abs(out[k] - other[k]) / (np.abs(other[k]) + 1e-300))
             if not same and out[k].shape == other[k].shape else 0.0)
        print(f"{k:16s} {'bit-identical' if same else f'DIFFERS (max rel {d:.2e})'}")
