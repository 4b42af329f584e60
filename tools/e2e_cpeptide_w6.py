"""The c-peptide training recipe with the 2->6->6->1 network (the width that runs the layer-1 exponent table) on the
82 subjects of the reference's prepared train set, once with the table and once with direct exponentials
(CUDE_NO_EXPTAB=1): same restarts, same recipe; the distributions of final objectives must coincide.

usage: python tools/e2e_cpeptide_w6.py [K=10]
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "conditional-ude_amd"))
    from cude import api
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ohashi_cude.npz")))
    sel = np.flatnonzero(np.isin(g["subject_no"], g["train_subject_numbers"]))
    net = api.chain(6, 2, "tanh")
    models = [api.CPeptideConditionalUDEModel(g["glucose"][i], g["timepoints"], g["ages"][i], net, g["cpeptide"][i],
                                              g["t2dm"][i]) for i in sel]
    t0 = time.perf_counter()
    sols = api.train(models, g["timepoints"], g["cpeptide"][sel], np.random.default_rng(232705), initial_guesses=5000,
                     selected_initials=K, number_of_iterations_adam=500, number_of_iterations_lbfgs=300)
    dt = time.perf_counter() - t0
    obj = sorted(s.objective for s in sols)
    print(f"{'direct' if os.environ.get('CUDE_NO_EXPTAB') else 'table '}: {len(sols)} runs in {dt:.1f} s; objectives "
          f"{np.round(obj, 4).tolist()}")
else:
    for env in ({}, {"CUDE_NO_EXPTAB": "1"}):
        subprocess.run([sys.executable, __file__, str(K), "child"], env={**os.environ, **env}, check=True)
