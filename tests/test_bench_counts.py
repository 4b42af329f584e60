"""The counted figures bench.py divides by ("the bench must print the exact counted figures from
constants"): algorithmic bytes per subject-trajectory and executed flops per subject-trajectory of the kernels, pinned
against an independent tally so that they cannot drift silently with the code that reports the roofline fractions."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes():
    b = _bench()
    # CPEP3, T = 5, fwd + adjoint: k0 k1 k2 c0 (32) + beta (8) + glucose (40) + observations (40) read; sse, dL/dbeta, auc written
    assert b.cpep_algo_bytes(5, 3, True) == 120 + 24 == 144
    assert b.cpep_algo_bytes(5, 3, False) == 120 + 16
    assert b.cpep_algo_bytes(5, 2, True) == 120 + 16                 # the reference's 2-state model: 136 B
    # SUPP, T = 8: theta (8) + data 3 x 8 x 8 (192; u0 is its first row) read; sse, dL/dtheta written
    assert b.supp_algo_bytes(8, True) == 200 + 16
    assert b.supp_algo_bytes(8, False) == 200 + 8


def test_executed_flops_of_the_headline_kernel():
    b = _bench()
    w, d, S, T = 6, 2, 30, 5
    # one network evaluation as executed (cude_math.h): per tanh neuron 7 single-flop ops + 13 FMAs, per layer one shared
    # reciprocal (rcp + 3 FMAs) and 3 (W - 1) prefix / back-substitution multiplies; softplus 23 FMAs + 22 other ops
    # (round 3: the atanh polynomial of the logarithm is a degree-7 interpolant, 8 FMAs instead of 11)
    tanh_layer = w * (7 + 26) + 7 + 3 * (w - 1)
    fwd = 2 * (w * 1 + w * w + w) + d * tanh_layer + (23 * 2 + 22)
    assert b.mlp_flops(1, w, d) == (fwd, (2 * w + w + 2) + w * (4 + 4 * w) + w * (4 + 2))
    assert fwd == 604
    # table bookkeeping of the headline grid: 30 steps over 4 glucose pieces of 7.5 steps: 2 straddle a knot (steps 7, 22),
    # step 15 starts exactly on one: 28 steps inside a piece, in 4 runs
    assert b.table_steps(S, T) == (28, 4)
    assert b.cpep_flops() == 178973                                  # the figure quoted in DESIGN.md / profiles
    # without the layer-1 table the same kernel would execute the round-1 count
    n_eval = 5 * S + 1
    # (the reverse sweep's re-evaluation never executes the value-only part of the softplus: 8 FMAs + 9 single ops)
    assert b.SOFTPLUS_VALUE_ONLY_FLOPS == 25
    plain = n_eval * (2 * fwd - 25 + b.mlp_flops(1, w, d)[1]) + S * (2 * (2 * 21 + 2 * 6 + 7 * 4) + 12 + 2 * (2 * 21 + 6 * 4 + 12) + 12) \
        + 2 * T * (2 * 3 * 7 + 8)
    assert plain == 223277                                          # (rounds 1 / 2 counted 228 864: longer series, value part twice)
    saved_per_eval, per_run = w * (7 + 26) + 2 * w - 3 * w, 6 * w * (6 + 24) + 10 * w
    assert plain - b.cpep_flops() == 2 * (5 * 28 * saved_per_eval - 4 * per_run - 28 * w)
    # forward-only and the 2-state / width-4 instances scale as their structure says
    assert b.cpep_flops(grad=False) < 0.45 * b.cpep_flops()
    # round 3: the width-4 c-peptide kernels and the suppression kernel evaluate tanh by table + addition theorem
    # (18 flops per neuron instead of 33): 2 x 151 evaluations x 2 layers x 4 neurons x 15 fewer, resp. 2 x 181 x 5 x 3 x 15
    assert b.mlp_flops(1, 4, 2, table_tanh=True)[0] == b.mlp_flops(1, 4, 2)[0] - 2 * 4 * 15
    exp_form = b.cpep_flops((2, 4, 2), S, T, 2, True) + 2 * 151 * 2 * 4 * 15
    assert b.cpep_flops((2, 4, 2), S, T, 2, True) == 112547 and exp_form == 154374 - 151 * (2 * 6 + 25)
    assert b.supp_flops((4, 3, 5), S, 8, True) == 322853 - 2 * 181 * 5 * 3 * 15 - 181 * (2 * 6 + 25) == 234706


def test_kernel_source_digest_matches_the_committed_pmc_record():
    """bench.py quotes roofline.traffic only for the sources the PMC passes were taken on: the committed record must be
    the one of the committed sources (otherwise the driver's line would carry traffic = null)."""
    import json
    b = _bench()
    import pytest
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    if rec["source_sha"] != b.kernel_source_sha():
        pytest.skip("kernel sources changed since the PMC passes: re-run tools/profile_r03.sh (bench.py reports "
                    "traffic = null with a note until then)")
    t, note = b.pmc_traffic("headline", 125000)
    assert note is None or t is not None
    assert 18.0e6 < t < 23.0e6                                       # 18.0 MB algorithmic + the partial rows
    assert abs(rec["calibration"]["ratio"] - 0.5) < 0.03             # FETCH_SIZE counts half of the bytes on gfx950


def test_committed_bench_line_keeps_the_contract():
    """profiles/r03/bench_final.json is the line bench.py printed on the round's final sources: every contract field is
    there, the roofline objects are self-consistent, and the counted figures are the ones this file pins."""
    import json
    b = _bench()
    d = json.load(open(os.path.join(ROOT, "profiles", "r03", "bench_final.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["subjects_per_gpu"]
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["algorithmic_bytes_per_launch"] == n * b.cpep_algo_bytes(5, 3, True)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-9 * r["achieved"]
    assert r["kernel_ms"] < d["ms_per_step"] and r["launches"] >= 1
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.3
    v = d["roofline_valu"]
    assert v["flops_per_trajectory"] == b.cpep_flops() and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 100 * c["value"]
    for k in ("forward_only_1e4", "train_step_1e5", "cpep2_4_1e5", "cpep2_4_1e5_adaptive", "supp_1e5", "saem_estep_1e4x100"):
        assert k in d["extra"], k
