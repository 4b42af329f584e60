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


def test_executed_operations_of_the_kernels():
    """Round 4: the tally keeps three classes apart -- FMAs, other floating-point arithmetic (add, mul, rcp), and VALU
    instructions that do no floating-point work (clamps, rounding / conversion, ldexp, selects, sign transfer).  Only the
    first two form `flops_per_trajectory`; rounds 1-3 had counted every single-slot instruction as one flop (178 973 for
    the headline kernel, of which 10 173 were not floating-point operations)."""
    b = _bench()
    w, d, S, T = 6, 2, 30, 5
    O = b.Ops
    # the pieces (cude_math.h): reciprocal, exponential, one tanh neuron in either form, softplus
    assert (b.RCP.fma, b.RCP.fp1, b.RCP.other) == (3, 1, 0)
    assert (b.EXP2X.fma, b.EXP2X.fp1, b.EXP2X.other) == (12, 1, 3)
    assert b.TANH_EXP_NEURON.slots == 20 and b.TANH_EXP_NEURON.flops == 2 * 13 + 2       # 13 FMAs, mul, add; 5 others
    assert b.TANH_TAB_NEURON.slots == 15 and b.TANH_TAB_NEURON.flops == 2 * 3 + 9
    assert (b.SOFTPLUS.fma, b.SOFTPLUS.fp1, b.SOFTPLUS.other) == (23, 17, 15)
    assert (b.SOFTPLUS_VALUE_ONLY.fma, b.SOFTPLUS_VALUE_ONLY.fp1, b.SOFTPLUS_VALUE_ONLY.other) == (8, 8, 5)
    # one evaluation of the headline network 2-6-6-1 (one varying input): 48 FMAs of the linear maps, two tanh layers
    # with one reciprocal each, the softplus; backward: output unit, one hidden layer, the first layer
    fwd, bwd = b.mlp_ops(1, w, d)
    lin = w * 1 + w * w + w
    assert fwd.fma == lin + d * (w * 13 + 3) + 23 == 233
    assert fwd.fp1 == d * (w * 2 + 1 + 3 * (w - 1)) + 17 == 73 and fwd.other == d * w * 5 + 15 == 75
    assert (bwd.fma, bwd.fp1, bwd.other) == (w + (w + 2 * w * w) + 2 * w, (w + 2) + (2 * w + w) + 2 * w, 0) == (96, 38, 0)
    assert b.mlp_flops(1, w, d) == (fwd.flops, bwd.flops) == (539, 230)
    # table bookkeeping of the headline grid: 30 steps over 4 glucose pieces of 7.5 steps: 2 straddle a knot (steps 7, 22),
    # step 15 starts exactly on one: 28 steps inside a piece, in 4 runs
    assert b.table_steps(S, T) == (28, 4)
    head = b.cpep_ops()
    assert (head.fma, head.fp1, head.other) == (70460, 27880, 17959)
    assert b.cpep_flops() == head.flops == 168800                    # the figure quoted in DESIGN.md / profiles
    assert head.slots == 116299                                      # structural; SQ_INSTS_VALU / SQ_WAVES measures ~128.8 k
    # without the layer-1 table every evaluation pays its W exponentials and first-layer FMAs
    n_eval = 5 * S + 1
    plain = n_eval * (fwd + fwd - b.SOFTPLUS_VALUE_ONLY + bwd) \
        + S * (O(fma=2 * 21 + 2 * 6 + 7 * 4 + 6) + O(fma=2 * 21 + 6 * 4 + 12, fp1=12)) + 2 * T * O(fma=3 * 7, fp1=8)
    saved = w * (b.EXP2X + O(fp1=1, other=1) + O(fma=1)) - w * O(fp1=2, other=1)
    per_run = 6 * w * b.EXP2X + w * O(fp1=5, other=5)
    diff = plain - head
    want = 2 * (5 * 28 * saved - 4 * per_run - 28 * O(fp1=w))
    assert (diff.fma, diff.fp1, diff.other) == (want.fma, want.fp1, want.other)
    assert b.cpep_flops(grad=False) < 0.45 * b.cpep_flops()
    # the width-4 c-peptide kernels and the suppression kernel evaluate tanh by table + addition theorem
    assert b.mlp_ops(1, 4, 2, table_tanh=True)[0].flops == b.mlp_ops(1, 4, 2)[0].flops - 2 * 4 * (28 - 15)
    assert b.cpep_flops((2, 4, 2), S, T, 2, True) == 104544
    # suppression kernel, round 4: state 1 is a table lookup (one multiply per evaluation), the Runge-Kutta algebra and
    # the adjoint are those of two states, the first layer's input derivative is formed for two inputs
    supp = b.supp_ops((4, 3, 5), S, 8, True)
    assert (supp.fma, supp.fp1, supp.other) == (69026, 76478, 20815) and b.supp_flops((4, 3, 5), S, 8, True) == 214530
    f3, b3 = b.mlp_ops(3, 3, 5, 2, table_tanh=True)
    assert f3.slots == 378 and (b3.fma, b3.fp1) == (105, 51)        # (the forward loop body of the ISA: 381 VALU instructions)


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
    """profiles/r04/bench_final.json is the line bench.py printed on the round's final sources: every contract field is
    there, the roofline objects are self-consistent, and the counted figures are the ones this file pins."""
    import json
    import pytest
    b = _bench()
    path = os.path.join(ROOT, "profiles", "r04", "bench_final.json")
    if not os.path.exists(path):
        pytest.skip("no round-4 bench line committed yet")
    d = json.load(open(path))
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
    # the profiler's own duration of the same kernel on the same sources (committed rocprofv3 pass): its median agrees
    # with the HIP-event figure of the line (its mean includes the clock-ramp launches of a fresh process)
    assert abs(r["kernel_ms_rocprof"]["median_ms"] - r["kernel_ms"]) <= 0.02 * r["kernel_ms"]
    assert r["kernel_ms_rocprof"]["launches"] >= 100
    v = d["roofline_valu"]
    assert v["flops_per_trajectory"] == b.cpep_flops() and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-12
    assert v["counted"]["valu_slots_per_trajectory"] == b.cpep_ops().slots
    assert 2 * v["counted"]["fma"] + v["counted"]["other_fp_ops"] == v["flops_per_trajectory"]
    if v["valu_slot_utilisation"] is not None:
        assert 0.5 < v["valu_slot_utilisation"] < 1.0 and v["valu_instructions_per_wave"] >= v["counted"]["valu_slots_per_trajectory"]
    assert d["launch_mode"].startswith("plain+events")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 100 * c["value"]
    for k in ("forward_only_1e4", "train_step_1e5", "cpep2_4_1e5", "cpep2_4_1e5_adaptive", "supp_1e5", "supp_1e5_adaptive",
              "saem_estep_1e4x100"):
        assert k in d["extra"], k
