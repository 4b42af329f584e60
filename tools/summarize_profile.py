"""Condenses rocprofv3 CSV output (kernel stats + PMC passes, tools/profile_r02.sh) into a short text summary and
profiles-ready pmc_traffic.json (HBM bytes per launch of the dominant kernels, tagged with the digest of the kernel
sources they were measured on)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def kernel_stats(tag):
    print(f"== kernel stats ({tag}: rocprofv3 --kernel-trace --stats)")
    res = {}
    for r in rows(f"{tag}/**/*kernel_stats.csv"):
        print(f"{r.get('Name','')[:86]:86s} calls={r.get('Calls'):>5s} total_ns={r.get('TotalDurationNs'):>12s} "
              f"avg_ns={r.get('AverageNs'):>12s} pct={r.get('Percentage')}")
        res[r.get("Name", "")] = (float(r.get("AverageNs", 0)), int(r.get("Calls", 0)))
    return res


def rocprof_avg(st, kern_sub, tag="stats"):
    """{"rocprof_avg_ms", "rocprof_calls", "rocprof_median_ms"} of the kernel whose name contains kern_sub: the
    profiler's own average over every launch of the run (kernel_stats() result st) and the median of the per-launch
    durations of its kernel trace (the mean includes the clock-ramp launches at the start of the process, the median
    does not).  bench.py prints them beside its HIP-event figure."""
    res = {}
    for name, (avg, calls) in st.items():
        if kern_sub in name:
            res = {"rocprof_avg_ms": avg * 1e-6, "rocprof_calls": calls}
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows(f"{tag}/**/*kernel_trace.csv")
               if kern_sub in r.get("Kernel_Name", ""))
    if d:
        res["rocprof_median_ms"] = 0.5e-6 * (d[(len(d) - 1) // 2] + d[len(d) // 2])
    return res


def counters(tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(f"{tag}/**/*counter_collection.csv"):
        acc[r.get("Kernel_Name", "")][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
    return acc


def mean_ctr(tag, kern_sub, ctr, skip=0):
    """mean over the launches of the kernel whose name contains kern_sub (the first `skip` launches dropped)."""
    v = [float(r["Counter_Value"]) for r in rows(f"{tag}/**/*counter_collection.csv")
         if kern_sub in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr]
    v = v[skip:] if len(v) > skip else v
    return (sum(v) / len(v), len(v)) if v else (None, 0)


stats = kernel_stats("stats")
wait_tags = sorted(os.path.basename(p) for p in glob.glob(os.path.join(out, "pmc_w_*")))
for tag in ["pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"] + wait_tags:
    print(f"== {tag}")
    for k, d in counters(tag).items():
        if "cude::" not in k:
            continue
        for c, v in sorted(d.items()):
            print(f"{k[:70]:70s} {c:24s} n={len(v)} mean={sum(v)/len(v):.6g}")

# HBM traffic per launch, corrected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE (KB) counts 64 B per 128-B
# request on gfx950 -> x2, calibrated below on prepare_cpep_kernel whose byte count is known exactly in THIS access
# pattern (8 B per lane, subject-major); WRITE_SIZE (KB) is exact.
import bench  # noqa: E402

rec = {"source_sha": bench.kernel_source_sha(), "fetch_correction": 2.0, "kernels": {}}
n = None
try:
    n = json.loads(open(os.path.join(out, "bench_stats.json")).read().strip().splitlines()[-1])["config"]["subjects_per_gpu"]
except Exception as e:
    print("no bench line:", e)
GRAD = "true, 0>(cude::CpepArgs)"      # cpep_kernel<Net, NS, GRAD = true, KEEP = 0>
f, nf = mean_ctr("pmc_fetch", GRAD, "FETCH_SIZE")
w, nw = mean_ctr("pmc_write", GRAD, "WRITE_SIZE")
pf, _ = mean_ctr("pmc_fetch", "prepare_cpep_kernel", "FETCH_SIZE")
if f is not None and w is not None:
    rec["kernels"]["headline"] = {"kernel": "cpep_kernel<Mlp<2,6,2,1>,3,grad>", "subjects_per_gpu": n, "launches": nf,
                                  "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "source_sha": rec["source_sha"],
                                  "hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024}
    # SQ counters of the same kernel (per launch, summed over the chip): issue-slot utilisation in bench.py
    sq = {}
    for ctr in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
                "SQ_ACTIVE_INST_VALU"):
        v, nv = mean_ctr("pmc_sq", GRAD, ctr)
        if v is not None:
            sq[ctr] = v
    if sq:
        rec["kernels"]["headline"]["sq"] = sq
    rec["kernels"]["headline"].update(rocprof_avg(stats, GRAD))
    if pf and n:
        rec["calibration"] = {"kernel": "prepare_cpep_kernel", "known_read_bytes": n * (7 * 8 + 1), "FETCH_SIZE_KB": pf,
                              "ratio": pf * 1024 / (n * (7 * 8 + 1))}
for mode in ("stage_inputs", "steps"):
    sst = kernel_stats(f"supp_{mode}_stats")
    f, nf = mean_ctr(f"supp_{mode}_fetch", "supp_kernel", "FETCH_SIZE", skip=3)
    w, nw = mean_ctr(f"supp_{mode}_write", "supp_kernel", "WRITE_SIZE", skip=3)
    if f is not None and w is not None:
        rec["kernels"][f"supp_{mode}"] = {"kernel": "supp_kernel<3,5,grad> (+ forward-only launches of the same run)",
                                         "subjects_per_gpu": 100000, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                                         "source_sha": rec["source_sha"],
                                         "hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024}
        if mode == "stage_inputs":
            sq = {}
            for ctr in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
                        "SQ_ACTIVE_INST_VALU"):
                v, nv = mean_ctr("supp_sq", "supp_kernel<3, 5, true, false, false", ctr, skip=3)
                if v is not None:
                    sq[ctr] = v
            if sq:
                rec["kernels"][f"supp_{mode}"]["sq"] = sq
        rec["kernels"][f"supp_{mode}"].update(rocprof_avg(sst, "supp_kernel<3, 5, true, false, false", f"supp_{mode}_stats"))
    try:
        print(open(os.path.join(out, f"supp_{mode}.log")).read().strip())
    except Exception:
        pass
print("== supp_sq")
for k, d in counters("supp_sq").items():
    if "supp_kernel" in k:
        for c, v in sorted(d.items()):
            print(f"{k[:70]:70s} {c:24s} n={len(v)} mean={sum(v)/len(v):.6g}")
try:
    print(open(os.path.join(out, "occupancy.txt")).read().strip())
except Exception:
    pass
ast = kernel_stats("adaptive_stats")
AGRAD = "> >, true>(cude::CpepAd<"            # adaptive_unrolled_kernel<CpepAd<...>, GRAD = true>(CpepAd<...>::Args)
f, nf = mean_ctr("adaptive_fetch", AGRAD, "FETCH_SIZE", skip=3)
w, nw = mean_ctr("adaptive_write", AGRAD, "WRITE_SIZE", skip=3)
if f is not None and w is not None:
    rec["kernels"]["adaptive_grad"] = {"kernel": "adaptive_unrolled_kernel<CpepAd<Mlp<2,4,2,1>>,grad>", "subjects_per_gpu": 100000,
                                       "launches": nf, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                                       "source_sha": rec["source_sha"],
                                       "hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024,
                                       "note": "tape: 8 B (dt) per accepted step and subject written and read back (20 steps "
                                               "typical) + 5 saved outputs each way"}
    rec["kernels"]["adaptive_grad"].update(rocprof_avg(ast, AGRAD, "adaptive_stats"))
    sq = {}
    for ctr in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
                "SQ_ACTIVE_INST_VALU"):
        v, nv = mean_ctr("adaptive_sq", AGRAD, ctr, skip=3)
        if v is not None:
            sq[ctr] = v
    if sq:
        rec["kernels"]["adaptive_grad"]["sq"] = sq
try:
    print(open(os.path.join(out, "adaptive.log")).read().strip())
except Exception:
    pass
asst = kernel_stats("adaptive_supp_stats")
ASGRAD = "adaptive_unrolled_supp_kernel<cude::SuppAd<3, 5, 0, 0>, true>"
f, nf = mean_ctr("adaptive_supp_fetch", ASGRAD, "FETCH_SIZE", skip=3)
w, nw = mean_ctr("adaptive_supp_write", ASGRAD, "WRITE_SIZE", skip=3)
if f is not None and w is not None:
    rec["kernels"]["adaptive_supp_grad"] = {"kernel": "adaptive_unrolled_supp_kernel<SuppAd<3,5>,grad>", "subjects_per_gpu": 100000,
                                            "launches": nf, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                                            "source_sha": rec["source_sha"],
                                            "hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024,
                                            "note": "tape: 136 B per accepted step and subject written and read back (24 steps "
                                                    "typical) + 16 B per observation each way"}
    rec["kernels"]["adaptive_supp_grad"].update(rocprof_avg(asst, ASGRAD, "adaptive_supp_stats"))
try:
    print(open(os.path.join(out, "adaptive_supp.log")).read().strip())
except Exception:
    pass
# round 5: the suppression kernel at one full fill, the restart trainer, the speculative E-step, the small-population gradient
for tag in ("supp_fill", "train", "estep", "smallgrad"):
    kernel_stats(f"{tag}_stats")
    try:
        print("\n".join(l for l in open(os.path.join(out, f"{tag}.log")).read().strip().splitlines() if "amdgpu.ids" not in l))
    except Exception:
        pass
json.dump(rec, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("== pmc_traffic.json", json.dumps(rec, indent=1))
