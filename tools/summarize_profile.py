"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv, glob, os, sys, collections
out = sys.argv[1]

def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r

print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for r in rows("stats/**/*kernel_stats.csv"):
    print(f"{r.get('Name','')[:70]:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", None)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(f"{tag}/**/*counter_collection.csv"):
        acc[r.get("Kernel_Name", "")[:60]][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
    print(f"== {tag}")
    for k, d in acc.items():
        for c, v in d.items():
            print(f"{k:60s} {c:22s} n={len(v)} mean={sum(v)/len(v):.6g}")

# HBM traffic of the dominant kernel per launch, corrected as MI355X_MICROARCH.md "HBM" prescribes:
# FETCH_SIZE (KB) counts 64 B per 128-B request on gfx950 -> x2 (calibrated below on prepare_cpep_kernel,
# whose byte count is known exactly); WRITE_SIZE (KB) is exact.
import json
def mean_ctr(tag, kern_sub, ctr):
    v = [float(r["Counter_Value"]) for r in rows(f"{tag}/**/*counter_collection.csv")
         if kern_sub in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr]
    return sum(v) / len(v) if v else None
f = mean_ctr("pmc_fetch", "true>(cude::CpepArgs)", "FETCH_SIZE")
w = mean_ctr("pmc_write", "true>(cude::CpepArgs)", "WRITE_SIZE")
pf = mean_ctr("pmc_fetch", "prepare_cpep_kernel", "FETCH_SIZE")
if f is not None and w is not None:
    n = None
    try:
        n = json.load(open(os.path.join(out, "bench_stats.json")))["config"]["subjects_per_gpu"]
    except Exception:
        pass
    rec = {"subjects_per_gpu": n, "kernel": "cpep_kernel<2,6,2,3,grad>", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "fetch_correction": 2.0, "hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024}
    if pf and n:
        rec["calibration"] = {"kernel": "prepare_cpep_kernel", "known_read_bytes": n * (7 * 8 + 1),
                              "FETCH_SIZE_KB": pf, "ratio": pf * 1024 / (n * (7 * 8 + 1))}
    json.dump(rec, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("== pmc_traffic.json", rec)
