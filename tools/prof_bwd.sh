#!/bin/bash
# Per-kernel averages (rocprofv3 --kernel-trace --stats) of the bench's fwd+bwd leg for one workload, and an SQ counter pass of the
# backward kernels (MFMA busy, LDS bank conflicts).  usage: tools/prof_bwd.sh [config]   -> gpurun_out/prof_bwd_<cfg>/
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
CFG=${1:-c2}
OUT=gpurun_out/prof_bwd_$CFG; rm -rf $OUT; mkdir -p $OUT
STEPS=40; [ "$CFG" != "c2" ] && STEPS=8
BENCH="python3 bench.py --config $CFG --steps $STEPS --warmup 4 --settle-ms 200 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc -- $BENCH > /dev/null 2> $OUT/pmc.err
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
st = glob.glob(out + "/trace/*/*kernel_stats.csv")
rows = [r for r in csv.DictReader(open(st[0])) if "nnop" in r["Name"]] if st else []
with open(out + "/kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"]); w.writeheader(); w.writerows(rows)
for r in rows: print(f'{r["Name"][:88]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:10.1f} us')
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(out + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_bwd" in r["Kernel_Name"]: acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/pmc/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_bwd" in r["Kernel_Name"]: dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
summ = {}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    e = {"counters_mean": m}
    if dur[k] and "GRBM_GUI_ACTIVE" in m:
        d = sorted(dur[k])[len(dur[k]) // 2]
        clk = m["GRBM_GUI_ACTIVE"] / 8 / d
        e.update(kernel_ns_median_profiled=d, clock_ghz_est=round(clk, 3),
                 mfma_busy_ratio=round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * clk * d), 4),
                 lds_conflict_per_active=round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4))
    summ[k[:70]] = e
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
for k, e in summ.items(): print(k, {x: e[x] for x in e if x != "counters_mean"})
PY
