#!/bin/bash
# Evidence for the backward of one bench.py workload: per-kernel averages (rocprofv3 --kernel-trace --stats) of the bench's fwd+bwd
# legs, then SEPARATE counter passes of the same command (each with --kernel-trace for the per-dispatch rows, never with
# --sys-trace / hip / hsa trace domains; at most 4 counters per pass so that no pass is silently split): MFMA busy + clock,
# VALU / LDS-bank-conflict counters, and HBM bytes (FETCH_SIZE / WRITE_SIZE with the gfx950 correction of MI355X_MICROARCH.md).
# usage: tools/prof_bwd.sh [config]   -> gpurun_out/prof_bwd_<cfg>/{kernel_stats.csv,pmc_summary.json}; copy into profiles/rNN/
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
CFG=${1:-c2}
OUT=gpurun_out/prof_bwd_$CFG; rm -rf $OUT; mkdir -p $OUT
STEPS=40; [ "$CFG" != "c2" ] && STEPS=8
BENCH="python3 bench.py --config $CFG --steps $STEPS --warmup 4 --settle-ms 200 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_traced.json 2> $OUT/trace.err
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH > /dev/null 2> $OUT/pmc$i.err
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
WANT = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT",
        "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE"]
st = glob.glob(out + "/trace/*/*kernel_stats.csv")
rows = [r for r in csv.DictReader(open(st[0])) if "nnop" in r["Name"]] if st else []
with open(out + "/kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"]); w.writeheader(); w.writerows(rows)
for r in rows: print(f'{r["Name"][:88]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:10.1f} us')
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_bwd" in r["Kernel_Name"]: acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/pmc1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_bwd" in r["Kernel_Name"]: dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
summ, missing = {}, set()
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    missing |= {c for c in WANT if c not in m}
    e = {"counters_mean": m}
    if dur[k] and "GRBM_GUI_ACTIVE" in m:
        d = sorted(dur[k])[len(dur[k]) // 2]
        clk = m["GRBM_GUI_ACTIVE"] / 8 / d
        e.update(kernel_ns_median_profiled=d, clock_ghz_est=round(clk, 3),
                 mfma_busy_ratio=round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * clk * d), 4),
                 lds_conflict_per_active=round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)      # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    summ[k[:90]] = e
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
for k, e in summ.items(): print(k, {x: e[x] for x in e if x != "counters_mean"})
if missing:
    print("MISSING COUNTERS (pass silently dropped?):", sorted(missing)); sys.exit(1)
PY
