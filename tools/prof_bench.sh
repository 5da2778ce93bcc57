#!/bin/bash
# Evidence for one bench.py workload (default c2): rocprofv3 --kernel-trace --stats of the bench command, then SEPARATE --pmc
# passes (each with --kernel-trace for the per-dispatch rows, never with --sys-trace / hip / hsa trace domains): HBM bytes (FETCH_SIZE / WRITE_SIZE, gfx950 correction applied in the summary) and
# the MFMA-busy ratio SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles).  Output: gpurun_out/prof_<cfg>/{kernel_stats.csv,
# pmc_summary.json}; copy what should be judged into profiles/rNN/.
# usage: tools/prof_bench.sh [config] [extra bench.py flags]
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
CFG=${1:-c2}; shift || true
OUT=gpurun_out/prof_$CFG; rm -rf $OUT; mkdir -p $OUT
STEPS=100; [ "$CFG" != "c2" ] && STEPS=10
BENCH="python3 bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --no-bwd $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_traced.json 2> $OUT/trace.err
i=0
# (at most 4 counters per pass, so that no pass is silently split into several)
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1)); rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH > /dev/null 2> $OUT/pmc$i.err
done
python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, json, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/trace/*/*kernel_stats.csv")
if st:
    rows = [r for r in csv.DictReader(open(st[0])) if "nnop" in r["Name"]]
    with open(out + "/kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"]); w.writeheader(); w.writerows(rows)
    for r in rows:
        print(f'{r["Name"][9:70]:62s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:10.2f} us')
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_fwd" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/pmc3/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_fwd" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
summ = {}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    e = {"counters_mean": m, "launches": {c: len(x) for c, x in v.items()}}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
        e["correction"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests at 64 B; separate --pmc passes)"
    if dur.get(k) and "GRBM_GUI_ACTIVE" in m:
        d = sorted(dur[k]); med_ns = d[len(d) // 2]
        clk = m["GRBM_GUI_ACTIVE"] / 8 / med_ns          # GHz (sum over 8 XCDs)
        e["kernel_ns_median_profiled"] = med_ns
        e["clock_ghz_est"] = round(clk, 3)
        # matrix-pipe busy: MFMA busy cycles summed over all SIMDs / (1024 SIMDs x kernel cycles at the estimated clock)
        e["mfma_busy_ratio"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * med_ns * clk), 4)
        e["mfma_busy_ratio_at_2.4GHz"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * med_ns * 2.4), 4)
    summ[k] = e
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps({k[:60]: {a: b for a, b in v.items() if a != "counters_mean"} for k, v in summ.items()}, indent=1))
PY
