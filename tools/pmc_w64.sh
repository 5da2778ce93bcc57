#!/bin/bash
# dev: SQ / GRBM counters of the forward kernels that tools/w64_check.py launches for ONE shape (old form and 64-row form
# side by side), separate --pmc passes (never combined with trace domains); output: gpurun_out/pmc_w64/summary.txt
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
SHAPE=${1:-bf16:128:4096:16:16:4:plain}
OUT=gpurun_out/pmc_w64; mkdir -p $OUT
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" \
         "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS"; do
  i=$((i+1)); rm -rf $OUT/p$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/w64_check.py $SHAPE > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_fwd" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_fwd" in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:70]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open(out + "/summary.txt", "w") as fo:
    for k, v in sorted(acc.items()):
        d = sorted(dur[k]); med = d[len(d) // 2] if d else 0
        line = k + f"\n   kernel_ns_median={med:.0f} n={len(d)}\n   " + "  ".join(f"{c}={sum(x)/len(x):.5g}" for c, x in sorted(v.items()))
        if "GRBM_GUI_ACTIVE" in v and med:
            line += f"\n   clock_GHz~{sum(v['GRBM_GUI_ACTIVE'])/len(v['GRBM_GUI_ACTIVE'])/8/med:.3f}"
        print(line); fo.write(line + "\n")
PY
