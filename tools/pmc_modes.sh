#!/bin/bash
# dev: dynamic instruction / cycle counters of the forward kernels that tools/perf_modes.py launches (one shape)
cd /root/repo; export TMPDIR=/tmp
SHAPE=${1:-bf16:64:4096:16:4}
for c in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  tag=$(echo $c | tr ' ' '_'); rm -rf gpurun_out/pmc_$tag
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/perf_modes.py $SHAPE > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/root/repo/gpurun_out/pmc_SQ_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_fwd" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:95]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k)
    print("   " + "  ".join(f"{c}={sum(x)/len(x):.4g}" for c, x in sorted(v.items())))
PY
