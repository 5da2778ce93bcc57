#!/bin/bash
# dev: rocprofv3 per-kernel averages of tools/perf_pair_modes.py
cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/pkm; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pkm -- python3 tools/perf_pair_modes.py "$@" > gpurun_out/pkm.log 2>&1
cat gpurun_out/pkm.log | grep -v amdgpu.ids
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/root/repo/gpurun_out/pkm/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "nnop" in r["Name"] or "fill" in r["Name"].lower() or "memset" in r["Name"].lower():
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
