#!/bin/bash
# usage: tools/prof_rows.sh <softmax|norms>   -> per-kernel average durations via rocprofv3 --kernel-trace --stats
cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/pk; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk -- python3 tools/perf_rows.py "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/root/repo/gpurun_out/pk/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "nnop" in r["Name"]:
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f}')
PY
