#!/bin/bash
# usage: tools/prof_kernels.sh <perf.py cfg>   -> per-kernel average durations via rocprofv3 --kernel-trace --stats
cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/pk; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk -- python3 tools/perf.py "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/root/repo/gpurun_out/pk/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "nnop" in r["Name"]:
        print(f'{r["Name"][9:60]:52s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
