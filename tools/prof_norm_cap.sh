#!/bin/bash
# dev: rocprofv3 kernel times of the norm pullback kernels (main + fold) for several caps on the partial rows
cd /root/repo; export TMPDIR=/tmp
for cap in 256 512 1024; do
  rm -rf gpurun_out/pk; NNOP_NORM_BWD_CAP=$cap rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk -- python3 tools/pmc_rows.py > /dev/null 2>&1
  echo "cap $cap"
  python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/root/repo/gpurun_out/pk/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "norm_bwd" in r["Name"] or "norm_fold" in r["Name"]:
        print(f'  {r["Name"][12:70]:60s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"])/1e3:8.1f} us min {float(r["MinNs"])/1e3:8.1f}')
PY
done
