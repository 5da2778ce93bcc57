#!/bin/bash
# HBM bytes per launch of the row-wise operators from PMC counters (separate passes, as the guide prescribes);
# bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950 (FETCH_SIZE tallies 128-B requests at 64 B).
cd /root/repo; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_rows_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_rows_$c -- python3 tools/pmc_rows.py > gpurun_out/pmc_rows_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob(f"/root/repo/gpurun_out/pmc_rows_{c}/*/*counter_collection.csv"))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "nnop" in r["Kernel_Name"] and r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k][c] = sum(v) / len(v)
res = {}
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        res[k] = dict(v, hbm_bytes=int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), read_bytes=int(2 * v["FETCH_SIZE"] * 1024),
                      write_bytes=int(v["WRITE_SIZE"] * 1024))
print(json.dumps(res, indent=1))
PY
tail -1 gpurun_out/pmc_rows_FETCH_SIZE.log
