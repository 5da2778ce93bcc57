#!/bin/bash
# dev: HBM / L2 traffic of the pair-bias kernels (tools/perf_pair_modes.py) from separate rocprofv3 --pmc passes; set NNOP_LIB_PATH for another build.
# TCC_EA0_RDREQ_sum ~ 64-B read requests L2 -> fabric, FETCH_SIZE in KiB (gfx950: counts 128-B requests at 64 B -> x2)
cd /root/repo; export TMPDIR=/tmp
TAG=${1:-cur}
for c in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_'); rm -rf gpurun_out/pmcp_${TAG}_$tag
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcp_${TAG}_$tag -- python3 tools/perf_pair_modes.py bf16 > /dev/null 2>&1
done
python3 - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/root/repo/gpurun_out/pmcp_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "nnop" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
PAIR = 4 * 2048 * 2048 * 4 * 2
for k, v in sorted(acc.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    hbm = (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024
    print(f"{k}\n   " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(m.items())) + f"   HBM bytes {hbm/1e6:.1f} MB = {hbm/PAIR:.2f} x |pair|")
PY
