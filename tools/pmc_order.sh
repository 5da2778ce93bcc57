cd /root/repo; export TMPDIR=/tmp
for asc in 0 1; do for c in c5 c3; do
  out=gpurun_out/order_${c}_$asc; rm -rf $out; mkdir -p $out
  NNOP_FWD_PERSIST_ASC=$asc rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-bwd > /dev/null 2> $out/err
  NNOP_FWD_PERSIST_ASC=$asc rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-bwd > /dev/null 2>> $out/err
  python3 - $out $c $asc <<'PY'
import csv, glob, sys
out, c, asc = sys.argv[1:]
v = {}
for k in ("f", "w"):
    xs = []
    for f in glob.glob(f"{out}/{k}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fa_fwd" in r["Kernel_Name"]: xs.append(float(r["Counter_Value"]))
    v[k] = sum(xs) / max(len(xs), 1)
print(f"{c} asc={asc}: FETCH_SIZE {v['f']:.0f} KiB WRITE_SIZE {v['w']:.0f} KiB -> HBM bytes per launch {(2 * v['f'] + v['w']) * 1024 / 1e9:.3f} GB", flush=True)
PY
done; done
