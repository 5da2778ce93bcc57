#!/bin/bash
# dev: pullback time vs the cap on partial rows (= persistent workgroups)
cd /root/repo
for cap in 256 512 1024 2048; do
  echo "cap $cap:"; NNOP_NORM_BWD_CAP=$cap python tools/perf_rows.py norms 2>/dev/null | grep grad_ | grep -o '"op": "[a-z_]*", "shape": "[a-z0-9 ]*", "dtype": "[a-z0-9]*", "us": [0-9.]*' | sed 's/"op": //; s/"shape": //; s/"dtype": //; s/"us": //' | paste - - - - 
done
