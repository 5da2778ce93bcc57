#!/bin/bash
# dev: wall time of the GPU suite per test file (one pytest process each; the import cost of ~3 s per process is included)
cd "$(dirname "$0")/.."
for f in tests/test_*.py; do
  s=$(date +%s.%N); r=$(timeout -k 10 600 python -m pytest $f -q -m gpu 2>&1 | tail -1); e=$(date +%s.%N)
  printf "%-44s %6.1f s  %s\n" $f $(echo "$e - $s" | bc) "$r"
done
