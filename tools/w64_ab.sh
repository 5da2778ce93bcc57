#!/bin/bash
# dev: A/B of library builds on ONE box: tools/w64_ab.sh "TAG TAG ..." shape ...   (TAG: lib_var<TAG> directory, or `lib` for the release)
# alternates the libraries three times (box-to-box and minute-to-minute clock differences exceed most kernel changes)
cd "$(dirname "$0")/.."
tags=$1; shift
for r in 1 2 3; do for t in $tags; do
  d=nnop.jl_amd/lib_var$t; [ "$t" = lib ] && d=nnop.jl_amd/lib
  printf "%-8s " $t; NNOP_LIB_PATH=$PWD/$d/libnnop_hip.so python tools/w64_time.py "$@" 2>/dev/null
done; done
