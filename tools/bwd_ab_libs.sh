#!/bin/bash
# dev: A/B of library builds for the backward on ONE box: tools/bwd_ab_libs.sh "TAG TAG" shape ...  (TAG: lib_var<TAG> directory or `lib`)
cd "$(dirname "$0")/.."
tags=$1; shift
for r in 1 2; do for t in $tags; do
  d=nnop.jl_amd/lib_var$t; [ "$t" = lib ] && d=nnop.jl_amd/lib
  printf "%-6s " $t; NNOP_LIB_PATH=$PWD/$d/libnnop_hip.so python tools/bwd_ab.py default -- "$@" 2>/dev/null | tr '\n' '|'; echo
done; done
