#!/bin/bash
# dev: stamped variants of the bf16 64-row forward (tools/w64_stamp.py).
#   build TAG:"-Dflags" ...   (CPU box)  -> nnop.jl_amd/lib_var<TAG>s/ ;   run TAG ...   (GPU box)
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  make -C nnop.jl_amd/csrc -j8 > /dev/null
  for a in "$@"; do
    tag=${a%%:*}; fl=${a#*:}
    ( rm -rf nnop.jl_amd/build_var${tag}s nnop.jl_amd/lib_var${tag}s; cp -r nnop.jl_amd/build nnop.jl_amd/build_var${tag}s; rm -f nnop.jl_amd/build_var${tag}s/fa_fwd_bf16.o
      make -C nnop.jl_amd/csrc DEV=1 VAR="-DNNOP_W64_STAMP=1 $fl" OUTDIR=../lib_var${tag}s BUILD=../build_var${tag}s 2>&1 | grep -i " error" ) &
  done; wait
else
  CFG=${CFG:-"bf16:64:4096:4:4:4 bf16:128:4096:4:4:4"}
  for tag in "$@"; do echo "== $tag"; NNOP_LIB_PATH=$PWD/nnop.jl_amd/lib_var${tag}s/libnnop_hip.so python tools/w64_stamp.py $CFG 2>/dev/null | sed 's/launches; /\n     /; s/; first entry.*//'; done
fi
