#!/bin/bash
# usage: run.sh <libpath>  -- runs bench_gemm2 against an alternative build of the library
cd $GRAFT_REPO_ROOT
cp epos_amd/lib/libepos_hip.so /tmp/orig.so
for v in 1 2; do
  cp tools/_abl/libepos_abl$v.so epos_amd/lib/libepos_hip.so
  echo "ABLATE=$v"; python tools/bench_gemm2.py 2>/dev/null | head -3
done
cp /tmp/orig.so epos_amd/lib/libepos_hip.so
echo BASE; python tools/bench_gemm2.py 2>/dev/null | head -3
