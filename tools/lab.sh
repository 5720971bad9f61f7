#!/bin/bash
# Build libopenvis_hip.so + build/gemm_lab here (hipcc cross-compiles) and run the lab on a GPU box.
#   tools/lab.sh <out name> <gemm_lab args...>      ->  gpurun_out/lab/<out name>.txt
set -e
cd /root/repo
make -s -j8 -C openvis_amd/csrc 2>&1 | grep -E "error|warning" || true
make -s -C openvis_amd/csrc lab
out=$1; shift
/usr/local/graft/bin/gpurun --timeout 600 -- "mkdir -p gpurun_out/lab; ./build/gemm_lab $* > gpurun_out/lab/$out.txt 2>&1; echo rc=\$?" 2>&1 | tail -4
