#!/bin/bash
# tools/micro/build_gemmp.sh [extra hipcc flags] -- builds tools/micro/gemmp_bench (stamps on)
set -e
D=$(dirname "$(readlink -f "$0")")
OUT=${OUT:-gemmp_bench}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGEMMP_TIMING "$@" -o "$D/$OUT" "$D/gemmp_bench.hip" -lrocblas 2>&1 | grep -E "error|static assertion" -A4 || true
test -x "$D/$OUT" && ls -la "$D/$OUT"
