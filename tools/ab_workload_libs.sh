#!/bin/bash
# tools/ab_workload_libs.sh <case> <variant...> -- same-box timing of one large workload (tools/ab_workload.py) for the default library
# and for experiment builds (make -C rust-brotli_amd variant NAME=...)
cd ${GRAFT_REPO_ROOT:-/root/repo}
CASE=$1; shift
for v in "" "$@"; do
  if [ -n "$v" ]; then export BROTLI_MI355X_LIB=$PWD/rust-brotli_amd/libbrotli_mi355x_$v.so; else unset BROTLI_MI355X_LIB; fi
  echo "[$v] $(timeout 300 python tools/ab_workload.py $CASE 2>/dev/null | tail -1 | cut -c1-200)"
done
