#!/bin/bash
# tools/ab_variants.sh -- time library variants (hyperscan_amd/lib/<name>/libhsgpu.so, built with
# `make OUT=../lib/<name> CXXFLAGS=...`) back to back on one GPU box. Usage: ab_variants.sh v4 v6 ...
cd "$(dirname "$0")/.."
cp hyperscan_amd/lib/libhsgpu.so /tmp/libhsgpu_default.so
for v in default "$@"; do
  if [ "$v" = default ]; then cp /tmp/libhsgpu_default.so hyperscan_amd/lib/libhsgpu.so; else cp hyperscan_amd/lib/$v/libhsgpu.so hyperscan_amd/lib/libhsgpu.so; fi
  for w in ${WORKLOADS:-teddy64 fdr10k}; do
    echo "$v $(timeout 120 python tools/kbench.py $w 2>&1 | grep -o "$w: kernel avg [0-9.]* ms.*confirm [0-9.]* ms; matches [0-9]*; candidates [^;]*")"
  done
done
cp /tmp/libhsgpu_default.so hyperscan_amd/lib/libhsgpu.so
