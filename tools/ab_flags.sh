#!/bin/bash
# tools/ab_flags.sh <variant dirs...> -- like ab_variants.sh, crossed with table build flags (HSGPU_BUILD_FLAGS)
cd "$(dirname "$0")/.."
cp hyperscan_amd/lib/libhsgpu.so /tmp/libhsgpu_default.so
for v in default "$@"; do
  if [ "$v" = default ]; then cp /tmp/libhsgpu_default.so hyperscan_amd/lib/libhsgpu.so; else cp hyperscan_amd/lib/$v/libhsgpu.so hyperscan_amd/lib/libhsgpu.so; fi
  for fl in ${FLAGS:-0 256 128}; do
  for w in teddy64 fdr10k; do
    echo "$v flags=$fl $(HSGPU_BUILD_FLAGS=$fl timeout 120 python tools/kbench.py $w 2>&1 | grep -o "$w: kernel avg [0-9.]* ms.*confirm [0-9.]* ms")"
  done; done
done
cp /tmp/libhsgpu_default.so hyperscan_amd/lib/libhsgpu.so
