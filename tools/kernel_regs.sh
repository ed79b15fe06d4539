#!/bin/bash
# tools/kernel_regs.sh <object.hip.o> [name-filter] -- registers, LDS and spills of every gfx950 kernel in a HIP object
# (extracts the device code object and reads its metadata notes). Runs on the CPU container.
O=$1; F=${2:-.}
D=$(mktemp -d); cp "$O" $D/x.o
(cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1)
CO=$(ls $D/x.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO | awk '
/\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_count:/ {vg=$2}
/\.sgpr_spill_count:/ {ss=$2}
/\.vgpr_spill_count:/ {sp=$2; print vg" vgpr "sg" sgpr("ss" spilled) "lds" lds "sp" spill  "name}' | while read l; do n=$(echo "$l" | awk '{print $NF}' | c++filt | cut -c1-150); echo "$(echo "$l" | cut -d' ' -f1-9) $n"; done | grep -E "$F"
rm -rf $D
