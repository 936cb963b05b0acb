#!/bin/bash
# kernel resource usage of one HIP source: tools/kres.sh gsoc-wav2vec2_amd/csrc/<file>.hip [extra hipcc flags]
# prints: VGPRs, AGPRs, scratch bytes / lane, LDS bytes, occupancy, demangled kernel name
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I$(dirname $0)/../include "$@" -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "error|Function Name|    VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -e 's/ \[-Rpass[^]]*\]//g' -e 's/.*remark: *//' \
 | awk '/Function Name/{if(n)print v,a,s,l,o,n; n=$3} /^VGPRs:/{v="v"$2} /AGPRs:/{a="a"$2} /ScratchSize/{s="scr"$NF} /LDS Size/{l="lds"$NF} /Occupancy/{o="occ"$NF} /error/{print} END{print v,a,s,l,o,n}' \
 | while read v a s l o n; do echo "$v $a $s $l $o $(echo $n | c++filt | cut -c1-110)"; done
