#!/bin/bash
# Compile every HIP source of the package for gfx950 and list the kernels that use scratch memory (spills or stack arrays).
# The product path is meant to have none in its loops: a spilling epilogue once cost the fine-tune step 12 ms (GEMM study section 14).
# Known: gemm_split_kernel<16, 4> parks 4 registers across its K loop (preheader store, exit reload).
# Known and accepted: gemm_bf16_kernel<1, 256, 256, 2, 4, 1> (tile-study instance behind W2V2_GEMM16_CFG=2, 256 VGPRs).
R=$(cd "$(dirname "$0")/.." && pwd)
for f in $R/gsoc-wav2vec2_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
    | grep -E "Function Name|ScratchSize" | paste - - | grep -v "ScratchSize \[bytes/lane\]: 0" \
    | sed -e 's/.*Function Name: //' -e 's/ \[-Rpass[^]]*\]//g' | awk -v f=$(basename $f) '{print f": "$0}'
done
echo "scan done"
