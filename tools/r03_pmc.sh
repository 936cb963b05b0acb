#!/bin/bash
# PMC evidence for the bf16 GEMM family: the round-3 routing (shipping library) and the round-2 kernel alone (tools-only build, W2V2_GEMM16_SW=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/pmc_bench.sh fwd_bf16_r3 gemm_bf16 --precision bf16 > gpurun_out/pmc_fwd_r3.log 2>&1; tail -3 gpurun_out/pmc_fwd_r3.log
W2V2_NATIVE_LIB=$R/gsoc-wav2vec2_amd/lib/libw2v2_tuning.so W2V2_GEMM16_SW=0 bash tools/pmc_bench.sh fwd_bf16_r2kernel gemm_bf16 --precision bf16 > gpurun_out/pmc_fwd_r2.log 2>&1; tail -3 gpurun_out/pmc_fwd_r2.log
bash tools/pmc_bench.sh train_bf16_r3 gemm_bf16 --precision bf16 --mode train > gpurun_out/pmc_train_r3.log 2>&1; tail -3 gpurun_out/pmc_train_r3.log
