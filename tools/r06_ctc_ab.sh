#!/bin/bash
# round 6: probability-domain CTC + oct hash -- parity tests, then the two fine-tune steps A (baseline library) / B (this build), arms interleaved
set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -6
bash tools/ab_lib.sh ctc_base gsoc-wav2vec2_amd/lib/libw2v2_base.so gsoc-wav2vec2_amd/lib/libw2v2.so 2 --precision bf16 --mode train --steps 10 --warmup 3 > /dev/null 2>&1
bash tools/ab_lib.sh ctc_large gsoc-wav2vec2_amd/lib/libw2v2_base.so gsoc-wav2vec2_amd/lib/libw2v2.so 2 --model large-robust --batch 16 --samples 480000 --precision bf16 --mode train --steps 6 --warmup 2 > /dev/null 2>&1
cat gpurun_out/abl_ctc_base.txt gpurun_out/abl_ctc_large.txt | cut -c1-300
