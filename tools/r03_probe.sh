#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/c7; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_bf16 or shadow" 2>&1 | tail -3
timeout 300 python tools/gemm16_ab.py 2>&1 | grep -v amdgpu.ids | tail -22 | tee $O/ab.log
