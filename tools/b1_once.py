#!/usr/bin/env python
"""A few B = 1 forwards at a given length (for rocprofv3 --kernel-trace): python tools/b1_once.py 50000"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import torch, wav2vec2
L = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
m = wav2vec2.Wav2Vec2ForCTC(wav2vec2.Wav2Vec2Config())
x = torch.randn(1, L, device="cuda")
for _ in range(6): m(x)
torch.cuda.synchronize()
