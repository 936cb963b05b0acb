#!/usr/bin/env python
"""How fast does the numpy oracle run on this host, by BLAS thread count and batch?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import numpy as np
from threadpoolctl import threadpool_limits, threadpool_info
from oracle import w2v2_oracle as O
from wav2vec2 import variables as V
from wav2vec2.config import Wav2Vec2Config
print("cpus", os.cpu_count(), [(d["internal_api"], d["num_threads"]) for d in threadpool_info()])
cfg = Wav2Vec2Config(); w = V.seeded_weights(cfg, 0); L = 246000
for B in (1, 4):
    x = V.hash_normal("probe", B * L, 0).reshape(B, L)
    for nt in (256, 64, 32, 16):
        with threadpool_limits(limits=nt):
            O.ctc_forward(cfg, w, x[:1])
            t0 = time.perf_counter(); O.ctc_forward(cfg, w, x); dt = time.perf_counter() - t0
        print(f"B={B} blas_threads={nt}: {dt:.2f} s  {B*L/16000/dt:.1f} audio-s/s", flush=True)
