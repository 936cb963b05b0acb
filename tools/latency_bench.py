#!/usr/bin/env python
"""Forward latency / throughput across batch sizes and lengths (base config): python tools/latency_bench.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import torch
import wav2vec2
from wav2vec2 import variables as V
from wav2vec2.config import RobustWav2Vec2Config
torch.cuda.set_device(0)
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32"      # fp32 | bf16 | bf16x3
out = {}
for name, cfg, cases in (("base", wav2vec2.Wav2Vec2Config(), [(1, 50000), (1, 246000), (4, 246000), (8, 246000), (32, 246000), (64, 246000), (16, 480000)]),
                         ("large-robust", RobustWav2Vec2Config(), [(16, 246000)])):
    m = wav2vec2.Wav2Vec2ForCTC(cfg)
    m.set_precision(PREC)
    for B, L in cases:
        x = torch.randn(B, L, device="cuda")
        mask = torch.ones(B, L, dtype=torch.int32, device="cuda") if cfg.is_robust else None
        for _ in range(2): m(x, attention_mask=mask)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
        for _ in range(n): m(x, attention_mask=mask)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        out[f"{name} B={B} L={L}"] = dict(ms=round(dt * 1e3, 2), audio_s_per_s=round(B * L / 16000 / dt, 1))
        print(f"{name:13s} B={B:3d} L={L}: {dt*1e3:8.2f} ms  {B*L/16000/dt:9.1f} audio-s/s", flush=True)
    del m
print(json.dumps(out))
