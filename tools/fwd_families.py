#!/usr/bin/env python
"""Per-family kernel times of one configuration (no parity assertions: usable with the timing ablations of the kernels).
    python tools/fwd_families.py --precision bf16 [--mode train] [--batch 32] [--samples 246000] [--steps 5]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gsoc-wav2vec2_amd")]
import numpy as np
import torch
import wav2vec2
from wav2vec2 import variables as V

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16"); ap.add_argument("--mode", default="forward"); ap.add_argument("--model", default="base")
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--samples", type=int, default=246000); ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = wav2vec2.Wav2Vec2Config() if args.model == "base" else wav2vec2.RobustWav2Vec2Config()
m = wav2vec2.Wav2Vec2ForCTC(cfg, input_shape=(args.batch, args.samples))
m.set_precision(args.precision)
B, L = args.batch, args.samples
x = torch.randn((B, L), device=dev)
mask = torch.ones((B, L), device=dev, dtype=torch.int32) if cfg.is_robust else None
if args.mode == "train":
    rs = np.random.RandomState(7)
    labels = np.zeros((B, 256), np.int32)
    for b in range(B):
        n = rs.randint(24, 201); labels[b, :n] = rs.randint(1, 32, size=n)
    labels = torch.from_numpy(labels).to(dev)
    m.freeze_feature_extractor()
    tr = wav2vec2.Trainer(m, wav2vec2.CTCLoss(cfg, (B, L), division_factor=B), learning_rate=1e-4)
    step = lambda: tr.step(x, labels, attention_mask=mask)
else:
    step = lambda: m(x, attention_mask=mask)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps): step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.steps * 1e3
m.profile(True); m.profile_reset()
for _ in range(2): step()
torch.cuda.synchronize()
p = m.profile_read()
out = {"ms_per_step": round(wall, 3)}
out.update({k: {"ms": round(v["ms"] / 2, 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else 0} for k, v in p.items() if v["launches"]})
print(json.dumps(out))
