#!/usr/bin/env python
"""B = 1 forward latency: eager launches vs one captured HIP graph replay (torch.cuda.CUDAGraph on ROCm = hipGraph).
The C ABI only enqueues on the caller's stream, so the whole forward is capturable once the workspace exists."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsoc-wav2vec2_amd"))
import torch
import wav2vec2
torch.cuda.set_device(0)
m = wav2vec2.Wav2Vec2ForCTC(wav2vec2.Wav2Vec2Config())
m.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
for L in (50000, 246000):
    x = torch.randn(1, L, device="cuda")
    for _ in range(3): ref = m(x)                       # warm-up: workspace allocation, function attributes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): m(x)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m(x)                                            # settle the side stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = m(x)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(out, ref))
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 20
    print(f"B=1 L={L}: eager {eager*1e3:.2f} ms   hipGraph replay {graph*1e3:.2f} ms   identical logits: {same}   "
          f"({L/16000/graph:.0f} audio-s/s)", flush=True)
