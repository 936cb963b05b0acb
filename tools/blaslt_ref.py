#!/usr/bin/env python
"""Reference point only (not part of the product path): torch.matmul (hipBLASLt) bf16 rates on the GEMM shapes of the base model,
to know what the hardware gives a tuned library kernel on the same shapes.   python tools/blaslt_ref.py"""
import torch, time, json
dev = torch.device("cuda:0")
shapes = [("qkv", 24576, 2304, 768), ("out", 24576, 768, 768), ("ffn1", 24576, 3072, 768), ("ffn2", 24576, 768, 3072),
          ("conv1", 32 * 12299, 512, 1536), ("conv5", 32 * 1537, 512, 1024), ("big", 8192, 8192, 8192)]
out = {}
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5): c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out[name] = {"M": M, "N": N, "K": K, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}
print(json.dumps(out, indent=1))
