#!/usr/bin/env python
"""Find hidden LDS-DMA drains: compile every csrc/*.hip to gfx950 ISA and report, for kernels that use global_load_lds, each
`s_waitcnt vmcnt(0)` whose next instruction is an LDS READ (i.e. the compiler made the loop wait for the tile it had just
requested).  A `vmcnt(0)` in front of `s_barrier` (the intended end-of-tile wait) or in front of arithmetic on a loaded register
(bias / residual in an epilogue) is not reported.   python tools/isa_drain_scan.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gsoc-wav2vec2_amd", "csrc")
bad = 0
for f in sorted(os.listdir(SRC)):
    if not f.endswith(".hip"):
        continue
    out = os.path.join(tempfile.gettempdir(), "scan_" + f + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + SRC,
                        "-S", "--cuda-device-only", os.path.join(SRC, f), "-o", out], capture_output=True, text=True)
    if r.returncode:
        print(f, "does not compile:", r.stderr[-300:]); bad += 1; continue
    lines = open(out).read().split("\n")
    name, dma, hits = None, False, []
    def flush():
        global bad
        if name and dma and hits:
            bad += len(hits)
            print(f"{f}: {name[:100]}: vmcnt(0) followed by {sorted(set(h for _, h in hits))} at ISA lines {[i for i, _ in hits]}")
    for i, l in enumerate(lines):
        if l.startswith("_ZN") and "; @" in l:
            flush(); name, dma, hits = l.split(":")[0], False, []
        if "global_load_lds" in l:
            dma = True
        if re.search(r"s_waitcnt vmcnt\(0\)", l):
            j = i + 1
            while j < len(lines) and (not lines[j].strip() or lines[j].strip()[0] in ";."):
                j += 1
            nxt = lines[j].split()[0] if j < len(lines) and lines[j].split() else ""
            if nxt.startswith("ds_read"):
                hits.append((i + 1, nxt))
    flush()
print("drains found:", bad)
