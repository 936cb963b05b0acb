#!/usr/bin/env python
"""Audit of kernels that issue ds_read_* from inline asm and wait for them with a later asm s_waitcnt (gemm_bf16_pp.hip,
gemm_bf16.hip, attention_bf16.hip): between an asm read and the wait that retires it, hipcc must not touch the destination
registers (it counts them as written at the end of the asm statement -- cdna_hip_programming.md, 'What hipcc does not do').

    python tools/asm_audit_reads.py gsoc-wav2vec2_amd/csrc/gemm_bf16_pp.hip [-DW2V2_TUNING]

Compiles to ISA, then for every kernel lists compiler instructions that read or write a pending destination, plus scratch use."""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]; extra = sys.argv[2:]
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-S", "--cuda-device-only", src, "-o", out] + extra,
                      stderr=subprocess.DEVNULL)
text = open(out).read().splitlines(); os.unlink(out)
reg = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")
def regs_of(tok):
    s = set()
    for m in reg.finditer(tok):
        if m.group(1): s.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: s.add(int(m.group(3)))
    return s
kernel, pending, in_asm, bad, nreads, scratch, order = None, {}, False, 0, 0, 0, []
for ln, line in enumerate(text, 1):
    t = line.strip()
    m = re.match(r"^(_Z\w+):", t)
    if m:
        kernel, pending, order = m.group(1), {}, []
    if t.startswith(";;#ASMSTART"): in_asm = True; continue
    if t.startswith(";;#ASMEND"): in_asm = False; continue
    if not t or t.startswith(";") or t.startswith("."): continue
    if "scratch_" in t: scratch += 1
    if in_asm:
        if t.startswith("ds_read"):
            dst = t.split()[1].rstrip(",")
            order.append((ln, regs_of(dst)))
            nreads += 1
        elif t.startswith("ds_write"):
            order.append((ln, set()))          # counts in lgkmcnt, owns no destination
        else:
            m2 = re.search(r"lgkmcnt\((\d+)\)", t)
            if m2:                               # LDS operations retire in order: all but the N youngest are done
                n = int(m2.group(1))
                order = order[len(order) - n:] if n else []
        pending = {r: l0 for l0, rs in order for r in rs}
        continue
    if pending:
        ops = t.split(None, 1)
        touched = regs_of(ops[1]) if len(ops) > 1 else set()
        hit = touched & set(pending)
        if hit:
            bad += 1
            print(f"{kernel}: line {ln}: `{t}` touches v{sorted(hit)[:4]}... pending since line {min(pending[r] for r in hit)}")
print(f"{nreads} asm ds_reads audited, {bad} suspicious instruction(s), {scratch} scratch access(es)")
sys.exit(1 if bad else 0)
