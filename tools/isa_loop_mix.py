#!/usr/bin/env python
"""Instruction mix of the largest loop of every kernel in a .hip file (device ISA from hipcc -S): how many VALU / MFMA / LDS / SALU
instructions one trip of the main loop issues -- the input of a VALU-roofline estimate.   python tools/isa_loop_mix.py csrc/x.hip [-DW2V2_TUNING]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-I" + os.path.join(ROOT, "include"), "-S",
                "--cuda-device-only", "-o", out, src] + sys.argv[2:], check=True, capture_output=True)
lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[^:]*:", l)]
for idx, (i, name) in enumerate(starts):
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    body = lines[i:end]
    labels = {}
    for j, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = j
    best = None
    for j, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            span = (labels[m.group(1)], j)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not best:
        print(demangled[:110], ": no loop")
        continue
    loop = [l.strip() for l in body[best[0]:best[1] + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
    c = collections.Counter(l.split()[0] for l in loop)
    cls = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    valu = cls("v_") - cls("v_mfma")
    trans = sum(v for k, v in c.items() if k in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32"))
    print(f"{demangled[:110]}\n  loop: {len(loop)} instructions, VALU {valu} (transcendental {trans}), MFMA {cls('v_mfma')}, LDS {cls('ds_')}, "
          f"SALU {cls('s_')}, VMEM {cls('global_') + cls('buffer_')}")
    print("  " + ", ".join(f"{k}:{v}" for k, v in c.most_common(30)))
