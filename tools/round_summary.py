#!/usr/bin/env python
"""Tables of a round's end-of-round evidence (tools/refresh_round.sh -> gpurun_out/final/) for profiles/rNN_summary.md:
   python tools/round_summary.py gpurun_out/final > /tmp/tables.md
Every number is read from a bench JSON line in that directory; nothing is typed by hand."""
import glob, json, os, sys
D = sys.argv[1]


def load(name):
    p = os.path.join(D, name)
    if not os.path.exists(p):
        return None
    lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


ROWS = [("bench_n1_full.json", "base fp32 forward 32x246000 (headline, BASELINE configs[1]; the driver's command)"),
        ("bench_n1_launched_full.json", "the same under torch.distributed.run with one rank (RCCL group of 1)"),
        ("bench_f16x2_n1_full.json", "base f16x2 forward 32x246000 (opt-in, fp32-grade)"),
        ("bench_bf16x3_n1_full.json", "base bf16x3 forward 32x246000 (opt-in, fp32-grade)"),
        ("bench_bf16_n1_full.json", "base bf16 forward 32x246000"),
        ("bench_train_n1_full.json", "base fp32 CTC fine-tune step 32x246000"),
        ("bench_bf16_train_n1_full.json", "base bf16 CTC fine-tune step 32x246000 (configs[2] per GPU)"),
        ("bench_large_robust_fwd_n1_full.json", "large-robust fp32 forward 16x246000 (configs[3])"),
        ("bench_large_robust_bf16_n1_full.json", "large-robust bf16 forward 16x480000"),
        ("bench_large_robust_bf16_train_n1_full.json", "large-robust bf16 fine-tune step 16x480000 (configs[4] per GPU)")]
print("| workload | ms / step | audio-s / s | dominant GEMM family: TF, frac of nominal peak | shader clock under load | frac at that clock | launches / step | max abs logit err vs HF fp64 |")
print("|---|---|---|---|---|---|---|---|")
for fn, label in ROWS:
    d = load(fn)
    if not d:
        continue
    r = d.get("roofline") or {}
    err = d.get("max_abs_logit_err")
    print(f"| {label} | {d['ms_per_step']} | {d['value']:.0f} | {r.get('achieved')} TF of {r.get('peak')} = {r.get('frac')} | "
          f"{r.get('clock_mhz_under_load')} MHz | {r.get('frac_clock_adjusted')} | {d.get('kernel_launches_per_step', '-')} | "
          f"{('%.1e' % err) if isinstance(err, (int, float)) else '-'} |")
d = load("bench_n1_full.json")
if d:
    print("\nThe driver's line also carries, measured by the same process right after the headline:\n")
    print("| object | ms / step | audio-s / s | family frac (nominal / at clock) | clock MHz | kernel launches / step | unattributed ms | traffic / compulsory per step |")
    print("|---|---|---|---|---|---|---|---|")
    for k, v in d.items():
        if isinstance(v, dict) and "ms_per_step" in v and "roofline" in v:
            r = v["roofline"]
            print(f"| `{k}` | {v['ms_per_step']} | {v.get('value', 0):.0f} | {r.get('frac')} / {r.get('frac_clock_adjusted')} | {r.get('clock_mhz_under_load')} | "
                  f"{v.get('kernel_launches_per_step', '-')} | {v.get('unattributed_ms', '-')} | {r.get('traffic_over_algorithmic', '-')} |")
    cb = d.get("cpu_baseline") or {}
    print(f"\nWhole default run: {d.get('bench_wall_s')} s; CPU baseline {cb.get('value')} {cb.get('unit')} on {cb.get('cores')} cores ({cb.get('kind')}); GPU / CPU = {d.get('gpu_over_cpu')}.")
    for k in ("configs2_train_bf16", "configs4_large_train_bf16"):
        f = (d.get(k) or {}).get("families") or {}
        if f:
            print(f"\n`{k}` families (ms / step): " + ", ".join(f"{n} {v['ms_per_step']}" for n, v in f.items()))
