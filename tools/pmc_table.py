#!/usr/bin/env python
"""Compact per-kernel table from the per-counter summaries tools/pmc_bench.sh leaves (gpurun_out/pmc_<tag>.md):
    python tools/pmc_table.py gpurun_out/pmc_fwd_bf16_r3.md [...]
One row per kernel: launches, MFMA pipe busy, MFMA work, L2 hit rate, fetch (x 2 corrected) / write MB, the wave-cycle split, LDS
bank conflicts, and the clock the kernel ran at if SQ_BUSY / GRBM counters allow it (GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import re, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    print(f"<!-- {path} -->")
    print("| kernel | launches / pass | MFMA pipe busy | MFMA MOPS x 512 / launch (TFLOP) | L2 hit rate | fetch MB (x2 corrected) | write MB | waiting / issue-stalled / issuing | LDS conflicts |")
    print("|---|---|---|---|---|---|---|---|---|")
    for sec in txt.split("### ")[1:]:
        head, body = sec.split("\n", 1)
        m = re.match(r"`(.*)` \((\d+) launches per pass\)", head)
        if not m:
            continue
        name, n = m.group(1), int(m.group(2))
        c = {r[0].strip(): float(r[1]) for r in re.findall(r"^\| ([A-Za-z0-9_]+) \| ([0-9.e+\-]+) \|$", body, re.M)}
        g = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 256 * 4 / 8) if g else float("nan")
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / 1e12
        hit = c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0))
        fetch, write = 2 * c.get("FETCH_SIZE", 0.0) * 1024 / 1e6, c.get("WRITE_SIZE", 0.0) * 1024 / 1e6
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        split = " / ".join("%.2f" % (c.get(x, 0.0) / wc) for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")) if wc else "-"
        print(f"| `{name}` | {n} | {busy:.3f} | {mops:.3f} | {hit:.3f} | {fetch:.1f} | {write:.1f} | {split} | {c.get('SQ_LDS_BANK_CONFLICT', 0):.4g} |")
    agg = re.findall(r"^Family aggregate: .*$", txt, re.M)
    print()
    print(agg[-1] if agg else "")
    print()
