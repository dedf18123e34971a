"""Per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (any counter list).
Usage: pmc_sq_summary.py DB OUT_TXT "<header line>" [kernel-name substrings...]"""
import collections
import re
import sqlite3
import sys

db, out, header = sys.argv[1], sys.argv[2], sys.argv[3]
want = sys.argv[4:]
c = sqlite3.connect(db)
rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by 1,2,3").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for k, cn, _, v in rows:
    name = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", k)
    if want and not any(w in name for w in want):
        continue
    a = agg[name][cn]
    a[0] += 1
    a[1] += v
counters = sorted({cn for d in agg.values() for cn in d})
lines = [f"# {header}", "# per-dispatch averages, summed over all SQ instances; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave,",
         "# SQ_VALU_MFMA_BUSY_CYCLES and the LDS counters count cycles (MI355X_MICROARCH.md, rocprofv3 PMC slots)",
         "# kernel | launches | " + " | ".join(counters)]
for name, d in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1].values())):
    n = max(x[0] for x in d.values())
    lines.append(f"{name} | {n} | " + " | ".join(f"{d[cn][1] / max(d[cn][0], 1):.4g}" if cn in d else "-" for cn in counters))
    wc = d.get("SQ_WAVE_CYCLES")
    if wc and wc[1] > 0:
        parts = []
        for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
            if cn in d:
                parts.append(f"{cn[3:]} {d[cn][1] / wc[1]:.1%}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"][1] > 0:
            parts.append(f"MFMA_BUSY / BUSY_CYCLES {d['SQ_VALU_MFMA_BUSY_CYCLES'][1] / d['SQ_BUSY_CYCLES'][1]:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d and d["SQ_LDS_IDX_ACTIVE"][1] > 0:
            parts.append(f"LDS_BANK_CONFLICT / LDS_IDX_ACTIVE {d['SQ_LDS_BANK_CONFLICT'][1] / d['SQ_LDS_IDX_ACTIVE'][1]:.3f}")
        lines.append("#    of WAVE_CYCLES: " + ", ".join(parts))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
