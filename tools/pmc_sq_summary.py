"""Summarise the SQ / LDS / TCC counter passes of tools/gpu_pmc.sh (rocpd databases) per GEMM kernel -> text."""
import collections, glob, re, sqlite3, sys
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for db in sorted(glob.glob(sys.argv[1] + "/*_results.db")):
    c = sqlite3.connect(db)
    try:
        rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by 1,2,3").fetchall()
    except sqlite3.OperationalError:
        continue
    for k, cn, _, v in rows:
        if "gemm_bf16_kernel" in k:
            a = agg[re.sub(r"\(anonymous namespace\)::|void |\(.*", "", k)][cn]
            a[0] += 1; a[1] += v
print("# rocprofv3 --kernel-trace --pmc <SQ / LDS / TCC counters> of tools/gemm_pmc.py (M=5152, N=12288, K=4096; one launch set per", file=out)
print("# layout), separate passes per counter group (tools/gpu_pmc.sh); per-dispatch averages, summed over the counter's instances", file=out)
for k, d in sorted(agg.items()):
    print(f"\n{k}", file=out)
    for cn, (n, tot) in sorted(d.items()):
        print(f"  {cn:<28} {tot / n:>16.1f}   ({n} dispatches)", file=out)
    g = {cn: tot / n for cn, (n, tot) in d.items()}
    if "SQ_WAVE_CYCLES" in g:
        w = g["SQ_WAVE_CYCLES"]
        print("  -> of wave cycles: " + ", ".join(f"{c} {100 * g[c] / w:.1f}%" for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS") if c in g), file=out)
    if "SQ_LDS_IDX_ACTIVE" in g and "SQ_INSTS_LDS" in g and g["SQ_INSTS_LDS"]:
        print(f"  -> LDS cycles per LDS instruction {g['SQ_LDS_IDX_ACTIVE'] / g['SQ_INSTS_LDS']:.2f}, bank-conflict cycles {g.get('SQ_LDS_BANK_CONFLICT', 0):.0f}", file=out)
    if "TCC_HIT_sum" in g and "TCC_MISS_sum" in g:
        print(f"  -> L2 hit rate {100 * g['TCC_HIT_sum'] / (g['TCC_HIT_sum'] + g['TCC_MISS_sum']):.1f}%", file=out)
