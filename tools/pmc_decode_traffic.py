"""FETCH_SIZE per dispatch of the decode-step kernels (rocprofv3 --kernel-trace --pmc FETCH_SIZE of tools/decode_probe.py) against the
bytes each launch has to read: weights N*K*2 (gemv_stream, by template variant) and 2*L*256 B per (sample, head) (attn_decode).
gfx950 correction as in tools/pmc_traffic.py: FETCH_SIZE (KiB) counts half of a wide coalesced stream -> bytes = 2 * 1024 * FETCH.
Usage: pmc_decode_traffic.py DB OUT"""
import collections, re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name='FETCH_SIZE' group by 1,2").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for k, _, v in rows:
    if "gemv_stream_kernel" in k or "attn_decode_kernel" in k or "rope_scatter" in k:
        name = re.sub(r"\(anonymous namespace\)::|void ", "", k)
        m = re.match(r"([\w:]+)(<[^(]*>)?", name)
        a = agg[m.group(1) + (m.group(2) or "")]
        a[0] += 1; a[1] += v; a[2] = min(a[2], v); a[3] = max(a[3], v)
d, ff, V = 4096, 11008, 32064
want = {"16, 2>": ("gate|up + RMSNorm + SwiGLU", 2 * ff * d * 2), "8, 1>": ("q|k|v + RMSNorm", 3 * d * d * 2),
        "2, 0>": ("o / down (mixed)", (d * d * 2 + d * ff * 2) / 2), "16, 0>": ("lm_head", V * d * 2)}
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE of tools/decode_probe.py (Vicuna-7B, B=8, ~650 cached tokens), per dispatch",
         "# HBM-side read bytes = 2 * 1024 * FETCH_SIZE[KiB] (gfx950 half-count correction, MI355X_MICROARCH.md)",
         f"# {'kernel':<58} {'launches':>8} {'avg_MB':>9} {'min_MB':>9} {'max_MB':>9}   must read"]
for k, (n, tot, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tag = next((f"{w[0]}: {w[1] / 1e6:.1f} MB of weights" for s_, w in want.items() if k.endswith(s_)), "")
    if "attn_decode" in k:
        tag = "K and V of the cache: 2 * L * 256 B * 32 heads * 8 samples = %.1f MB at L = 650" % (2 * 650 * 256 * 32 * 8 / 1e6)
    lines.append(f"{k:<60} {n:>8d} {2 * 1024 * tot / n / 1e6:>9.1f} {2 * 1024 * lo / 1e6:>9.1f} {2 * 1024 * hi / 1e6:>9.1f}   {tag}")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
