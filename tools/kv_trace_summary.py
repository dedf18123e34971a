"""busy / idle split and per-kernel totals of the LAST 15 steps' worth of a tools/kv_trace.py rocprofv3 trace (window = the last
15 head_fwd_kernel launches).  Usage: kv_trace_summary.py DB [OUT]"""
import collections, re, sqlite3, sys
db = sys.argv[1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
rows = sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start").fetchall()
heads = [i for i, r in enumerate(rows) if "head_fwd_kernel" in r[0]]
lo, hi = heads[-16] + 1, heads[-1] + 1
win = rows[lo:hi]
t0, t1 = win[0][1], win[-1][2]
busy, cur = 0.0, t0
agg = collections.defaultdict(lambda: [0, 0.0])
gaps = collections.defaultdict(lambda: [0, 0.0])
for i, (n, s, e) in enumerate(win):
    k = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    k = re.match(r"([\w:]+)(<[^(]*>)?", k)
    k = (k.group(1) + (k.group(2) or "")) if ("gemm_bf16" in n or "gemv_stream" in n) else k.group(1)
    if s > cur:
        prev = re.sub(r"\(anonymous namespace\)::|void |<.*|\(.*", "", win[i - 1][0]) if i else "-"
        g = gaps[(prev[:40], k[:40])]; g[0] += 1; g[1] += s - cur
    busy += max(0, e - max(s, cur)); cur = max(cur, e)
    a = agg[k[:90]]; a[0] += 1; a[1] += e - s
span = t1 - t0
print(f"# 15 K/V-reuse steps: {len(win)} dispatches, span {span/1e6:.2f} ms ({span/15e6:.2f} ms per step), GPU busy {100*busy/span:.1f} %", file=out)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k:<92} {v[0]:>6d} {v[1]/1e6:>9.3f} ms {v[1]/v[0]/1e3:>8.1f} us {100*v[1]/span:>6.2f} %", file=out)
print("# idle by (previous -> next), top 12", file=out)
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t/1e6:>8.3f} ms {n:>5d}x avg {t/n/1e3:>7.1f} us  {a} -> {b}", file=out)
