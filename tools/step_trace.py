"""Where one training episode's GPU time goes, from a rocprofv3 --kernel-trace rocpd database of bench.py:
the window between the last two optimizer updates (adamw_kernel launches with the largest grid = the decoder segment) is one
steady-state episode (6 nav steps).  Prints: busy vs idle time of the window, per-kernel totals with FULL names for the
torch-native kernels (so the functor is visible), and the largest idle gaps with the kernels either side.
Usage: step_trace.py DB [OUT]"""
import collections
import re
import sqlite3
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    if n.startswith("void at::") or n.startswith("at::"):
        n = re.sub(r"^void ", "", n)
        return n[:230]
    m = re.match(r"(void )?([\w:]+)(<[^(]*>)?", n)
    return (m.group(2) + (m.group(3) or "")) if "gemm_bf16" in n else m.group(2)


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    big = [i for i, (n, s, e) in enumerate(rows) if "adamw_kernel" in n and (e - s) > 5e6]
    if len(big) < 2:
        print("fewer than two optimizer updates in the trace", file=out)
        return
    lo, hi = big[-2] + 1, big[-1] + 1
    # the small adamw launches of the same update (fp32 side, late-born segments) follow the big one: skip them at the front
    while "adamw_kernel" in rows[lo][0]:
        lo += 1
    win = rows[lo:hi]
    t0, t1 = win[0][1], win[-1][2]
    busy = 0.0
    cur_end = t0
    gaps = []
    agg = collections.defaultdict(lambda: [0, 0.0])
    for i, (n, s, e) in enumerate(win):
        if s > cur_end:
            gaps.append((s - cur_end, short(win[i - 1][0]) if i else "-", short(n)))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
        a = agg[short(n)]
        a[0] += 1
        a[1] += (e - s)
    span = t1 - t0
    ksum = sum(v[1] for v in agg.values())
    print(f"# steady-state episode window: {len(win)} dispatches, span {span/1e6:.2f} ms, GPU busy {busy/1e6:.2f} ms "
          f"({100*busy/span:.1f} %), idle {100*(1-busy/span):.1f} %, sum of kernel durations {ksum/1e6:.2f} ms "
          f"(overlap on side streams {100*(ksum-busy)/span:.1f} % of span)", file=out)
    print(f"# {'kernel':<90} {'calls':>6} {'total_ms':>9} {'avg_us':>9} {'pct_span':>8}", file=out)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:<92} {v[0]:>6d} {v[1]/1e6:>9.3f} {v[1]/v[0]/1e3:>9.1f} {100*v[1]/span:>8.2f}", file=out)
    print("# idle gaps: total %.2f ms in %d gaps; histogram (us): " % (sum(g[0] for g in gaps) / 1e6, len(gaps)) +
          ", ".join(f"{a}-{b}: {sum(1 for g in gaps if a*1e3 <= g[0] < b*1e3)} ({sum(g[0] for g in gaps if a*1e3 <= g[0] < b*1e3)/1e6:.2f} ms)"
                    for a, b in ((0, 2), (2, 5), (5, 10), (10, 50), (50, 1e9))), file=out)
    by_pair = collections.defaultdict(lambda: [0, 0.0])
    for g, a, b in gaps:
        p = by_pair[(a[:60], b[:60])]
        p[0] += 1
        p[1] += g
    print("# idle time by (previous kernel -> next kernel), top 25", file=out)
    for (a, b), (n, tot) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{tot/1e6:>8.3f} ms {n:>5d}x avg {tot/n/1e3:>7.1f} us   {a}  ->  {b}", file=out)


if __name__ == "__main__":
    main()
