"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace database, over the second half of the trace (steady state):
how much of the wall time is no kernel running, and after which kernels.  Usage: gap_summary.py DB [OUT]"""
import collections
import re
import sqlite3
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(void )?([\w:]+)(<[^(]*>)?", n)
    key = (m.group(2) + (m.group(3) or "")) if "gemm_bf16" in n else m.group(2)
    return key.replace("at::native::", "torch::")[:80]


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    t_mid = (rows[0][1] + rows[-1][2]) // 2
    rows = [r for r in rows if r[1] >= t_mid]
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = collections.defaultdict(lambda: [0, 0.0, 0.0])
    hist = collections.Counter()
    end_prev, name_prev = rows[0][2], rows[0][0]
    idle = 0
    for name, s, e in rows[1:]:
        g = s - end_prev
        if g > 0:
            idle += g
            a = gaps[(short(name_prev), short(name))]
            a[0] += 1; a[1] += g; a[2] = max(a[2], g)
            hist[min(int(g / 1e3) // 2 * 2, 50)] += 1
        if e > end_prev:
            end_prev, name_prev = e, name
    print(f"# second half of the trace: {len(rows)} dispatches, span {span / 1e6:.2f} ms, kernels running {busy / 1e6:.2f} ms, "
          f"idle between kernels {idle / 1e6:.2f} ms = {100 * idle / span:.2f} % of the span", file=out)
    print("# gap histogram (us: count): " + ", ".join(f"{k}-{k + 2}: {v}" if k < 50 else f">=50: {v}" for k, v in sorted(hist.items())), file=out)
    print(f"# {'after kernel -> before kernel':<120} {'gaps':>6} {'total_ms':>9} {'avg_us':>8} {'max_us':>8}", file=out)
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{(a + ' -> ' + b):<122} {v[0]:>6d} {v[1] / 1e6:>9.3f} {v[1] / v[0] / 1e3:>8.1f} {v[2] / 1e3:>8.1f}", file=out)


if __name__ == "__main__":
    main()
