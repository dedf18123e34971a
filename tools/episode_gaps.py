"""Idle time of the GPU per training episode from a rocprofv3 --kernel-trace database: episodes are delimited by the optimizer's big
adamw_kernel launch; per episode: span, kernel-busy time, idle time, and the largest gaps with the kernels around them.
usage: episode_gaps.py DB [OUT]"""
import re
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("at::native::", "torch::")
    m = re.match(r"(void )?([\w:]+)", n)
    return m.group(2)[:48] if m else n[:48]


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, (n, s, e) in enumerate(rows) if "adamw_kernel" in n and e - s > 5e6]      # the LM buffer's update: ~19 ms
    print(f"# {len(rows)} dispatches, {len(marks)} optimizer steps", file=out)
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        span = seg[-1][2] - rows[a][2]
        busy, end_prev, gaps = 0, rows[a][2], []
        prev_name = rows[a][0]
        for k, (n, s, e) in enumerate(seg):
            if s > end_prev:
                gaps.append((s - end_prev, prev_name, n, k))
            busy += max(0, e - max(s, end_prev))
            if e > end_prev:
                end_prev, prev_name = e, n
        idle = span - busy
        top = sorted(gaps, reverse=True)[:6]
        print(f"episode: span {span / 1e6:7.2f} ms  busy {busy / 1e6:7.2f}  idle {idle / 1e6:6.2f} ({100 * idle / span:4.1f} %)  {len(seg)} launches; gaps > 50 us: "
              f"{sum(1 for g in gaps if g[0] > 5e4)} = {sum(g[0] for g in gaps if g[0] > 5e4) / 1e6:.2f} ms; sum of gaps <= 50 us: {sum(g[0] for g in gaps if g[0] <= 5e4) / 1e6:.2f} ms", file=out)
        for g, p, n, k in top:
            print(f"      {g / 1e3:8.1f} us  after {short(p):<40} before {short(n):<40} (launch {k} of the episode)", file=out)


if __name__ == "__main__":
    main()
