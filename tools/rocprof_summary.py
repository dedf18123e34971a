"""Summarise a rocprofv3 --kernel-trace --stats rocpd database (bench_results.db) as text:
per-kernel launches / total / average duration, for profiles/.  Usage: rocprof_summary.py DB [OUT]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for name, s, e in rows:
        n = name.replace("(anonymous namespace)::", "")
        m = re.match(r"(void )?([\w:]+)(<[^(]*>)?", n)
        key = (m.group(2) + (m.group(3) or "")) if ("gemm_bf16" in n or "gemv_stream" in n) else m.group(2)
        if key.startswith("at::native::"):            # torch glue kernels: keep the functor + dtype so that fills / copies / adds stay apart
            f = re.search(r"(FillFunctor<[^>]*>|CUDAFunctor_add<[^>]*>|MulFunctor<[^>]*>|direct_copy_kernel_cuda|masked_fill|CatArrayBatchedCopy|"
                          r"scatter_gather|reduce_kernel|index_[a-z_]+|bernoulli|uniform|normal|random)", n)
            key = key.replace("at::native::", "torch::") + ("[" + f.group(1) + "]" if f else "")
        a = agg[key[:110]]
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(v[1] for v in agg.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}", file=out)
    print(f"# {len(rows)} dispatches, total kernel time {tot/1e3:.2f} ms, trace span {span/1e3:.2f} ms", file=out)
    print(f"# {'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}", file=out)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:<72} {v[0]:>7d} {v[1]/1e3:>10.2f} {v[1]/v[0]:>10.1f} {v[2]:>9.1f} {v[3]:>9.1f} {100*v[1]/tot:>6.2f}", file=out)


if __name__ == "__main__":
    main()
