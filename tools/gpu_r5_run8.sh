#!/bin/bash
# round 5, GPU call 8: kernel traces of the headline mode with and without the lazy prefix: kernel-time sums and idle gaps
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lz in 0 1; do
  rm -rf gpurun_out/prof_l$lz
  NAVILLM_EPISODE_LAZY_PREFIX=$lz timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l$lz -o b -- python bench.py --mode prefix_reuse --steps 12 --warmup 6 --prewarm 6 --no-extras --no-cpu-baseline --infer-steps 0 --no-profile --no-other-mode > gpurun_out/prof_l$lz.log 2>&1
  DB=$(find gpurun_out/prof_l$lz -name "*.db" | head -1)
  python tools/rocprof_summary.py "$DB" gpurun_out/r05_kernel_stats_lazy$lz.txt
  python - "$DB" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
k = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
rows = list(c.execute(f"select start, end from {k} order by start"))
# the last two whole episodes: find the 3 last adamw big gaps? simpler: overall busy/span of the last 60 % of the trace
n = len(rows); lo = int(n * 0.45)
rows = rows[lo:]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_e = rows[0][0]; gaps = []
for s, e in rows:
    if s > cur_e:
        gaps.append(s - cur_e)
    busy += max(0, e - max(s, cur_e)); cur_e = max(cur_e, e)
big = sorted(gaps, reverse=True)[:8]
print(f"tail of trace: span {span/1e6:.1f} ms busy {busy/1e6:.1f} ms idle {(span-busy)/1e6:.1f} ms ({100*(span-busy)/span:.1f} %); gaps >100us: {sum(1 for g in gaps if g>1e5)} totalling {sum(g for g in gaps if g>1e5)/1e6:.1f} ms; largest (ms): {[round(g/1e6,2) for g in big]}")
PY
  find gpurun_out/prof_l$lz -name "*.db" -delete
  head -24 gpurun_out/r05_kernel_stats_lazy$lz.txt | cut -c1-150
done
