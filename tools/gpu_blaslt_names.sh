#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bl
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bl -o bl -- python tools/blaslt_names.py ${1:-4744} > gpurun_out/bl.log 2>&1
DB=$(find gpurun_out/prof_bl -name "*.db" | head -1)
python - "$DB" <<'P'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in cur.execute("select name, start, end from kernels order by start"):
    if "Cijk" in n:
        a = agg[n]; a[0] += 1; a[1] += (e - s) / 1e3
for k, v in agg.items():
    print(f"{v[0]:3d}x {v[1]/v[0]:8.1f} us  {k}")
P
find gpurun_out/prof_bl -name "*.db" -delete
