#!/bin/bash
# same-box A/B of two checkouts of the package (ab_old/ = an export of an earlier commit with its own .so): whole prefix-reuse episodes
for rep in 1 2; do
for root in ab_old .; do
echo "== $root"
NAVILLM_PKG_ROOT=$(realpath $root) EPISODE_REPS=7 timeout 600 python tools/episode_profile.py 2>&1 | grep "^episode" | tail -6 | awk '{s+=$3; print} END {print "   mean of 6: " s/6 " ms"}' | tail -3
done
done
