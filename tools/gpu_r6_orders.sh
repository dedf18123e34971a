#!/bin/bash
# round 6: the GPU suite under NAVILLM_POISON=1 in reversed and two seeded-random orders, then the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
for ord in reverse random:1 random:2; do
  tag=$(echo $ord | tr ':' '_')
  (NAVILLM_POISON=1 timeout 1800 python -m pytest tests -q -m gpu --nv-order $ord > $O/r6_suite_poison_$tag.log 2>&1; echo "rc=$?" >> $O/r6_suite_poison_$tag.log)
  tail -4 $O/r6_suite_poison_$tag.log
done
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_bench_v1.json 2> $O/r6_bench_v1.err; echo "rc=$?" >> $O/r6_bench_v1.err)
tail -12 $O/r6_bench_v1.err
