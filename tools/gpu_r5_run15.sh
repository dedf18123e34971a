#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_episode_gpu.py tests/test_parity_gpu.py tests/test_dp_gpu.py tests/test_parity_r5_gpu.py tests/test_host_cpu.py -m gpu -q -x -k "teacher_forced or one_launch or g12 or prefix_reuse or segments or truncated or abort or guards or collate or rehearsal" > gpurun_out/r5_lazy_tests_v15.log 2>&1; echo rc=$? >> gpurun_out/r5_lazy_tests_v15.log
tail -6 gpurun_out/r5_lazy_tests_v15.log | cut -c1-300
MODES=prefix_reuse bash tools/gpu_pmc_bench_r4.sh r05 2>&1 | tail -6
