#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "g8" -s 2>&1 | grep -v "^$" | tail -12
timeout 600 python tools/host_profile_kv.py > gpurun_out/r03_host_profile_kv.txt 2>&1; echo rc=$?
head -120 gpurun_out/r03_host_profile_kv.txt
