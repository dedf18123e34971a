#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/fp8_overlap_ab.py > gpurun_out/r03_fp8_overlap_ab.txt 2>&1; echo rc=$? >> gpurun_out/r03_fp8_overlap_ab.txt
tail -12 gpurun_out/r03_fp8_overlap_ab.txt
bash tools/gpu_r3_round.sh ${1:-v6}
