#!/bin/bash
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "adamw or g8" 2>&1 | tail -2
timeout 600 python tools/adamw_time.py 2>&1 | tail -4
