#!/bin/bash
mkdir -p gpurun_out
for lz in 1 0; do echo "NAVILLM_EPISODE_LAZY_PREFIX=$lz"; NAVILLM_EPISODE_LAZY_PREFIX=$lz python tools/episode_host_probe.py 6 2>&1 | tail -8; done
