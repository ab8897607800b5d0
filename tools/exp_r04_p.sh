#!/bin/bash
set -e
mkdir -p gpurun_out
SWEEP_POINTS=${POINTS:-12:2.0,8:2.5,7:2.75,6:3.0,8:3.0,6:2.5} python tools/sweep_messy.py 2>&1 | tee gpurun_out/p_messy.log
