#!/bin/bash
set -e
mkdir -p gpurun_out
{
echo "## default rule"; python tools/sweep_15k.py | tail -1
echo "## sweep"; SWEEP_CUTS=${CUTS:-12,8,7} SWEEP_STRENGTHS=${STRENGTHS:-2.0,2.75,3.0} python tools/sweep_15k.py | tail -9
echo "## default rule"; python tools/sweep_15k.py | tail -1
} 2>&1 | tee gpurun_out/o_15k.log
