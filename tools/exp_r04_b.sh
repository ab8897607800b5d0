#!/bin/bash
# round 4: the general (asymmetric-W) branch of the C driver on the device - the tests that exercise it, then the bench
# line with the 15k pair and the messy 250k pair
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04_b
rm -rf $out && mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "eigs or spectrum or pair_driver or timeout or open_mesh or messy or end_to_end or large_hole or recursive_eig or paired" > $out/pytest.txt 2>&1
echo "pytest rc=$?" > $out/progress.txt
tail -5 $out/pytest.txt
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" >> $out/progress.txt
