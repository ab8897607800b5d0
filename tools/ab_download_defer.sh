#!/bin/bash
# eigenvector downloads behind the 1-NN search (default) against queued at once (PF_DOWNLOAD_DEFER=0), 250k pair
run() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['breakdown_ms_per_step'].items()})"; }
for rep in 1 2 3; do
echo "## deferred"; run
echo "## PF_DOWNLOAD_DEFER=0"; PF_DOWNLOAD_DEFER=0 run
done
