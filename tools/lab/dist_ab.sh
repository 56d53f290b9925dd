#!/bin/bash
# lab (round 4): the 20-step headline with and without an RCCL process group up (one rank), alternating on one box
one() { python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-scan-roofline --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.2f us/step' % (d['ms_per_step'] * 1e3))"; }
for rep in 1 2 3 4; do
    echo "plain:"; one
    echo "RCCL group (DOA_BENCH_FORCE_DIST=1):"; DOA_BENCH_FORCE_DIST=1 one
done
