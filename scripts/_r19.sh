#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
RUN_TESTS=1 RUN_BENCH=1 BENCH_ARGS="--no-cpu-baseline" bash scripts/gpu_round.sh 2>&1 | grep -E "passed|failed|exit|ms/step"
timeout 300 python bench.py --model dsn --per-type VGG --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DSN VGG', j['ms_per_step'], j['value'])"
