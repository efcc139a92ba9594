#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for mt in 1 2 1 2; do
DASR_DSN_MT=$mt timeout 300 python bench.py --model dsn --per-type VGG --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DSN VGG mt $mt', j['ms_per_step'], j['value'])"
done
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "dsn" 2>&1 | tail -3
