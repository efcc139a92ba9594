#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for pt in VGG LPIPS VGG; do
timeout 300 python bench.py --model dsn --per-type $pt --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$pt', j['ms_per_step'], j['value'], j['log'])"
done
