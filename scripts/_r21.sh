#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for b in 16 4; do timeout 300 python scripts/graph_probe.py --batch $b 2>&1 | tail -3; done
