#!/bin/bash
# round 4, GPU session 1: (a) full -m gpu suite with the multi-threaded enqueuer as default, (b) the stream-count x enqueue-mode A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; exec < /dev/null
rm -f gpurun_out/parity_margins.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r04_c1_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r04_c1_pytest.log
tail -4 gpurun_out/r04_c1_pytest.log
timeout 1500 python scripts/r04_streams.py --rounds 2 --steps 8 > gpurun_out/r04_streams_raw.txt 2> gpurun_out/r04_streams.err
echo "streams exit $?"
tail -14 gpurun_out/r04_streams_raw.txt
tail -3 gpurun_out/r04_streams.err
