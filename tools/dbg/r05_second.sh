#!/bin/bash
# round 5, second GPU call: the whole gpu suite (no -x), the a0 probes
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05_gputest.log
tail -6 gpurun_out/r05_gputest.log
TESTS=0 bash tools/dbg/a0_iter.sh
