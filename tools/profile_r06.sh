#!/bin/bash
# Round-5 profile: per workload of the bench line, the counters bench.py prices its kernels with (tools/profile_counters.py:
# kernel trace + stats, SQ_INSTS_VALU / SALU, FETCH_SIZE, WRITE_SIZE, each --pmc set in a pass of its own), then where the
# resident wave time goes for the headline, C5 and the block vote (tools/wave_pmc.sh).  Results under gpurun_out/; the
# summaries to keep go to profiles/r06_*.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
TAG=${TAG:-r06}
run() { name=$1; shift; echo "== $name"; timeout 1500 python tools/profile_counters.py --tag $TAG --name $name -- "$@" > gpurun_out/${TAG}_counters_$name.log 2>&1; tail -c 300 gpurun_out/${TAG}_counters_$name.log | tr '\n' ' '; echo; }
mkdir -p gpurun_out
for w in ${WORKLOADS:-c2 c4 c3 c5 a0 a1 c3_a0 c3_a1 blk}; do
  case $w in
    c2) run c2 --workload c2 ;;
    c4) run c4 --workload c4 --queries 20000 ;;
    c3) run c3 --workload c3 --queries 50000 ;;
    c5) run c5 --workload c5 --queries 32 ;;
    a0) run a0 --engines a0 --queries 1000 ;;
    a1) run a1 --engines a1 --queries 1000 ;;
    c3_a0) run c3_a0 --workload c3 --engines a0 --queries 1000 ;;
    c3_a1) run c3_a1 --workload c3 --engines a1 --queries 4000 ;;
    blk) run blk --workload blk --queries 200000 ;;
  esac
done
for w in ${WAVE:-c2 c5 blk}; do
  EXTRA=""; [ $w = c5 ] && EXTRA="--queries 32"; [ $w = blk ] && EXTRA="--queries 200000"
  EXTRA="$EXTRA" bash tools/wave_pmc.sh $w $TAG > gpurun_out/${TAG}_wave_$w.txt 2>&1
  tail -12 gpurun_out/${TAG}_wave_$w.txt
done
ls gpurun_out | head -50
