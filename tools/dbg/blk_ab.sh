#!/bin/bash
# development aid: the block vote's bench leg with the kernel built for 2 / 3 / 4 waves per SIMD
cd spaln_amd/csrc
for w in 0 3 4; do
  cp spdp_blk_vote.hip /tmp/keep.hip
  if [ $w != 0 ]; then sed -i "s/__global__ void __launch_bounds__(64) spdp_blk_vote_wave/__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu($w,$w))) spdp_blk_vote_wave/" spdp_blk_vote.hip; fi
  make > /dev/null 2>&1
  cp /tmp/keep.hip spdp_blk_vote.hip
  cd ../..
  echo "waves_per_eu $w: $(python bench.py --workload blk --queries 200000 --steps 3 --warmup 1 --legs none 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("identical_to_oracle_on_sample"))')"
  cd spaln_amd/csrc
done
make > /dev/null 2>&1
