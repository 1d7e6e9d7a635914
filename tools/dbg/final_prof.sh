ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
timeout 300 bash tools/hx_pmc.sh a1 1000 > gpurun_out/hx_a1.txt 2>&1
timeout 300 bash tools/hx_pmc.sh a0 1000 > gpurun_out/hx_a0.txt 2>&1
timeout 900 bash tools/profile_round.sh c2 r04f > gpurun_out/prof_c2.txt 2>&1
timeout 600 python tools/dropin_demo.py --queries 2000 --genes 200 --modes Q7,Q4 --gpu-threads 1000 > gpurun_out/dropin.json 2> gpurun_out/dropin.err
tail -3 gpurun_out/hx_a1.txt; tail -2 gpurun_out/prof_c2.txt | cut -c1-300; cut -c1-600 gpurun_out/dropin.json
