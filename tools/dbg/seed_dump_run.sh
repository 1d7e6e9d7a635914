#!/bin/bash
# development aid: the inputs spdp_align_s_seeded is handed for a few named queries, once from spdp_map_align_s (tools/e2e_q7.py) and once
# from the reference's CLI on the library (tools/dropin_demo.py); tools/dbg/seed_dump_diff.py compares the two files
mkdir -p gpurun_out
rm -f gpurun_out/dump_*.bin
H7=225695187,1023806735
H4=3664513946,3414961463,2165177895,1547868171,2552803560,79263496,2817295397,2529576989
SPDP_SEED_DUMP=$PWD/gpurun_out/dump_e2e_q7.bin SPDP_SEED_DUMP_HASH=$H7 timeout 300 python tools/e2e_q7.py --queries 20000 --genes 200 > gpurun_out/dump_e2e_q7.log 2>&1
SPDP_SEED_DUMP=$PWD/gpurun_out/dump_dropin_q7.bin SPDP_SEED_DUMP_HASH=$H7 timeout 300 python tools/dropin_demo.py --queries 20000 --genes 200 --modes Q7 --gpu-threads 16 > gpurun_out/dump_dropin_q7.log 2>&1
SPDP_SEED_DUMP=$PWD/gpurun_out/dump_e2e_c4.bin SPDP_SEED_DUMP_HASH=$H4 timeout 400 python tools/e2e_q7.py --queries 10000 --genes 400 --spacer 450000 --frag 500 --ori 3 > gpurun_out/dump_e2e_c4.log 2>&1
SPDP_SEED_DUMP=$PWD/gpurun_out/dump_dropin_c4.bin SPDP_SEED_DUMP_HASH=$H4 timeout 400 python tools/dropin_demo.py --queries 10000 --genes 400 --spacer 450000 --frag 500 --modes Q7 --gpu-threads 16 --strand=-S3 --antisense > gpurun_out/dump_dropin_c4.log 2>&1
ls -la gpurun_out/dump_*
python tools/dbg/seed_dump_diff.py gpurun_out/dump_e2e_q7.bin gpurun_out/dump_dropin_q7.bin > gpurun_out/dump_diff_q7.txt 2>&1
python tools/dbg/seed_dump_diff.py gpurun_out/dump_e2e_c4.bin gpurun_out/dump_dropin_c4.bin > gpurun_out/dump_diff_c4.txt 2>&1
rm -f gpurun_out/dump_*.bin
