#!/bin/bash
# the five BASELINE workloads at one HEAD on one box (review r5 item 7) + the forced single-rank RCCL exchange (item 6):
#   bash profiles/tools/r06_workloads.sh <tag>  -> gpurun_out/<tag>_bench_<workload>.json
T=${1:-r06}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-optimizer --profile-steps 0 --steps 20 --warmup 4"
for w in ddpm_b4_1024 flow_b4_1024 flow_b4_1344x768 flow_mixed_accum4 ddpm_b1_512; do
  $B --workload $w 2>gpurun_out/${T}_bench_$w.err | grep "^{" | tail -1 > gpurun_out/${T}_bench_$w.json
  python -c "import json,sys; d=json.load(open('gpurun_out/${T}_bench_$w.json')); print('$w', d['ms_per_step'], d['value'], d['step_mfma_frac'], d.get('clock',{}).get('sclk_mhz_under_load'))"
done
HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 $B --force-exchange --exchange zero1 2>gpurun_out/${T}_bench_rccl_forced.err | grep "^{" | tail -1 > gpurun_out/${T}_bench_rccl_forced_zero1.json
HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29573 $B --force-exchange --exchange allreduce 2>>gpurun_out/${T}_bench_rccl_forced.err | grep "^{" | tail -1 > gpurun_out/${T}_bench_rccl_forced_allreduce.json
for f in zero1 allreduce; do python -c "import json; d=json.load(open('gpurun_out/${T}_bench_rccl_forced_$f.json')); print('forced $f', d['ms_per_step'], d['config']['exchange']['backend'], d['config']['exchange']['what'][:60])"; done
