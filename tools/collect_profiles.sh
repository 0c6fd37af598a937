#!/bin/bash
# Re-creates the artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag>      ->  gpurun_out/profiles_<tag>/
# rocprofv3 is always given the program itself after `--` and counters are collected in their own passes.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (un-profiled), exactly as the driver runs it, and with the defaults
python3 $R/bench.py > $OUT/${TAG}_bench_cfg2.json 2> $OUT/bench.err
python3 $R/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OUT/${TAG}_bench_cfg2_steps20.json 2>> $OUT/bench.err
# 2. kernel trace + stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $R/bench.py --no-extra --no-cpu-baseline > $OUT/${TAG}_cfg2_rollout_bench_under_rocprof.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg2_rollout_kernel_stats.csv
cp $(find $OUT/trace -name "*domain_stats.csv" | head -1) $OUT/${TAG}_cfg2_rollout_domain_stats.csv
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes over the calibrated workload
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/tools/traffic_workload.py > $OUT/pmc_$C.log 2>&1
done
python3 $R/tools/parse_pmc.py $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv') > $OUT/${TAG}_pmc_fetch_write_summary.json
python3 $R/tools/make_traffic_json.py $OUT/${TAG}_pmc_fetch_write_summary.json > $OUT/hbm_traffic.json
# 4. kernel durations of that workload (per kernel and grid size): the per-call kernels and the big-grid rollouts
python3 $R/tools/trace_summary.py $(find $OUT/pmc_WRITE_SIZE -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_kernel_times_traffic_workload.txt
# 5. instruction mix / wave cycles of the cfg2 rollout kernel and of the cfg4 MultiSnake rollout
bash $R/tools/pmc_instmix.sh obs 512 256 > /dev/null
cp $R/gpurun_out/pmc_obs_512/summary.json $OUT/${TAG}_cfg2_rollout_instmix_pmc.json
# ... and of the one-env-per-lane rollout at the cfg3 shapes (8 192 envs per GPU, all 65 536 on one GPU)
bash $R/tools/pmc_instmix.sh obs 8192 128 > /dev/null
cp $R/gpurun_out/pmc_obs_8192/summary.json $OUT/${TAG}_lane_rollout_8192_instmix_pmc.json
bash $R/tools/pmc_instmix.sh obs 65536 64 > /dev/null
cp $R/gpurun_out/pmc_obs_65536/summary.json $OUT/${TAG}_lane_rollout_65536_instmix_pmc.json
bash $R/tools/pmc_multi.sh > /dev/null
cp $R/gpurun_out/pmc_multi_full/summary.json $OUT/${TAG}_multi_rollout_cfg4_full_instmix_pmc.json
cp $R/gpurun_out/pmc_multi_none/summary.json $OUT/${TAG}_multi_rollout_cfg4_noobs_instmix_pmc.json
# ... and of the per-call step on the resident mirror at 65 536 envs (reference form, and without the reset observation)
bash $R/tools/pmc_percall.sh 65536 ref > /dev/null
cp $R/gpurun_out/pmc_percall/summary.json $OUT/${TAG}_percall_65536x9_resident_ref_instmix_pmc.json
bash $R/tools/pmc_percall.sh 65536 noobs > /dev/null
cp $R/gpurun_out/pmc_percall/summary.json $OUT/${TAG}_percall_65536x9_resident_noobs_instmix_pmc.json
# 5b. in-kernel timelines (needs `make -C wurm_amd/csrc timeline`) and the per-call loop with / without the mirror
if [ -f $R/wurm_amd/libwurm_hip_timeline.so ]; then
  rm -f $OUT/${TAG}_kernel_timeline.txt
  for K in "--kernel lane_step" "--kernel resident --form ref" "--kernel resident --form noobs" "--kernel resident --form noobs --epw 16" "--kernel grid --envs 8192 --form noobs"; do
    WURM_HIP_LIBRARY=$R/wurm_amd/libwurm_hip_timeline.so python3 $R/tools/kernel_timeline.py $K 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_kernel_timeline.txt
  done
  # MultiSnake: the per-call step at cfg4 and at the speeds.py shape, the training-dynamics rollout (totals per step)
  for K in "" "--speeds --snakes 10 --size 36" "--rollout 16"; do
    WURM_HIP_LIBRARY=$R/wurm_amd/libwurm_hip_timeline.so python3 $R/tools/multi_timeline.py $K 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_kernel_timeline.txt
  done
fi
python3 $R/tools/percall_sweep.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_percall_sweep.txt
# 6. the other BASELINE shapes
python3 $R/tools/bench_configs.py > $OUT/${TAG}_percall_all_configs.jsonl 2>/dev/null
python3 $R/tools/bench_big.py > $OUT/${TAG}_big_configs.txt 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
