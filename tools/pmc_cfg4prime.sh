# SQ / instruction-cache counters of the cfg4' rollout (multi_rollout_kernel, training dynamics, partial_5), one rocprofv3 --pmc
# pass per counter group:  bash tools/pmc_cfg4prime.sh  ->  gpurun_out/pmc_cfg4prime/summary.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_cfg4prime
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/cfg4prime_probe.py --percall 0 > $OUT/p$i.log 2>&1
done
python3 $R/tools/parse_pmc.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.json
rm -rf $OUT/p?
