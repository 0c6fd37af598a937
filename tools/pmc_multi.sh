# SQ counters of multi_rollout_kernel on cfg4 (full / no observation), one rocprofv3 --pmc pass per counter group.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for V in full none; do
OUT=$R/gpurun_out/pmc_multi_$V
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/multi_rollout_only.py $V > $OUT/p$i.log 2>&1
done
python3 $R/tools/parse_pmc.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.json
rm -rf $OUT/p?
done
