# Instruction-mix / wave-cycle PMC passes over the rollout kernel (run on the GPU box:
#   bash tools/pmc_instmix.sh [obs|noobs] [num_envs] [batch-steps per launch])
cd /tmp && export TMPDIR=/tmp
R=/root/repo
V=${1:-obs}
N=${2:-512}
CHUNK=${3:-256}
OUT=$R/gpurun_out/pmc_${V}_$N
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/rollout_only.py $N $CHUNK $V > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/parse_pmc.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.json
