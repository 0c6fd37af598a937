# Instruction-mix PMC passes over the lane rollouts of 9 x 9 / 10 x 10 / 11 x 11 (run on the GPU box: bash tools/pmc_lane_wide.sh [mode] [T])
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-partial_2}
T=${2:-32}
for S in 9 10 11; do
OUT=$R/gpurun_out/pmc_wide_${S}_$M
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/lane_wide_only.py $S $M $T > $OUT/p$i.log 2>&1
  echo "S=$S pass $i rc=$?"
done
python3 $R/tools/parse_pmc.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.json
done
