# instruction mix / wave cycles of the per-call step: bash tools/pmc_percall.sh [envs] [ref|noobs]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_percall
rm -rf $OUT
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/percall_only.py ${1:-65536} ${2:-noobs} > $OUT/p$i.log 2>&1
done
python3 $R/tools/parse_pmc.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.json
