# SQ / instruction-cache counters of the MultiSnake per-call step at cfg4' (multi_step_kernel, training dynamics, partial_5;
# the same process also launches the fused rollout: parse_pmc.py keys by kernel), one rocprofv3 --pmc pass per counter group:
#   bash tools/pmc_multi_percall.sh  ->  gpurun_out/pmc_multi_percall/summary.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_multi_percall
rm -rf $OUT
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/cfg4prime_probe.py --percall 100 > $OUT/p$i.log 2>&1
done
python3 $R/tools/parse_pmc.py $(find $OUT -name '*counter_collection.csv') > $OUT/summary.json
rm -rf $OUT/p?
python3 - <<'P'
import json, os
d = json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out/pmc_multi_percall/summary.json')))
for k, v in d.items():
    if 'multi_' in k:
        print(k[:100])
        print('   ', {c: round(x['mean']) for c, x in v.items()})
P
