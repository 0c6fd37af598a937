#!/bin/bash
# One gpurun call of a build -> measure iteration: GPU test suite, bench line, kernel trace of the per-call loop.
# usage (from the repo root, through gpurun): bash tools/gpu_round.sh <tag> [pytest args]
TAG=${1:-a}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/round_$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 "$@" > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/tools/percall_only.py 512 > $OUT/trace.log 2>&1
python3 $R/tools/trace_summary.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/percall_512_kernels.txt 2>&1
cat $OUT/percall_512_kernels.txt
rm -rf $OUT/trace
python3 -c "
import json
d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])
for k,v in d.get('extra',{}).items(): print(k, v['value'], v.get('us_per_batch_step'), v.get('ms_per_launch'))
print(json.dumps(d.get('cpu_baseline'))[:600])
"
