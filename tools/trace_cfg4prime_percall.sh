cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_trace_percall -o t -- python3 $R/tools/cfg4prime_probe.py --percall 200 > $R/gpurun_out/r06_trace_percall.log 2>&1
find $R/gpurun_out/r06_trace_percall -name "*kernel_stats.csv" | head -1 | xargs head -12
