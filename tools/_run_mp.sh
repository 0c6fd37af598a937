R=$GRAFT_REPO_ROOT; cd $R
mkdir -p $R/gpurun_out/mp; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/mp/tr -o t -- python3 $R/tools/multi_percall.py 4096 10 36 speeds check noobs > $R/gpurun_out/mp/log.txt 2>&1
grep "us per" $R/gpurun_out/mp/log.txt
python3 $R/tools/trace_summary.py $(find $R/gpurun_out/mp/tr -name "*kernel_trace.csv" | head -1) | grep -E "check|step|reset"
rm -rf $R/gpurun_out/mp/tr
