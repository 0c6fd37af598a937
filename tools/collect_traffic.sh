#!/bin/bash
# HBM traffic of the env kernels only (steps 3-4 of tools/collect_profiles.sh): bash tools/collect_traffic.sh <tag>
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/tools/traffic_workload.py > $OUT/pmc_$C.log 2>&1
done
python3 $R/tools/parse_pmc.py $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv') > $OUT/${TAG}_pmc_fetch_write_summary.json
python3 $R/tools/make_traffic_json.py $OUT/${TAG}_pmc_fetch_write_summary.json > $OUT/hbm_traffic.json
python3 $R/tools/trace_summary.py $(find $OUT/pmc_WRITE_SIZE -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_kernel_times_traffic_workload.txt
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
