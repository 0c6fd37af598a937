# rocprofv3 kernel trace of the SimpleGridworld lane rollout (65 536 x 9 x 9 'default') at 1, 16 and 64 steps per launch: the
# kernels' own durations and the gaps between them
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for T in 1 16 64; do
  rm -rf $R/gpurun_out/tr_gw_$T
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_gw_$T -o t -- python3 $R/tools/gridworld_trace.py default $T -1 0 > /dev/null 2>&1
  echo "== $T steps per launch"
  python3 - "$R/gpurun_out/tr_gw_$T/t_kernel_trace.csv" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'wurm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-12:]
for a, b in zip(rows[:-1], rows[1:]):
    pass
for i, r in enumerate(rows):
    gap = int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']) if i else 0
    print(f"  {r['Kernel_Name'][:60]:60s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us   gap before {gap / 1e3:6.1f} us")
P
done
