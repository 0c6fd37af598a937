R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_lane_step.py tests/test_full_size_parity.py -m gpu -q -x 2>&1 | tail -4
python3 tools/bench_configs.py cfg3 2>&1 | grep -o '"cfg3[^"]*"\|"us_per_batch_step": [0-9.]*'
bash tools/pmc_percall.sh
python3 - <<PY
import json
d=json.load(open("/root/repo/gpurun_out/pmc_percall/summary.json"))
for k,v in d.items():
    if "lane_step" in k:
        for c,x in sorted(v.items()): print(c, round(x["mean"]/4096,1))
PY
