#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py > gpurun_out/bench_r04e.json 2> gpurun_out/bench_r04e.err
tail -c 300 gpurun_out/bench_r04e.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04e.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['verified'], d['configs_verified'])
for e in d['configs']:
    print("%-18s %-11s %8.1f GB/s frac %.3f  verified=%s  ms=%s" % (e['name'], e.get('kernel_family'), e.get('input_GBps',0), e.get('frac',0), e.get('verified'), e.get('ms_per_step')))
PY
