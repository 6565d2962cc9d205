#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r04d.json 2> gpurun_out/bench_r04d.err
tail -c 400 gpurun_out/bench_r04d.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04d.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['verified'], d['configs_verified'])
for e in d['configs']:
    print("%-18s %-11s %8.1f GB/s frac %.3f  verified=%s  ms=%s" % (e['name'], e.get('kernel_family'), e.get('input_GBps',0), e.get('frac',0), e.get('verified'), e.get('ms_per_step')))
PY
