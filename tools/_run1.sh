#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r04c.json 2> gpurun_out/bench_r04c.err
tail -c 600 gpurun_out/bench_r04c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04c.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['cpu_baseline'])
for k,v in d.get('other_configs',{}).items():
    print(k, {kk:v[kk] for kk in v if kk in ('gbps_in','value','kernel','verified','verified_full','gbps','kernel_family','frac')})
PY
