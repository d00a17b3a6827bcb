#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_union_real.py -x -q 2>&1 | tail -12
python bench.py --steps 20 --no-synth > gpurun_out/r4_bench_a.json 2> gpurun_out/r4_bench_a.err; tail -3 gpurun_out/r4_bench_a.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r4_bench_a.json").read().strip().splitlines()[-1])
print("ms_per_step", l["ms_per_step"], "value", l["value"])
print("roofline", l["roofline"]["frac"], "bwd", l["roofline_bwd"]["frac"])
print("pair", json.dumps(l.get("pair"))[:1500])
u = l.get("union", {})
print("union", json.dumps({k: v for k, v in u.items() if k not in ("cpu",)})[:2500])
PY
