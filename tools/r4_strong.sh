#!/bin/bash
cd /root/repo
python bench.py --workload synth-1m --scaling strong --synth-scale 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_strong_n1.json
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r4_strong_n1.json").read())
print(l["scaling"], l["ms_per_step"], l["value"], l["config"]["workload"][:120])
print(json.dumps(l["scaling_model"]["predicted"]))
PY
JMAC_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --scaling strong --synth-scale 0.1 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-synth 2>gpurun_out/r4_strong_n2.err | tail -1 > gpurun_out/r4_strong_n2.json
python - <<'PY'
import json
try:
    l = json.loads(open("gpurun_out/r4_strong_n2.json").read())
    print("N=2 shared GPU:", l["scaling"], l["n_gpus"], l["ms_per_step"], l["config"]["workload"][:100], l["comm"])
except Exception as ex:
    print("N=2 failed", ex); print(open("gpurun_out/r4_strong_n2.err").read()[-1500:])
PY
