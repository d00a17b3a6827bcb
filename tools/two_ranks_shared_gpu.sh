#!/bin/bash
# N > 1 lines at HEAD with two ranks SHARING the one GPU of the box (gloo, host-staged exchange: a correctness record of the lines'
# content, not a timing): the default weak-scaled line, the strong-scaled one, the slab-pipelined one
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export JMAC_BENCH_SHARE_GPU=1
python bench.py --gpus 2 --steps 2 --warmup 1 --synth-scale 0.05 2>/dev/null | tail -1 > gpurun_out/r5_2ranks_weak.json
python bench.py --gpus 2 --steps 2 --warmup 1 --synth-scale 0.05 --scaling strong 2>/dev/null | tail -1 > gpurun_out/r5_2ranks_strong.json
python bench.py --gpus 2 --steps 2 --warmup 1 --synth-scale 0.05 --pipeline-chunks 4 2>/dev/null | tail -1 > gpurun_out/r5_2ranks_pipelined.json
for f in weak strong pipelined; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r5_2ranks_$f.json').read())
print('$f', d['n_gpus'], d['scaling'], round(d['ms_per_step'],1), d['config'].get('exchange'), d['comm'], 'model8', round(d['scaling_model']['predicted']['8']['step_ms'],2))"; done
