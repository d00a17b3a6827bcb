#!/bin/bash
# sharded config-4 step at ONE rank: one-piece against the slab-pipelined path (what the partial passes + merge cost by themselves)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
python bench.py --workload synth-1m --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one-piece  ', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_bwd']['avg_launch_ms'], d['config'].get('exchange'))"
python bench.py --workload synth-1m --steps 3 --warmup 1 --pipeline-chunks 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined 4', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_bwd']['avg_launch_ms'], d['config'].get('exchange')); print(json.dumps(d['scaling_model']['predicted']['8']))"
JMAC_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 1 --warmup 1 --synth-scale 0.05 --pipeline-chunks 3 2>&1 | tail -1 | cut -c1-600
