#!/bin/bash
# N > 1 lines at HEAD with two ranks SHARING the one GPU of the box (gloo, host-staged exchange: a correctness record of the lines'
# content, not a timing): the default weak-scaled line, the strong-scaled one, the slab-pipelined one; plus the one-rank sharded
# workload (--workload synth-1m).  Each run's LAST stdout line is the compact record (<= 4 KB), the full one is bench_full.json.
#   -> gpurun_out/r6_bench_2ranks_shared_gpu.jsonl (compact lines) + gpurun_out/r6_2ranks_*_full.json
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out; : > gpurun_out/r6_bench_2ranks_shared_gpu.jsonl
run() {  # tag args...
  tag=$1; shift
  JMAC_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 2 --warmup 1 --synth-scale 0.05 "$@" 2> gpurun_out/r6_2ranks_$tag.err | tail -1 > gpurun_out/r6_2ranks_$tag.json
  cat gpurun_out/r6_2ranks_$tag.json >> gpurun_out/r6_bench_2ranks_shared_gpu.jsonl
  cp bench_full.json gpurun_out/r6_2ranks_${tag}_full.json 2>/dev/null
  python3 -c "
import json
c=json.loads(open('gpurun_out/r6_2ranks_$tag.json').read()); d=json.load(open('gpurun_out/r6_2ranks_${tag}_full.json'))
print('$tag', len(json.dumps(c)), 'bytes;', c['n_gpus'], c['scaling'], round(c['ms_per_step'],1), c['config'].get('exchange'), c['highlights'].get('comm.backend'), c['highlights'].get('comm.world_seen_by_all_reduce'), 'model8', round(d['scaling_model']['predicted']['8']['step_ms'],2))"
}
run weak
run strong --scaling strong
run pipelined --pipeline-chunks 4
python bench.py --workload synth-1m --steps 2 --warmup 1 --synth-scale 0.05 2> gpurun_out/r6_synth1m.err | tail -1 > gpurun_out/r6_synth1m_line.json
python3 -c "
import json; c=json.loads(open('gpurun_out/r6_synth1m_line.json').read()); print('synth-1m (one rank)', len(json.dumps(c)), 'bytes;', c['n_gpus'], round(c['ms_per_step'],2), c['roofline']['frac'])"
