#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python tools/union_agg_probe.py 2>/dev/null
python tools/pair_probe.py --batched 1 2>/dev/null
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4_gpu_tests_b.log
cat gpurun_out/r4_gpu_tests_b.log
