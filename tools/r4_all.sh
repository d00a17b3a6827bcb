#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r4_gpu_tests.log
cat gpurun_out/r4_gpu_tests.log
python tools/pair_probe.py --batched 1 --single 2>/dev/null
