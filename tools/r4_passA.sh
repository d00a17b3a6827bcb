#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_layer.py tests/test_gpu_ja_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_determinism.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -2
python tools/pair_probe.py --ja 2>&1 | tail -1
python tools/pair_probe.py --batched 1 2>&1 | tail -1
python tools/ja_sweep.py ja-real 2>&1 | tail -4
python tools/agg_sweep.py 1.0 auto 300 1 2>&1 | tail -3
