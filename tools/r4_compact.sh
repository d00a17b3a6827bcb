#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_layer.py tests/test_gpu_ja_oracle.py tests/test_gpu_pair.py tests/test_gpu_union_real.py tests/test_gpu_model.py tests/test_gpu_e2e.py tests/test_gpu_determinism.py tests/test_gpu_scoring.py -x -q 2>&1 | grep -E "passed|failed|Error|max\|err|assert " | head -12
python tools/pair_probe.py --batched 1 --single 2>/dev/null
