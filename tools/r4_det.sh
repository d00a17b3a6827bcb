#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_determinism.py tests/test_gpu_losses.py tests/test_gpu_e2e.py tests/test_gpu_pair.py tests/test_gpu_model.py -x -q -s 2>&1 | grep -E "passed|failed|Error|max\|err|assert |HIP replay|fused eval" | head -20
python tools/pair_probe.py --batched 1 --single 2>/dev/null
