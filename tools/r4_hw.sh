#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_layer.py tests/test_gpu_ja_oracle.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8
for HW in 0 1; do
JMAC_FWD_HW=$HW python tools/union_agg_probe.py 2>/dev/null
done
JMAC_FWD_HW_GP=2 python tools/union_agg_probe.py 2>/dev/null
