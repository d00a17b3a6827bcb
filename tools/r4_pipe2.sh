#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -3
bash tools/r4_pipe.sh
