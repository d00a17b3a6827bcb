#!/bin/bash
cd /root/repo/_r3
for i in 1 2; do python -m pytest "tests/test_gpu_ja_oracle.py" -x -q -k "256 and train and real" 2>&1 | grep -E "passed|failed|max\|err" | tail -3; done
cd /root/repo
python -m pytest "tests/test_gpu_ja_oracle.py" -x -q -k "256 and train and real" 2>&1 | grep -E "passed|failed|max\|err" | tail -3
JMAC_SMALL_ITEMS=16384 python -m pytest "tests/test_gpu_ja_oracle.py" -x -q -k "256 and train and real" 2>&1 | grep -E "passed|failed|max\|err" | tail -3
