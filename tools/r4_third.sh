#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r4_union_agg2.jsonl
for SB in 16384 65536; do
JMAC_SMALL_BWD_ITEMS=$SB python tools/union_agg_probe.py >> gpurun_out/r4_union_agg2.jsonl 2>gpurun_out/r4_union_agg.err
done
cat gpurun_out/r4_union_agg2.jsonl
for SB in 16384 65536; do
JMAC_SMALL_BWD_ITEMS=$SB python tools/pair_probe.py --batched 1 2>>gpurun_out/r4_union_agg.err
done
python -m pytest tests/test_gpu_pair.py -x -q 2>&1 | tail -15
for i in 1 2 3; do python -m pytest "tests/test_gpu_ja_oracle.py" -x -q -k "256 and train and real" 2>&1 | tail -3; done
