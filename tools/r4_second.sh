#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r4_union_agg.jsonl
for S in 16384 65536; do
JMAC_SMALL_ITEMS=$S python tools/union_agg_probe.py >> gpurun_out/r4_union_agg.jsonl 2>gpurun_out/r4_union_agg.err
done
JMAC_SMALL_ITEMS=65536 JMAC_FWD_U=4 python tools/union_agg_probe.py >> gpurun_out/r4_union_agg.jsonl 2>>gpurun_out/r4_union_agg.err
cat gpurun_out/r4_union_agg.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4_gpu_tests_a.log
cat gpurun_out/r4_gpu_tests_a.log
