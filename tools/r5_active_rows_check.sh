cd $GRAFT_REPO_ROOT
python tools/r5_rowrange_graph_probe.py > gpurun_out/r5_rowrange_probe.txt 2>&1
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_ja_oracle.py tests/test_gpu_determinism.py tests/test_gpu_e2e.py tests/test_gpu_model.py tests/test_gpu_pair.py tests/test_gpu_config1.py tests/test_gpu_harness.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5_ar_tests.log
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/r5_step_ja_ar; mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof -o p -- python3 $R/tools/pair_probe.py --ja > $R/$OUT/probe.json 2> $R/$OUT/prof.err
cd $R
python3 tools/step_breakdown.py $OUT/prof/p_kernel_trace.csv 70 > $OUT/step_breakdown.txt 2>&1
python3 tools/step_timeline.py $OUT/prof/p_kernel_trace.csv > $OUT/timeline.txt 2>&1
rm -f $OUT/prof/p_kernel_trace.csv
cat gpurun_out/r5_rowrange_probe.txt | grep -v amdgpu; cat gpurun_out/r5_ar_tests.log; head -40 $OUT/step_breakdown.txt; cat $OUT/probe.json; tail -3 $OUT/timeline.txt
