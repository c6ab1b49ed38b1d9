#!/bin/bash
# GPU box, round 4 first call: the new parity tests, the capture probe, one default bench line
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04a
mkdir -p $out
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_frame.py tests/test_bench_gpu.py -x -q -m gpu > $out/frame_tests.log 2>&1; echo "frame tests rc=$?"; tail -3 $out/frame_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_nets.py -x -q -m gpu -k "full_size_batch_against or bf16_path" > $out/nets_tests.log 2>&1; echo "nets tests rc=$?"; tail -3 $out/nets_tests.log
cp gpurun_out/parity_observed.json $out/parity_observed.json 2>/dev/null
timeout -k 10 900 python tools/graph_capture_probe.py > $out/graph_probe.log 2>&1; echo "probe rc=$?"; cat $out/graph_probe.log | cut -c1-600
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 3000 $out/bench.json
