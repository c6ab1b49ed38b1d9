#!/bin/bash
# GPU box: the driver's round-end sequence at HEAD - full GPU suite, smoke, default bench
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03n
mkdir -p $out
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
cp gpurun_out/parity_observed.json $out/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout -k 10 500 python bench.py ) > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/bench_default.json; tail -4 $out/bench.log | cut -c1-300
