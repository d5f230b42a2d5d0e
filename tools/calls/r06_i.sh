#!/bin/bash
# round 6, call I: the whole GPU suite + the default bench line on the re-tuned table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06_i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 3000 $O/bench_n1.json
