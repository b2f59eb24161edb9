#!/bin/bash
# steady-state kernel profile of bench.py
#   gpurun -- 'bash devtools/prof_bench_steady.sh TAG [bench args]'
TAG=${1:-steady}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd $REPO
rm -rf /tmp/pb_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$TAG -- python3 bench.py --steps 40 --warmup 5 --sustain 0 --no-cpu-baseline --no-model-roofline --strict-steps 0 "$@" > gpurun_out/${TAG}_bench.log 2>&1
f=$(ls /tmp/pb_$TAG/*/*_kernel_stats.csv | head -1)
head -40 "$f" > gpurun_out/${TAG}_kernel_stats.csv
k=$(ls /tmp/pb_$TAG/*/*_kernel_trace.csv | head -1)
python3 devtools/trace_by_grid.py "$k" "" 0.5 > gpurun_out/${TAG}_by_grid.txt
python3 devtools/trace_step_sequence.py "$k" > gpurun_out/${TAG}_step_sequence.txt 2>&1
grep '"metric"' gpurun_out/${TAG}_bench.log | cut -c1-200
