#!/bin/bash
# Collects the rocprofv3 evidence for one workload on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh c3 r1
# Outputs go to gpurun_out/<tag>_<workload>/; profiles/summarize.py turns them into the committed summary.
# Counters are collected in separate passes (kernel trace alone; FETCH_SIZE, WRITE_SIZE, SQ, TCC each alone).
set -e
W=${1:-c3}; TAG=${2:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/${TAG}_$W
rm -rf $P
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py --workload $W --steps 3 --warmup 1 > $P/bench.json 2> $P/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $P/trace.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $P/fetch.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $P/write.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $P/sq -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $P/sq.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/tcc -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $P/tcc.out 2>&1
echo collected $P
