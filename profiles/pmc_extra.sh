#!/bin/bash
# Extra stall-analysis PMC passes for one workload (run through gpurun from the repo root):
#   bash profiles/pmc_extra.sh c4 x1
set -e
W=${1:-c4}; TAG=${2:-x1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/gpurun_out/${TAG}_$W
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $P/p1 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $P/p1.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_SMEM --output-format csv -d $P/p2 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $P/p2.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_BUSY_CU_CYCLES --output-format csv -d $P/p3 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $P/p3.out 2>&1
python3 - $P <<'PY'
import sys, glob, csv, collections, os
P = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob(os.path.join(P, "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "episode" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(P, "extra.txt"), "w") as o:
    for k in sorted(agg):
        o.write("%-28s %.4e\n" % (k, agg[k]))
print(open(os.path.join(P, "extra.txt")).read())
PY
