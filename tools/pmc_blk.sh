#!/bin/bash
# SQ counters of the node-block numeric kernel on one configs-runner case (two passes of 8 counters):
#   bash tools/pmc_blk.sh <case> [tag]     -> gpurun_out/<tag>/  and a table on stdout
export TMPDIR=/tmp
CASE=${1:-cant}
OUT=$PWD/gpurun_out/${2:-pmc_blk}
mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAVES"
P3="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o p -- python tools/run_configs.py $CASE > /dev/null 2> $OUT/p$i.err
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_num_block" not in k: continue
        k = k.split("(")[0].replace("void nsp::spgemm::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
