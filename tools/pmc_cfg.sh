#!/bin/bash
# SQ counters for the kernels of one configs-runner case:  bash tools/pmc_cfg.sh rmat18 k_num_tiled
export TMPDIR=/tmp
CASE=${1:-rmat16}; PAT=${2:-k_num_tiled}
OUT=$PWD/gpurun_out/pmc_$CASE; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAVES"
timeout 900 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -o p1 -- python tools/one_call_cfg.py $CASE > /dev/null 2> $OUT/p1.err
timeout 900 rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -o p2 -- python tools/one_call_cfg.py $CASE > /dev/null 2> $OUT/p2.err
python - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "$PAT" not in k: continue
            agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-24s %16.0f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
