#!/bin/bash
# Memory-pipeline counters for the kernels of one configs-runner case:  bash tools/pmc_mem.sh rmat18 k_num_tiled
export TMPDIR=/tmp
CASE=${1:-rmat16}; PAT=${2:-k_num_tiled}
OUT=$PWD/gpurun_out/pmcmem_$CASE; mkdir -p $OUT
P1="TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"
P2="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum"
P3="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LEVEL_WAVES TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"
P4="TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o p$i -- python tools/one_call_cfg.py $CASE > /dev/null 2> $OUT/p$i.err
done
python - <<PY
import csv, glob, collections
for p in ("p1","p2","p3","p4"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "$PAT" not in k: continue
            agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            print("   %-36s %16.0f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
