#!/bin/bash
# (NSPARSE_SPMV_PIPE / NSPARSE_SPMV_PLAIN: the library must be built with EXTRA=-DNSPARSE_EXPERIMENTS, see csrc/Makefile)
# Counters of the AMB SpMV kernel on the nlpkkt-class matrix, round-1 form (NSPARSE_SPMV_PIPE=0 with the
# XCD remap) against the round-2 default: memory traffic (FETCH_SIZE / WRITE_SIZE in their own passes),
# L2 hits / misses, SQ wave cycles and wait cycles.  Output: gpurun_out/spmv_counters.json
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_spmv; mkdir -p $OUT
PASSES=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU")
for V in old new; do
  if [ $V = old ]; then ENVV="NSPARSE_SPMV_PIPE=0 NSPARSE_SPMV_REMAP=1"; else ENVV="NSPARSE_SPMV_PIPE=4"; fi
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1))
    env $ENVV timeout 600 rocprofv3 --pmc $P --output-format csv -d $OUT/$V$i -o p -- python tools/pmc_one.py spmv_hbm > /dev/null 2> $OUT/$V$i.err
  done
done
python - <<PY
import csv, glob, collections, json
res = {}
for V in ("old", "new"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s*/**/*counter_collection.csv" % V, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_spmv_amb" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[V] = {k: sum(v[len(v)//2:]) / len(v[len(v)//2:]) for k, v in agg.items()}
    res[V]["hbm_bytes_per_launch"] = res[V].get("FETCH_SIZE", 0) * 2048 + res[V].get("WRITE_SIZE", 0) * 1024
    if "TCC_HIT_sum" in res[V]:
        res[V]["l2_hit_rate"] = res[V]["TCC_HIT_sum"] / (res[V]["TCC_HIT_sum"] + res[V]["TCC_MISS_sum"])
res["note"] = "old = round-1 kernel form (unroll 4) with the XCD block remap; new = whole-row-in-flight kernel, natural block order; nlpkkt120-class stand-in, fp64; counters per launch (mean of the later half of the launches)"
json.dump(res, open("$PWD/gpurun_out/spmv_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
