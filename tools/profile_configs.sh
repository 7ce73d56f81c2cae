#!/bin/bash
# Per-config profiles (verdict r02 item 1a): for every case, the kernels of ONE configs-runner call with
# the bins serialised (tools/one_call_cfg.py), as
#   1. rocprofv3 --kernel-trace --stats            -> per-kernel calls / average duration
#   2. rocprofv3 --pmc <SQ set 1>, <SQ set 2>      -> waits, instruction mix
#   3. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (own passes, MI355X_MICROARCH.md HBM section)
# and the whole-call time with the bins overlapped (tools/run_configs.py, no bin timing, oracle check).
# Usage (through gpurun): bash tools/profile_configs.sh <tag> case [case ...]
# Output: gpurun_out/<tag>/<case>.{stats.csv,pmc.json,call.json}; tools/summarize_configs.py -> profiles/
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"
for c in "$@"; do
  W=/tmp/pc_$c; rm -rf $W; mkdir -p $W
  NSPARSE_RUN_CHECK=${NSPARSE_RUN_CHECK:-1} timeout 600 python tools/run_configs.py $c > $OUT/$c.call.json 2> $OUT/$c.call.err
  timeout 600 python tools/one_call_cfg.py $c 3 > $OUT/$c.serial.json 2> $OUT/$c.serial.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/t -o t -- python tools/one_call_cfg.py $c 3 > /dev/null 2> $OUT/$c.t.err
  find $W/t -name "*kernel_stats.csv" -exec cp {} $OUT/$c.stats.csv \;
  i=0
  for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $P --output-format csv -d $W/p$i -o p -- python tools/one_call_cfg.py $c 2 > /dev/null 2> $OUT/$c.p$i.err
  done
  python - "$c" "$W" "$OUT" <<'PY'
import csv, glob, json, sys, collections, re
c, W, OUT = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(W + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("nsp::spgemm::", "")
        if not k.startswith("k_"): continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {cn: round(sum(v) / len(v), 1) for cn, v in d.items()} for k, d in agg.items()}
json.dump(out, open(f"{OUT}/{c}.pmc.json", "w"), indent=0, sort_keys=True)
PY
  rm -rf $W
  echo "== $c"; cat $OUT/$c.call.json | cut -c1-400; head -12 $OUT/$c.stats.csv | cut -c1-160
done
du -sh $OUT
