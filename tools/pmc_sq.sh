#!/bin/bash
# SQ-level counters for the hot kernels (two passes of <= 8 SQ counters each).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-sq}
mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAVES"
timeout 600 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -o p1 -- python bench.py --steps 2 --warmup 1 --no-cpu --no-large --spmv-steps 2 > /dev/null 2> $OUT/p1.err
timeout 600 rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -o p2 -- python bench.py --steps 2 --warmup 1 --no-cpu --no-large --spmv-steps 2 > /dev/null 2> $OUT/p2.err
python - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("k_num_","k_sym_","k_spmv_amb","k_row_prod","k_b_minmax")): continue
            k = k.split("(")[0].replace("void nsp::spgemm::","").replace("void nsp::spmv::","")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()):
            print("   %-24s %14.0f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
