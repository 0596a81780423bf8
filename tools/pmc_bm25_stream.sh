#!/bin/bash
# PMC passes over the BM25 stream kernel alone (tools/bm25_bench.py ... stream-only): instruction mix, waits, LDS, TLB.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_bm25_stream}
N=${2:-10000000}
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -io "[A-Z_0-9]*UTCL[A-Z_0-9]*\|[A-Z_0-9]*TLB[A-Z_0-9]*" | sort -u | head -40 > $OUT/tlb_counters.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/tools/bm25_bench.py $N 3 64 stream-only > $OUT/pass$i.log 2>&1
  echo "pass $i ($C): exit $?"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/pass*/")):
    fs=glob.glob(d+"*/*counter_collection.csv")
    if not fs: print(d,"no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0][-30:]
        if "bm25_stream" not in k: continue
        key=(k, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X",""))
        agg[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key,r["Counter_Name"])]+=1
    for key,cs in agg.items():
        print(key, {c: round(v/n[(key,c)]) for c,v in cs.items()})
PY
