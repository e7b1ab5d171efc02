#!/bin/bash
# PMC passes over the fused Hilbert kernel (one pass per counter group; --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-8000}; P=${2:-1036800}
mkdir -p $R/gpurun_out/hfft_pmc
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/hfft_pmc/$tag -o out --output-format csv -- $R/build/hfft_probe $N $P 1 0 1 > /dev/null 2>&1
  f=$(find $R/gpurun_out/hfft_pmc/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "hilbert_fft" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
for k,v in acc.items(): print(f"{k}: {v/cnt[k]:.4g} per launch ({cnt[k]} launches)")
PY
done
