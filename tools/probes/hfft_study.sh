#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hfft_study
for cfg in "8000 1036800" "4000 1036800" "2000 1036800" "1000 1036800" "500 1036800"; do
  set -- $cfg
  echo "=== n=$1 p=$2"
  $R/build/hfft_probe $1 $2 1 0 3 | grep kernel
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rm -rf $R/gpurun_out/hfft_study/$tag
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/hfft_study/$tag -o out --output-format csv -- $R/build/hfft_probe $1 $2 1 0 1 > /dev/null 2>&1
    f=$(find $R/gpurun_out/hfft_study/$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "hilbert_fft" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
for k,v in acc.items(): print(f"   {k}: {v/cnt[k]:.4g}")
PY
    rm -rf $R/gpurun_out/hfft_study/$tag
  done
done
