cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j15; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$set -o p --output-format csv -- python $R/tools/c5_operator_probe.py > $O/run_$set.txt 2>&1
  (cd $R && python tools/prof_summary.py $O/pmc_$set > $O/full_$set.txt 2>&1; awk '/^# PMC/{p=1} p' $O/full_$set.txt | grep -A 3 "atb_f16_kernel<2, true\|axb_f16_dma_kernel\|hilbert_fft_kernel\|colstats_tr_kernel\|^# PMC" > $O/c5_pmc_$set.txt)
  rm -rf $O/pmc_$set
done
cat $O/c5_pmc_FETCH_SIZE.txt $O/c5_pmc_WRITE_SIZE.txt | head -60
