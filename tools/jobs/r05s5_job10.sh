cd $GRAFT_REPO_ROOT; O=gpurun_out/j10; mkdir -p $O
python tools/wide_small_probe.py > $O/default.txt 2>&1
for parts in 2 6 12 24; do EOFX_GRAM_PARTS=$parts python tools/wide_small_probe.py 2>&1 | grep panel_gram > $O/parts_$parts.txt; done
EOFX_PMM_KW=128 python tools/wide_small_probe.py 2>&1 | grep panel_matmul > $O/kw128.txt
grep panel $O/default.txt; for f in $O/parts_*.txt $O/kw128.txt; do echo $f; cat $f; done
