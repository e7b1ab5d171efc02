#!/bin/bash
# round 4 job M: timeline of one config-3 MCA call (TSC, in place, sketch joined lazily by the engine)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m; mkdir -p $O
TSC=1 LAYOUT=inplace timeout 600 rocprofv3 --kernel-trace -d $O/mca -o p --output-format csv -- python $R/tools/mca_timeline.py > $O/mca.log 2>&1
(cd $R && python tools/trace_gaps.py $O/mca panel_import_kernel > $O/mca_timeline.txt 2>&1)
rm -rf $O/mca
tail -2 $O/mca_timeline.txt
awk 'NR>2 && $3>60' $O/mca_timeline.txt | head -12
