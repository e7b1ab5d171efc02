# rocprofv3 --kernel-trace over the config-5 probe (Hilbert stage + complex rSVD at 8000 x 1 036 800): per-kernel table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02c_cplx}
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG -o p --output-format csv -- python $R/tools/complex_probe.py 8000 720 1440 20 > $R/gpurun_out/$TAG.log 2> $R/gpurun_out/$TAG.err
cd $R
(grep "^rep" gpurun_out/$TAG.log; echo; python tools/prof_summary.py gpurun_out/$TAG) > gpurun_out/${TAG}_summary.txt 2>&1
head -30 gpurun_out/${TAG}_summary.txt | cut -c1-200
