# rocprofv3 --kernel-trace summaries of the rotation probe and of the bootstrap probe (config 4 size) -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02c_rot -o p --output-format csv -- python $R/tools/rot_probe.py > $R/gpurun_out/r02c_rot.log 2>&1
N=10000 NLAT=720 NLON=1440 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02c_boot -o p --output-format csv -- python $R/tools/bootstrap_probe.py > $R/gpurun_out/r02c_boot.log 2>&1
cd $R
(grep -E "gpu promax|rot_step" gpurun_out/r02c_rot.log; echo; python tools/prof_summary.py gpurun_out/r02c_rot | head -14) > gpurun_out/r02c_rotation_probe_kernel_trace_summary.txt 2>&1
(grep -E "member|row norms|driver" gpurun_out/r02c_boot.log; echo; python tools/prof_summary.py gpurun_out/r02c_boot | head -24) > gpurun_out/r02c_bootstrap_probe_kernel_trace_summary.txt 2>&1
cut -c1-170 gpurun_out/r02c_rotation_probe_kernel_trace_summary.txt | head -12; cut -c1-170 gpurun_out/r02c_bootstrap_probe_kernel_trace_summary.txt | head -30
