cd $GRAFT_REPO_ROOT
echo "== default dw"; python tools/bench_gemm_dw_mix.py > gpurun_out/mix_a.txt 2>&1; echo rc=$?; tail -12 gpurun_out/mix_a.txt
echo "== dw <= 64 VGPRs (spills)"; EPOS_HIP_LIB=/root/repo/epos_amd/lib/libepos_hip_dwm8.so python tools/bench_gemm_dw_mix.py > gpurun_out/mix_b.txt 2>&1; echo rc=$?; tail -6 gpurun_out/mix_b.txt
