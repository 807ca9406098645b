cd /root/repo
(echo "# scripts/ubench/valu_cost.hip (hipcc --offload-arch=gfx950 -O2): cycles per wave64 VALU instruction per SIMD"; timeout 60 ./build_exp/valu_cost) > gpurun_out/ubench_valu_cost.txt 2>&1
(echo "# scripts/ubench/step_loop.hip (hipcc --offload-arch=gfx950 -O3 -D...): the byte-table kernel's step loop in isolation, 256 workgroups x 16 waves, 2000 steps"; for v in base dyn dyn_noreads dyn_noadds dyn_addr0 dyn_filt2 dyn_d4 dyn_d12; do timeout 30 ./build_exp/sl_$v 2000 256; done) > gpurun_out/ubench_step_loop.txt 2>&1
