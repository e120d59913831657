for lib in default pq pp0; do
  if [ "$lib" = "default" ]; then unset VBNN_HIP_LIB; else export VBNN_HIP_LIB=/root/repo/vbnn_amd/lib/$lib/libvbnn_hip.so; fi
  for k in "gemm_nt_v3+EpiFwd+Lb1ELb0ELb0E" "gemm_nt_v3+EpiDx" "gemm_nt_v3+EpiDwLb0ELb0"; do
    echo "== $lib $k"
    python3 tools/pmc_kernel.py "$k" GRBM_GUI_ACTIVE,SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES -- python3 /root/repo/bench.py --no-cpu-baseline --no-reporting-config --no-deep-config --no-train-step --steps 10 --warmup 5 --repeats 1 2>&1 | tail -3
  done
done
