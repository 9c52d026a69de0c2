cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    VFML_TAPX=$v rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcab/t${v}_$c -- python3 tools/conv_one.py > gpurun_out/pmcab_t${v}_$c.log 2>&1 || exit 1
  done
  python3 tools/pmc_traffic.py gpurun_out/pmcab/t${v}_FETCH_SIZE gpurun_out/pmcab/t${v}_WRITE_SIZE gpurun_out/pmcab_t$v.json "conv_one" | grep -i "conv_gemm"
done
