# round 2: SQ counters of the final per-tap kernel and of the shared-stage kernel on tools/conv_one.py (1x5 gate shape)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 2; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    VFML_TAPX=$v timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r02/t${v}_s$i -- python3 tools/conv_one.py > gpurun_out/pmc_r02_t${v}_s$i.log 2>&1 || echo "set $v/$i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections
for v in (0, 2):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(f"gpurun_out/pmc_r02/t{v}_s*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "conv_gemm" not in row["Kernel_Name"]:
                continue
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    print("VFML_TAPX =", v, {k: round(s / 5) for k, (s, n) in sorted(acc.items())})
PY
