# round 3: SQ counters of the ONE-MFMA kernels (64-channel steps, NM = 5) on the 1080p 1x5 gate shape - the shared-stage kernel
# the dispatcher picks (VFML_TAPX=1) and the per-tap LDS-DMA kernel (VFML_TAPX=0) - with the three-MFMA per-tap kernel beside
# them for scale.  Separate rocprofv3 --pmc passes per counter set, five launches each (tools/conv_one.py).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r03
for cfg in "1 1" "1 0" "3 0"; do
  set -- $cfg; m=$1; v=$2
  i=0
  for cs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    MB_MFMA=$m VFML_TAPX=$v timeout -k 10 150 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r03/m${m}_t${v}_s$i -- python3 tools/conv_one.py > gpurun_out/pmc_r03_m${m}_t${v}_s$i.log 2>&1 || echo "set $m/$v/$i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections
for m, v in ((1, 1), (1, 0), (3, 0)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    name = ""
    for path in glob.glob(f"gpurun_out/pmc_r03/m{m}_t{v}_s*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "conv_gemm" not in row["Kernel_Name"]:
                continue
            name = row["Kernel_Name"][:90]
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    print(f"MB_MFMA={m} VFML_TAPX={v} {name}", {k: round(s / 5) for k, (s, n) in sorted(acc.items())})
PY
