# round 2: SQ counters of the lookup kernel on tools/lookup_one.py (tiled volumes, 3 maps x 32400 queries), the final kernel and -
# with VFML_LIB pointing at a library built from the previous flow_ops.hip - the one before the instruction diet
#     bash tools/exp/pmc_lookup.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-new}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lookup/${TAG}_s$i -- python3 tools/lookup_one.py > gpurun_out/pmc_lookup_${TAG}_s$i.log 2>&1 || { echo "set $i failed"; exit 1; }
done
python3 - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(f"gpurun_out/pmc_lookup/{tag}_s*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "corr_lookup_fixed_kernel<4, true" not in row["Kernel_Name"]:
            continue
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
print(tag, "per launch (split-row output):", {k: round(s / n) for k, (s, n) in sorted(acc.items())})
PY
