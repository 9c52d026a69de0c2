#!/bin/bash
# The profile set of a round's final library (one GPU box, back to back): bench lines of every workload, rocprofv3 kernel
# stats of the headline command (as shipped, and with the encoder prefetch off so that no launch shares the chip), PMC traffic.
#     gpurun --timeout 1200 -- 'bash tools/final_profiles.sh r02_f'
# Outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
step() { echo "== $1" | tee -a $OUT/progress.log; }
step "bench (default)";   timeout -k 10 500 python3 bench.py > $OUT/bench.log 2>&1 && tail -1 $OUT/bench.log > $OUT/${TAG}_bench.json || exit 1
step "bench f16x3";       timeout -k 10 300 python3 bench.py --precision f16x3 --no-cpu-baseline --no-jobs > $OUT/bench_f16x3.log 2>&1 && tail -1 $OUT/bench_f16x3.log > $OUT/${TAG}_bench_f16x3.json || exit 1
step "bench corr16";      timeout -k 10 300 python3 bench.py --corr-volume f16 --no-cpu-baseline --no-jobs > $OUT/bench_corr16.log 2>&1 && tail -1 $OUT/bench_corr16.log > $OUT/${TAG}_bench_corr16.json || exit 1
step "bench corr32";      timeout -k 10 300 python3 bench.py --corr-volume f32 --no-cpu-baseline --no-jobs > $OUT/bench_corr32.log 2>&1 && tail -1 $OUT/bench_corr32.log > $OUT/${TAG}_bench_corr32.json || exit 1
step "bench prefetch off"; VFML_PREFETCH=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-jobs > $OUT/bench_pf0.log 2>&1 && tail -1 $OUT/bench_pf0.log > $OUT/${TAG}_bench_prefetch_off.json || exit 1
step "bench memflow";     timeout -k 10 300 python3 bench.py --workload memflow1080p --steps 18 --no-cpu-baseline > $OUT/bench_memflow.log 2>&1 && tail -1 $OUT/bench_memflow.log > $OUT/${TAG}_memflow1080p_bench.json || exit 1
step "bench bof720p";     timeout -k 10 300 python3 bench.py --workload bof720p --steps 32 --no-cpu-baseline > $OUT/bench_bof.log 2>&1 && tail -1 $OUT/bench_bof.log > $OUT/${TAG}_bof720p_bench.json || exit 1
step "bench 4k tile";     timeout -k 10 300 python3 bench.py --workload mof4k-tile --steps 12 --warmup 4 --no-cpu-baseline > $OUT/bench_4k.log 2>&1 && tail -1 $OUT/bench_4k.log > $OUT/${TAG}_mof4k_tile_bench.json || exit 1
step "rocprofv3 stats (as shipped)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-jobs > $OUT/prof.log 2>&1 || exit 1
grep "^{\"metric\"" $OUT/prof.log | tail -1 > $OUT/${TAG}_bench_profiled.json
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_mixed_kernel_stats.csv
step "rocprofv3 stats (prefetch off)"
VFML_PREFETCH=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof0 -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-jobs > $OUT/prof0.log 2>&1 || exit 1
grep "^{\"metric\"" $OUT/prof0.log | tail -1 > $OUT/${TAG}_bench_profiled_prefetch_off.json
cp $(find $OUT/prof0 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_mixed_prefetch_off_kernel_stats.csv
step "pmc FETCH_SIZE"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-jobs > $OUT/pmc_fetch.log 2>&1 || exit 1
step "pmc WRITE_SIZE"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-jobs > $OUT/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline" >> $OUT/progress.log 2>&1
rm -rf $OUT/prof $OUT/prof0 $OUT/pmc_fetch $OUT/pmc_write
step done
ls -la $OUT
