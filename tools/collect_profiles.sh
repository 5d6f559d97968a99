#!/bin/bash
# Round evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r04 v1
# -> gpurun_out/<round>prof/: bench line (plain run), kernel stats of the same command under rocprofv3 --kernel-trace --stats,
#    four PMC passes over tools/pmc_target.py (FETCH_SIZE / WRITE_SIZE / SQ + GRBM / MFMA + cache counters) and the traffic summary.
# The program stands directly behind `--` (no env / bash -c hop: the profiler has initialised the GPU by then).
set -o pipefail
RND=${1:-r04}; VER=${2:-v1}
ROOT=$PWD
OUT=$ROOT/gpurun_out/${RND}prof
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 3 > $OUT/${RND}_bench_${VER}.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${RND}_bench_${VER}_profiled.json 2> $OUT/bench_prof.err
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/${RND}_bench_kernel_stats_${VER}.csv
for pass in "fetch_size:FETCH_SIZE" "write_size:WRITE_SIZE" "sq:GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "sq2:SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_ANY TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf /tmp/prof_pmc && rocprofv3 --pmc $ctrs --output-format csv -d /tmp/prof_pmc -- python3 $ROOT/tools/pmc_target.py > /dev/null 2> $OUT/pmc_${name}.err
    cp $(find /tmp/prof_pmc -name "*counter_collection.csv" | head -1) $OUT/${RND}_pmc_${name}.csv
done
cd $ROOT
ls -la $OUT | tail -12
