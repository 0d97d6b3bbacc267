# usage: bash tools/diagnostics/gpu_profile.sh <tag> [bench args...]   (run on the GPU box through gpurun)
# kernel-trace stats and three PMC passes (SQ, FETCH_SIZE, WRITE_SIZE: separate passes as MI355X_MICROARCH.md prescribes) of the
# same bench command; tools/diagnostics/summarize_profile.py turns the result into profiles/<tag>/ + profiles/hbm_traffic.json
set -e
TAG=${1:-r1}; shift || true
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT          # stale runs of the same tag would be picked up by the summariser
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || (tail -20 $OUT/trace.err; exit 1)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || (tail -20 $OUT/pmc_sq.err; exit 1)
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || (tail -20 $OUT/pmc_fetch.err; exit 1)
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || (tail -20 $OUT/pmc_write.err; exit 1)
find $OUT -name "*.csv" | head -40
du -sh $OUT
