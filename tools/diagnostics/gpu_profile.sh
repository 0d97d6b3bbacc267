# usage: bash tools/diagnostics/gpu_profile.sh <tag> [bench args...]   (run on the GPU box through gpurun)
# kernel-trace stats and four PMC passes (SQ, FETCH_SIZE, WRITE_SIZE: separate passes as MI355X_MICROARCH.md prescribes) of the
# same bench command; tools/diagnostics/summarize_profile.py turns the result into profiles/<tag>/ + profiles/hbm_traffic.json
set -e
TAG=${1:-r1}; shift || true
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT          # stale runs of the same tag would be picked up by the summariser
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --batched-extra off "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || (tail -20 $OUT/trace.err; exit 1)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --batched-extra off "$@" > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || (tail -20 $OUT/pmc_sq.err; exit 1)
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --batched-extra off "$@" > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || (tail -20 $OUT/pmc_fetch.err; exit 1)
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --batched-extra off "$@" > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || (tail -20 $OUT/pmc_write.err; exit 1)
# matrix-pipe occupancy (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_bf16; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --batched-extra off "$@" > $OUT/bench_pmc_mfma.json 2> $OUT/pmc_mfma.err || (tail -20 $OUT/pmc_mfma.err; exit 1)
# the shader clock the chip holds inside the scan: the -DRT_SOLO_STAMPS=3 build (make clock), 2+ s of back-to-back frames, no profiler
CFGNAME=C2; for a in "$@"; do case $a in C1|C2|C4|C5) CFGNAME=$a;; esac; done
if [ -f $R/raytracer.glsl_amd/librtgl_amd_clock.so ]; then
  (cd $R && RTGL_AMD_LIB=$R/raytracer.glsl_amd/librtgl_amd_clock.so python3 tools/diagnostics/solo_frames.py $CFGNAME ${CLOCK_FRAMES:-600} > $OUT/clock.txt 2>&1) || tail -5 $OUT/clock.txt
  grep "rtgl clock:" $OUT/clock.txt || true
fi
find $OUT -name "*.csv" | head -40
du -sh $OUT
