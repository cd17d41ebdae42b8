# usage (on the GPU box, from the repo root): bash tools/refresh_profiles.sh <tag>      e.g. r03_a
# Writes under gpurun_out/: the default bench line, the rocprofv3 kernel stats of the same command (+ its bench line), the
# two HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) summarised by tools/pmc_traffic.py,
# and the MFMA-busy pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS..., GRBM_GUI_ACTIVE) summarised by tools/pmc_mfma.py.
# The program itself follows `--` (never env / bash -c: the profiler has initialised the GPU by then).
set -e
tag=${1:-r03_x}
out=gpurun_out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench_bf16.json 2> $out/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o p -- python3 bench.py --steps 20 --warmup 5 \
  --no-other-precision --no-cpu-baseline > $out/${tag}_bench_bf16_under_rocprofv3.json 2> $out/${tag}_prof.err
cp $out/${tag}_prof/p_kernel_stats.csv $out/${tag}_kernel_stats_bf16_pipelined.csv
echo "kernel stats (bf16) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof32 -o p -- python3 bench.py --steps 10 --warmup 3 \
  --dtype f32 --no-other-precision --no-cpu-baseline > $out/${tag}_bench_f32_under_rocprofv3.json 2> $out/${tag}_prof32.err
cp $out/${tag}_prof32/p_kernel_stats.csv $out/${tag}_kernel_stats_f32_pipelined.csv
echo "kernel stats (f32) done"
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-precision --no-graph --no-overlap"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_f -o f -- $cmd > /dev/null 2> $out/${tag}_pmc_f.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_w -o w -- $cmd > /dev/null 2> $out/${tag}_pmc_w.err
echo "WRITE_SIZE pass done"
python3 tools/pmc_traffic.py $out/${tag}_pmc_f $out/${tag}_pmc_w $out/${tag}_pmc_traffic_bf16.json "$cmd"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $out/${tag}_pmc_m -o m -- $cmd > /dev/null 2> $out/${tag}_pmc_m.err
echo "MFMA-busy pass done"
python3 tools/pmc_mfma.py $out/${tag}_pmc_m $out/${tag}_pmc_mfma_bf16.json "$cmd"
rm -rf $out/${tag}_pmc_f $out/${tag}_pmc_w $out/${tag}_pmc_m $out/${tag}_prof/p_kernel_trace.csv $out/${tag}_prof32/p_kernel_trace.csv
echo "all done"
