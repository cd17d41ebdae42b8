# usage (on the GPU box, from the repo root): bash tools/refresh_profiles.sh <tag>      e.g. r01_h
# Writes under gpurun_out/: the default bench line, the rocprofv3 kernel stats of the same command (+ its bench line), and
# the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) summarised by tools/pmc_traffic.py.
set -e
tag=${1:-r01_x}
out=gpurun_out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench_bf16.json 2> $out/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o p -- python3 bench.py --steps 20 --warmup 5 \
  > $out/${tag}_bench_bf16_under_rocprofv3.json 2> $out/${tag}_prof.err
cp $out/${tag}_prof/p_kernel_stats.csv $out/${tag}_kernel_stats_bf16_pipelined.csv
echo "kernel stats done"
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-overlap"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_f -o f -- $cmd > /dev/null 2> $out/${tag}_pmc_f.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_w -o w -- $cmd > /dev/null 2> $out/${tag}_pmc_w.err
echo "WRITE_SIZE pass done"
python3 tools/pmc_traffic.py $out/${tag}_pmc_f $out/${tag}_pmc_w $out/${tag}_pmc_traffic_bf16.json "$cmd"
rm -rf $out/${tag}_pmc_f $out/${tag}_pmc_w $out/${tag}_prof/p_kernel_trace.csv
echo "all done"
