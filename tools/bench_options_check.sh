# every schedule / dtype option of bench.py runs and prints one JSON line (2 steps each)
set -e
for opts in "" "--no-pipeline" "--no-graph" "--no-graph --no-overlap" "--dtype f32" "--dtype f32 --no-pipeline" "--global-batch 32"; do
  echo "opts: $opts"
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $opts 2>&1 | tail -1 | cut -c1-160
done
