# usage: bash tools/tune_sweep.sh <nfk_set_tuning values...>   (one bench.py run per value)
for t in "$@"; do echo "tuning $t"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --tuning $t 2>&1 | tail -1 | cut -c1-140; done
