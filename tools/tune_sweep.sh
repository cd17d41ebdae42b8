for t in 208 216 232; do echo "tuning $t"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --tuning $t 2>&1 | tail -1 | cut -c1-140; done
for t in 1256 1384 1512; do echo "tuning $t"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --tuning $t 2>&1 | tail -1 | cut -c1-140; done
