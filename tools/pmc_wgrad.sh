# per-dispatch fabric reads and L2 hit rate of level_wgrad_kernel (the level's batched weight-gradient GEMMs): bash tools/pmc_wgrad.sh
set -e
out=gpurun_out
export TMPDIR=/tmp
cmd="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-precision --no-graph --no-overlap"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/wg_f -o f -- $cmd > /dev/null 2> $out/wg_f.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/wg_h -o h -- $cmd > /dev/null 2> $out/wg_h.err
python3 - <<'PY'
import csv, glob, collections
def rows(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    return list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("gpurun_out/wg_f", "gpurun_out/wg_h"):
    for r in rows(d):
        if "level_wgrad_kernel" in r["Kernel_Name"]:
            agg[int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for wg, c in sorted(agg.items()):
    print("workgroups", wg, {k: round(sum(v) / len(v), 1) for k, v in c.items()}, "n", {k: len(v) for k, v in c.items()})
PY
