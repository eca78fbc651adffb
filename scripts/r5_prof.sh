# kernel-trace stats of the train step (eager, two streams): which kernels the step spends its time in
# usage: bash scripts/r3_prof.sh TAG [bench args]      -> gpurun_out/TAG_kernel_stats.csv + a printed summary
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
TAG=${1:-r04}; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-extras --no-roofline "$@" > $R/gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?"
cd $R; python - <<PY
import csv, glob, shutil
f = glob.glob("gpurun_out/${TAG}_prof/*/*kernel_stats.csv")[0]
shutil.copy(f, "gpurun_out/${TAG}_kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
nst = max(1, sum(int(r["Calls"]) for r in rows if "clip_adam" in r["Name"]))
print("total kernel time %.1f ms over %d steps (warm-up, launch-mode probes, timed) -> %.2f ms/step; %d launches/step" % (tot / 1e6, nst, tot / 1e6 / nst, sum(int(r["Calls"]) for r in rows) / nst))
for r in rows[:45]:
    print("%-110s %6d %9.2f ms %8.1f us %5.1f%%" % (r["Name"][:110], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
