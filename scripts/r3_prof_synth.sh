R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_synth_prof -- python $R/bench.py --mode synth --steps 10 --warmup 3 > $R/gpurun_out/r3_synth_prof.log 2>&1; echo "prof rc=$?"
cd $R; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_synth_prof/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over 2 runs" % (tot / 1e6))
for r in rows[:16]:
    print("%-90s %6d %9.2f ms %8.1f us %5.1f%%" % (r["Name"][:90], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
