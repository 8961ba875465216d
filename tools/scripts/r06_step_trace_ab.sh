cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for x in 1 0; do
  O=$R/gpurun_out/r06kt$x; rm -rf $O; mkdir -p $O
  cd $R
  TMF_WINO_X=$x TMF_BENCH_SETUP_STEPS=3 timeout 400 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-also --no-cpu-baseline > $O/log.txt 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== TMF_WINO_X=$x"; python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
print("total kernel ms", tot/1e6)
PY
  grep -o '"value": [0-9.]*' $O/log.txt | head -1
done
