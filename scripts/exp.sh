cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in $MODES; do
  for mp in 2; do
  CM_EXP=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp_${m}_${mp} -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --min-pts $mp > /dev/null 2> gpurun_out/exp.err
  f=$(find gpurun_out/exp_${m}_${mp} -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$m" "$mp" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in rows:
    m=re.search(r"(k_\w+(<\w+>)?)", r["Name"])
    if m: out.append("%s=%.1f"%(m.group(1), float(r["AverageNs"])/1e3))
print("mode",sys.argv[2],"min_pts",sys.argv[3]," ".join(sorted(out)))
PY
  rm -rf gpurun_out/exp_${m}_${mp}
  done
done
