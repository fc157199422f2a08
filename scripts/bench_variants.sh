for v in "" "--jump-every 0" "--static" "--min-pts 0"; do
  timeout -k 10 200 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-e2e $v > gpurun_out/var.json 2> gpurun_out/var.err || { tail -5 gpurun_out/var.err; exit 1; }
  python3 - "$v" <<PY
import json,sys
d=json.load(open("gpurun_out/var.json"))
print("%-16s ms/step %.4f frac %.4f alone %.4f ms redone %s quant %s"%(sys.argv[1], d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["one_frame_alone"]["t_device_ms"],d["config"].get("redone_frames"),d["config"].get("quantile_frames")))
PY
done
