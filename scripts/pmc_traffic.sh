#!/bin/bash
# usage: [MP=0] [BENCH_ARGS="--config 3 --dense"] pmc_traffic.sh TAG
# HBM traffic of one frame from the PMC counters (separate rocprofv3 --pmc passes, as
# MI355X_MICROARCH.md §HBM prescribes). Writes gpurun_out/${TAG}_traffic.json. The stream runs without box misses
# (--jump-every 0): every frame but a context's first takes the steady-state path, whose kernels the per-frame figures are for
# (a kernel launched for fewer than half of the frames — the first frames' fixed-grid passes — is left out).
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_$C
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/${TAG}_$C -- \
      python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --profile-frames 2 --jump-every 0 --min-pts ${MP:-2} ${BENCH_ARGS:-} > gpurun_out/${TAG}_$C.json 2> gpurun_out/${TAG}_$C.err || { tail -5 gpurun_out/${TAG}_$C.err; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, glob, json, re, sys, collections
tag = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != c: continue
        m = re.search(r"(k[234]?_\w+)", r["Kernel_Name"])
        if not m: continue
        per[m.group(1)][0] += float(r["Counter_Value"]); per[m.group(1)][1] += 1
    res[c] = {k: {"sum_kb": v[0], "dispatches": v[1]} for k, v in per.items()}
bench = json.load(open(f"gpurun_out/{tag}_FETCH_SIZE.json"))
frames = max(res["FETCH_SIZE"].get(k, {"dispatches": 0})["dispatches"] for k in ("k_keys", "k2_local", "k3_local"))
out = {"frames": frames, "unit": "bytes per frame", "kernels": {},
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB -> bytes). gfx950 FETCH_SIZE "
               "counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md): 'fetch_x2' doubles it; narrower "
               "accesses are uncalibrated, so the true figure lies between raw and x2."}
tot_f = tot_w = 0.0
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    if k in ("k_probe_lds_order", "k_setup") or res["FETCH_SIZE"].get(k, {"dispatches": 0})["dispatches"] < frames // 2: continue   # one-off kernels (probe, bootstrap)
    # per frame that ran this kernel: the quantile passes' kernels (k4_*) run on fewer frames than the finish (a context's first
    # frame takes the fixed-grid passes), and a fixed-grid kernel may be launched two or three times per frame
    def per_frame(c):
        r_ = res[c].get(k, {"sum_kb": 0, "dispatches": 0})
        ref = res[c].get("k4_hist" if k.startswith("k4_") else ("k2_hist0" if (k.startswith("k2_") or k == "k_gscan") else ""), None)
        nf = ref["dispatches"] if ref and ref["dispatches"] >= frames // 2 else frames
        return r_["sum_kb"] * 1024 / max(1, nf)
    f = per_frame("FETCH_SIZE")
    w = per_frame("WRITE_SIZE")
    out["kernels"][k] = {"fetch_raw": f, "fetch_x2": 2 * f, "write": w}
    tot_f += f; tot_w += w
out["fetch_raw"] = tot_f; out["fetch_x2"] = 2 * tot_f; out["write"] = tot_w
out["traffic_low"] = tot_f + tot_w; out["traffic_high"] = 2 * tot_f + tot_w
out["algorithmic_bytes_per_frame"] = bench["roofline"]["algorithmic_bytes_per_frame"]
import subprocess
out["source_hash"] = subprocess.run([sys.executable, "bench.py", "--source-hash"], capture_output=True, text=True).stdout.strip()
out["path"] = bench["config"]["path"]
json.dump(out, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("frames", "fetch_raw", "fetch_x2", "write", "traffic_low", "traffic_high", "algorithmic_bytes_per_frame")}))
for k, v in out["kernels"].items(): print("%-14s fetch_raw %8.1f MB  write %8.1f MB" % (k, v["fetch_raw"] / 1e6, v["write"] / 1e6))
PY
