"""usage: pmc_account.py TAG — per-frame HBM traffic from the two rocprofv3 --pmc passes pmc_traffic.sh left under
gpurun_out/TAG_{FETCH,WRITE}_SIZE/ (can be rerun in the build container on the merged CSVs). Writes gpurun_out/TAG_traffic.json."""
import csv, glob, json, re, subprocess, sys, collections
tag = sys.argv[1]
# one entry per kernel VARIANT (template arguments kept): a frame's steady-state route and a context's first frames launch
# different instantiations of the same kernel, and k3_local has a second, larger shape launched beside the usual one
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != c: continue
        m = re.search(r"(k[234g]?_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        per[m.group(1)][0] += float(r["Counter_Value"]); per[m.group(1)][1] += 1
    res[c] = {k: {"sum_kb": v[0], "dispatches": v[1]} for k, v in per.items()}
bench = json.load(open(f"gpurun_out/{tag}_FETCH_SIZE.json"))
disp = lambda pre: sum(v["dispatches"] for k, v in res["FETCH_SIZE"].items() if k.startswith(pre))
# the route the figures are for is the one most frames took: the quantile passes (one k4_hist per frame), else the fixed-grid
# passes (one k2_hist0 per frame), else the general path (one k_keys per frame)
route, frames = max((("k4_hist", disp("k4_hist")), ("k2_hist0", disp("k2_hist0")), ("k_keys", disp("k_keys"))), key=lambda t: t[1])
out = {"frames": frames, "unit": "bytes per frame", "kernels": {},
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB -> bytes). gfx950 FETCH_SIZE "
               "counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md): 'fetch_x2' doubles it; narrower "
               "accesses are uncalibrated, so the true figure lies between raw and x2. Per kernel variant: mean per launch x "
               "launches per frame of the route most frames took (a variant launched on fewer than a quarter of them — the "
               "other route's, the probe, the bootstrap — is left out)."}
tot_f = tot_w = 0.0
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    n = res["FETCH_SIZE"].get(k, {"dispatches": 0})["dispatches"]
    if k.startswith(("k_probe_lds_order", "k_setup")) or n < max(1, frames // 4): continue
    # launches per frame: a whole number for a kernel every frame runs once or more (k2_scatter: 2-3 passes), the fraction of
    # frames for one launched on some of them (k3_local's large shape); a kernel both routes share (k3_compact) counts once
    lpf = n / frames if n < 0.9 * frames else max(1, round(n / frames)) if n > 1.4 * frames else 1.0
    def per_frame(c):
        r_ = res[c].get(k, {"sum_kb": 0, "dispatches": 0})
        return r_["sum_kb"] * 1024 / max(1, r_["dispatches"]) * lpf
    f, w = per_frame("FETCH_SIZE"), per_frame("WRITE_SIZE")
    out["kernels"][k] = {"launches": n, "launches_per_frame": round(lpf, 3), "fetch_raw": f, "fetch_x2": 2 * f, "write": w}
    tot_f += f; tot_w += w
out["fetch_raw"] = tot_f; out["fetch_x2"] = 2 * tot_f; out["write"] = tot_w
out["traffic_low"] = tot_f + tot_w; out["traffic_high"] = 2 * tot_f + tot_w
out["algorithmic_bytes_per_frame"] = bench["roofline"]["algorithmic_bytes_per_frame"]
out["source_hash"] = subprocess.run([sys.executable, "bench.py", "--source-hash"], capture_output=True, text=True).stdout.strip()
out["path"] = bench["config"]["path"]
json.dump(out, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("frames", "fetch_raw", "fetch_x2", "write", "traffic_low", "traffic_high", "algorithmic_bytes_per_frame")}))
for k, v in out["kernels"].items(): print("%-52s x%-5.2f fetch_raw %8.1f MB  write %8.1f MB" % (k[:52], v["launches_per_frame"], v["fetch_raw"] / 1e6, v["write"] / 1e6))
