"""Writes tests/golden/oracle_checksums.json: SHA-256 of what the oracle returns for seeded synthetic frames (merged cloud,
kept cells, counts, centroids with stable tie order). Not golden vectors of the reference (it has none: parity unpinned) —
a tripwire against silent changes of the oracle, of the synthetic generators or of the build flags that decide fp32
results (-ffp-contract=off). Run from the repository root: python tests/golden/make_oracle_checksums.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cloud_merger_amd import synth
from oracle import oracle


def frames():
    yield "cfg1_2x100k_xyz_10cm", synth.config1()
    yield "cfg2_4x50k_5cm_min2", synth.config2(n_per_sensor=50_000, min_pts=2)
    yield "cfg3_6x60k_roi_5cm", synth.config3(n_per_sensor=60_000, n_sensors=6, min_pts=0, leaf=0.05)
    s, p = synth.config2(n_per_sensor=40_000, min_pts=0)
    p.crop_min, p.crop_max = (-30.0, -25.0, -3.0), (35.0, 30.0, 4.0)
    p.outlier_radius, p.outlier_min_neighbors = 0.4, 2
    yield "cfg2_4x40k_crop_outlier_r0.4", (s, p)


def digest(sensors, params):
    st, merged, out, rep = oracle.merge_voxelize(sensors, params, threads=1, stable=True)
    h = hashlib.sha256()
    def xyzi(p):                                           # (the 32-byte records' padding is not part of the result)
        return np.ascontiguousarray(np.stack([p["x"], p["y"], p["z"], p["intensity"]], axis=1).astype("<f4"))
    for a in (xyzi(merged), xyzi(out), np.ascontiguousarray(rep.cells, dtype="<i4"), np.ascontiguousarray(rep.counts, dtype="<u4")):
        h.update(a.tobytes())
    return {"status": int(st), "n_merged": int(rep.n_merged), "n_out": int(rep.n_out), "sha256": h.hexdigest()}


if __name__ == "__main__":
    res = {name: digest(*f) for name, f in frames()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_checksums.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))
