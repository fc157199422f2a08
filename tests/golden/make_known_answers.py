#!/usr/bin/env python3
"""Writes tests/golden/known_answers.json: the hand-derived known-answer cases of SURVEY.md
Appendix B.  The reference holds no fixtures for this path (parity unpinned), so every expectation
below is derived here from the algorithm statement with explicit scalar
float32 arithmetic — this script imports neither oracle nor the product.

Floats are stored as uint32 bit patterns so the fixture is exact.
Run:  python tests/golden/make_known_answers.py
"""
import json
import os

import numpy as np

F = np.float32


def bits(v):
    return int(np.asarray(v, dtype=F).view(np.uint32))


def pts(rows):
    """rows of (x,y,z,i) -> list of 4 bit patterns"""
    return [[bits(c) for c in r] for r in rows]


def mean_seq(vals):
    """fp32 running sum in the given order, then one fp32 division by the count."""
    acc = F(0)
    for v in vals:
        acc = F(acc + F(v))
    return F(acc / F(len(vals)))


IDENT = dict(q=[0.0, 0.0, 0.0, 1.0], t=[0.0, 0.0, 0.0])
cases = []

# 1. Identity, one voxel.
a, b = (0.01, 0.01, 0.01, 10.0), (0.09, 0.09, 0.09, 30.0)
c = [mean_seq([a[k], b[k]]) for k in range(4)]
cases.append(dict(name="identity_one_voxel", leaf=0.1, min_pts=0, crop=None,
                  sensors=[dict(points=pts([a, b]), **IDENT)],
                  expect=dict(status="OK", out=pts([c]), cells=[[0, 0, 0]], counts=[2])))

# 2. Face inclusion: fl32(x * fl32(1/0.1f)).
inv = F(1) / F(0.1)
assert inv == F(10.0)
xs = [F(0.1), F(0.3)]
cells2 = [int(np.floor(F(x * inv))) for x in xs]
assert cells2 == [1, 3], cells2          # 0.3f*10 is an exact tie, rounds to even = 3.0f
cases.append(dict(name="face_inclusion", leaf=0.1, min_pts=0, crop=None,
                  sensors=[dict(points=pts([(x, 0.05, 0.05, 1.0) for x in xs]), **IDENT)],
                  expect=dict(status="OK", cells=[[1, 0, 0], [3, 0, 0]], counts=[1, 1],
                              out=pts([(x, 0.05, 0.05, 1.0) for x in xs]))))

# 3. Negative coordinates: floor, not truncation.
xn = [F(-0.01), F(-0.1), np.nextafter(F(-0.1), F(-1))]
cells3 = [int(np.floor(F(x * inv))) for x in xn]
assert cells3 == [-1, -1, -2], cells3
cases.append(dict(name="negative_floor", leaf=0.1, min_pts=0, crop=None,
                  sensors=[dict(points=pts([(x, 0.05, 0.05, 0.0) for x in xn]), **IDENT)],
                  expect=dict(status="OK",
                              cells=[[-2, 0, 0], [-1, 0, 0]], counts=[1, 2],
                              out=pts([(xn[2], 0.05, 0.05, 0.0),
                                       (mean_seq([xn[0], xn[1]]), mean_seq([0.05, 0.05]),
                                        mean_seq([0.05, 0.05]), 0.0)]))))

# 4. min_points_per_voxel: voxels holding 1, 2, 3 points -> 3,3,2,1 voxels for 0,1,2,3.
p4 = [(0.05, 0.05, 0.05, 1.0),
      (0.15, 0.05, 0.05, 2.0), (0.16, 0.05, 0.05, 4.0),
      (0.25, 0.05, 0.05, 3.0), (0.26, 0.05, 0.05, 6.0), (0.27, 0.05, 0.05, 9.0)]
for mp, nvox in [(0, 3), (1, 3), (2, 2), (3, 1)]:
    groups = [[0], [1, 2], [3, 4, 5]]
    kept = [g for g in groups if len(g) >= mp]
    out = [[mean_seq([p4[i][k] for i in g]) for k in range(4)] for g in kept]
    assert len(out) == nvox
    cases.append(dict(name=f"min_points_{mp}", leaf=0.1, min_pts=mp, crop=None,
                      sensors=[dict(points=pts(p4), **IDENT)],
                      expect=dict(status="OK", out=pts(out), counts=[len(g) for g in kept],
                                  cells=[[len(g) - 1, 0, 0] for g in kept])))

# 5. Output order = ascending linear index: x fastest, then y, then z.
p5 = [(0.15, 0.05, 0.05, 1.0), (0.05, 0.15, 0.05, 2.0), (0.05, 0.05, 0.15, 3.0), (0.05, 0.05, 0.05, 4.0)]
cases.append(dict(name="ordering", leaf=0.1, min_pts=0, crop=None,
                  sensors=[dict(points=pts(p5), **IDENT)],
                  expect=dict(status="OK", out=pts([p5[3], p5[0], p5[1], p5[2]]), counts=[1, 1, 1, 1],
                              cells=[[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])))

# 6. Crop inclusivity with the reference ROI (Parameter.h:31-35): closed interval, NaN dropped.
z6 = [F(-0.5), F(3.0), np.nextafter(F(3.0), F(10)), F(np.nan)]
p6 = [(1.0 + k, 0.0, z6[k], float(k)) for k in range(4)]
cases.append(dict(name="crop_inclusive", leaf=0.1, min_pts=0,
                  crop=dict(min=[-15.0, -5.0, -0.5], max=[60.0, 5.0, 3.0]),
                  sensors=[dict(points=pts(p6), is_dense=False, **IDENT)],
                  expect=dict(status="OK", merged=pts(p6[:2]), counts=[1, 1],
                              out=pts([p6[0], p6[1]]),
                              cells=[[10, 0, -5], [20, 0, 30]])))

# 7. Transform rounding: 90 degree yaw given as doubles.
s = float(np.sqrt(0.5))
q7 = [0.0, 0.0, s, s]
x, y, z, w = F(q7[0]), F(q7[1]), F(q7[2]), F(q7[3])
tx, ty, tz = F(2) * x, F(2) * y, F(2) * z
twx, twy, twz = F(tx * w), F(ty * w), F(tz * w)
txx, txy, txz = F(tx * x), F(ty * x), F(tz * x)
tyy, tyz, tzz = F(ty * y), F(tz * y), F(tz * z)
m7 = [F(1) - F(tyy + tzz), F(txy - twz), F(txz + twy), F(0.5),
      F(txy + twz), F(1) - F(txx + tzz), F(tyz - twx), F(-0.25),
      F(txz - twy), F(tyz + twx), F(1) - F(txx + tyy), F(0.125)]
pin = (F(1), F(2), F(3))
img = [F(F(F(m7[4 * r] * pin[0]) + F(m7[4 * r + 1] * pin[1])) + F(m7[4 * r + 2] * pin[2])) + m7[4 * r + 3]
       for r in range(3)]
cases.append(dict(name="transform_rounding", leaf=0.1, min_pts=0, crop=None,
                  sensors=[dict(points=pts([(1.0, 2.0, 3.0, 7.0)]), q=q7, t=[0.5, -0.25, 0.125])],
                  expect=dict(status="OK", matrix=[bits(v) for v in m7],
                              merged=pts([(img[0], img[1], img[2], 7.0)]),
                              out=pts([(img[0], img[1], img[2], 7.0)]), counts=[1])))

# 8. Overflow guard: 40001^3 cells > INT32_MAX -> output = input.
p8 = [(-1000.0, -1000.0, -1000.0, 1.0), (1000.0, 1000.0, 1000.0, 2.0)]
cases.append(dict(name="overflow_guard", leaf=0.05, min_pts=0, crop=None,
                  sensors=[dict(points=pts(p8), **IDENT)],
                  expect=dict(status="GRID_OVERFLOW", out=pts(p8))))

# 9. Concatenation order: sensor 0's points precede sensor 1's in the merged cloud.
p9a = [(0.05, 0.05, 0.05, 1.0), (0.55, 0.05, 0.05, 2.0)]
p9b = [(0.35, 0.05, 0.05, 3.0)]
cases.append(dict(name="concat_order", leaf=0.1, min_pts=0, crop=None,
                  sensors=[dict(points=pts(p9a), **IDENT), dict(points=pts(p9b), **IDENT)],
                  expect=dict(status="OK", merged=pts(p9a + p9b), counts=[1, 1, 1],
                              out=pts([p9a[0], p9b[0], p9a[1]]),
                              cells=[[0, 0, 0], [3, 0, 0], [5, 0, 0]])))

# 10. Empty input: width = height = 0.
cases.append(dict(name="empty_input", leaf=0.1, min_pts=0, crop=None,
                  sensors=[dict(points=[], **IDENT)],
                  expect=dict(status="EMPTY_INPUT", out=[])))

# 11. downsample_all_data = false: xyz only, intensity left at 0 (A.4 step 8).
cases.append(dict(name="xyz_only_centroid", leaf=0.1, min_pts=0, crop=None, downsample_all=False,
                  sensors=[dict(points=pts([a, b]), **IDENT)],
                  expect=dict(status="OK", out=pts([(c[0], c[1], c[2], 0.0)]), counts=[2])))

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "known_answers.json"), "w") as f:
    json.dump(dict(note="SURVEY.md Appendix B known-answer cases; floats as uint32 bit patterns "
                        "[x,y,z,intensity]; derived by tests/golden/make_known_answers.py",
                   cases=cases), f, indent=1)
print("wrote", len(cases), "cases")
