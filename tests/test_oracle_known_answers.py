"""Pins both CPU oracles to the hand-derived known-answer cases.

The reference has no tests or fixtures for this path (parity unpinned), so
these cases — derived in tests/golden/make_known_answers.py from the algorithm statement — are
what the oracle is anchored on.
"""
import numpy as np
import pytest

from oracle import np_oracle, oracle
from tests.util import bits_to_xyzi, case_inputs, load_known_answers, same_bits, xyzi_of

CASES = load_known_answers()
STATUS = {"OK": oracle.OK, "EMPTY_INPUT": oracle.EMPTY_INPUT, "GRID_OVERFLOW": oracle.GRID_OVERFLOW}


def _expected_out(case):
    e = bits_to_xyzi(case["expect"]["out"])
    return np.stack([e["x"], e["y"], e["z"], e["intensity"]], axis=1) if len(e) else np.zeros((0, 4), np.float32)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("stable", [False, True])
def test_cpp_oracle_known_answers(case, stable):
    sensors, params = case_inputs(case)
    st, merged, out, rep = oracle.merge_voxelize(sensors, params, threads=1, stable=stable)
    exp = case["expect"]
    assert st == STATUS[exp["status"]]
    want = _expected_out(case)
    assert rep.n_out == len(want)
    assert same_bits(xyzi_of(out), want), (xyzi_of(out), want)
    if len(out):
        assert np.all(out["pad"] == 1.0)
    if "merged" in exp:
        m = bits_to_xyzi(exp["merged"])
        assert same_bits(xyzi_of(merged), np.stack([m["x"], m["y"], m["z"], m["intensity"]], axis=1))
    if "cells" in exp and st == oracle.OK:
        assert rep.cells.tolist() == exp["cells"]           # occupancy, in output order
    if "counts" in exp and st == oracle.OK:
        assert rep.counts.tolist() == exp["counts"]
    if "matrix" in exp:
        m = oracle.quat_to_matrix(sensors[0].q_xyzw, sensors[0].t_xyz)
        assert same_bits(m.reshape(-1), np.array(exp["matrix"], dtype=np.uint32).view(np.float32))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_numpy_oracle_known_answers(case):
    sensors, params = case_inputs(case)
    exp = case["expect"]
    n_total = sum(s.n for s in sensors)
    if n_total == 0:
        st, xyz, inten, cnt, cell = np_oracle.voxelgrid(np.zeros((0, 3), np.float32), np.zeros(0, np.float32),
                                                        params.leaf)
        assert st == np_oracle.EMPTY_INPUT
        return
    st, xyz, inten, cnt, cell = np_oracle.merge_voxelize(sensors, params, sequential=True)
    assert st == STATUS[exp["status"]]
    want = _expected_out(case)
    got = np.concatenate([xyz, inten[:, None]], axis=1)
    assert same_bits(got, want), (got, want)
    if "counts" in exp:
        assert cnt.tolist() == exp["counts"]
    if "cells" in exp:
        assert cell.tolist() == exp["cells"]
    if "matrix" in exp:
        m = np_oracle.quat_to_matrix(sensors[0].q_xyzw, sensors[0].t_xyz)
        assert same_bits(m.reshape(-1), np.array(exp["matrix"], dtype=np.uint32).view(np.float32))


def test_multithreaded_ingest_matches_serial():
    from cloud_merger_amd import synth
    sensors, params = synth.config2(n_per_sensor=20_000, min_pts=2)
    st1, m1, o1, _ = oracle.merge_voxelize(sensors, params, threads=1)
    st6, m6, o6, rep = oracle.merge_voxelize(sensors, params, threads=6)
    assert st1 == st6 == oracle.OK and rep.threads_used == 4
    assert m1.tobytes() == m6.tobytes() and o1.tobytes() == o6.tobytes()


def test_oracle_checksums_of_seeded_frames():
    """tests/golden/oracle_checksums.json (make_oracle_checksums.py): the oracle's output for seeded synthetic frames must
    not drift — neither through the oracle, nor the generators, nor the flags that decide fp32 results."""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_checksums", os.path.join(here, "make_oracle_checksums.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "oracle_checksums.json")))
    got = {name: mod.digest(*f) for name, f in mod.frames()}
    assert got == want
