"""Synthesises the recorded sequence BASELINE config 4 replays (the reference ships no .pcd/.bag
data: my_cloud_fusion/data holds only an RViz layout) and reads/writes PCD v0.7 from Python.

    python -m cloud_merger_amd.replay_data OUT_DIR --frames 100 --sensors 4
"""
import argparse
import os

import numpy as np

from . import synth


def write_pcd(path, xyz, intensity=None):
    xyz = np.asarray(xyz, dtype="<f4").reshape(-1, 3)
    cols = [xyz] if intensity is None else [xyz, np.asarray(intensity, dtype="<f4").reshape(-1, 1)]
    rec = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = "x y z" if intensity is None else "x y z intensity"
    k = rec.shape[1]
    hdr = (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS {names}\nSIZE {' '.join(['4'] * k)}\n"
           f"TYPE {' '.join(['F'] * k)}\nCOUNT {' '.join(['1'] * k)}\nWIDTH {len(rec)}\nHEIGHT 1\n"
           f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(rec)}\nDATA binary\n")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(rec.tobytes())


def read_pcd(path):
    """Returns (n, k) float32 and the field names (binary FLOAT32 files only)."""
    with open(path, "rb") as f:
        names, n = [], 0
        while True:
            line = f.readline().decode().strip()
            if line.startswith("FIELDS"):
                names = line.split()[1:]
            elif line.startswith("POINTS"):
                n = int(line.split()[1])
            elif line.startswith("DATA"):
                assert line.split()[1] == "binary", line
                break
        data = np.frombuffer(f.read(n * 4 * len(names)), dtype="<f4").reshape(n, len(names))
    return data, names


def sensor_poses(n_sensors, seed=4000):
    """Vehicle-mounted layout: sensors on a ring, yaw outwards, small tilt. (q_xyzw, t_xyz) each."""
    rng = np.random.Generator(np.random.PCG64(seed))
    poses = []
    for s in range(n_sensors):
        yaw = 2 * np.pi * s / n_sensors + rng.uniform(-0.05, 0.05)
        pitch = rng.uniform(-0.02, 0.02)
        cy, sy, cp, sp = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2)
        q = np.array([-sy * sp, cy * sp, sy * cp, cy * cp])          # yaw * pitch
        t = np.array([1.2 * np.cos(yaw), 1.2 * np.sin(yaw), 1.8])
        poses.append((q / np.linalg.norm(q), t))
    return poses


def write_sequence(out_dir, frames=100, sensors=4, rings=32, azimuths=3750):
    os.makedirs(out_dir, exist_ok=True)
    poses = sensor_poses(sensors)
    with open(os.path.join(out_dir, "transforms.txt"), "w") as f:
        for q, t in poses:
            f.write(" ".join(repr(float(v)) for v in list(q) + list(t)) + "\n")
    for fr in range(frames):
        for s in range(sensors):
            xyz, inten = synth.velodyne_frame(fr, s, rings=rings, azimuths=azimuths)
            write_pcd(os.path.join(out_dir, f"frame_{fr:04d}_sensor_{s}.pcd"), xyz, inten)
    return poses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--sensors", type=int, default=4)
    a = ap.parse_args()
    write_sequence(a.out_dir, a.frames, a.sensors)
    print("wrote", a.frames * a.sensors, "clouds to", a.out_dir)
