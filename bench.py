#!/usr/bin/env python3
"""Headline benchmark: merged+voxelised points/s of the merge -> voxel-grid hot path.

A step = one frame: every sensor cloud (already resident in HBM) handed to the library, one
cm_merge_voxelize (transform + crop + concatenate + VoxelGrid), result count read back.
Workload = BASELINE.json configs[1]: 4 x 1 M XYZI points, random SE(3) per sensor, 5 cm voxels.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: frames are independent units, so each rank (one process per GPU) runs its own frame
stream with no data-path collective ("weak" scaling); torch.distributed (RCCL) only provides the
barriers and the max-over-ranks reduction of the timing.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "merged+voxelized points/sec at 5 cm leaf, 4\u00d71 M-pt inputs; HBM GB/s fraction"   # BASELINE.json's metric, verbatim
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0          # measured float4-copy ceiling, same guide (SURVEY.md §8d asks for both)


def kernel_source_hash():
    """sha256 over the library's sources (csrc/ + include/cloudmerge.h): the PMC traffic files under profiles/ carry the hash of
    the build they were measured on, and a file from another build is not quoted as this one's traffic."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "cloud_merger_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".cpp", ".h", ".hpp")):
            h.update(name.encode()); h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "cloudmerge.h"), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--stream-frames", type=int, default=6, help="distinct frames of the moving stream (6 x 64 MB of input > 256 MiB)")
    ap.add_argument("--jump-every", type=int, default=64, help="every this many frames one reaches 30 %% further out (0: never)")
    ap.add_argument("--static", action="store_true", help="the round-1 loop: one frame resubmitted every step")
    ap.add_argument("--moving", action="store_true", help="--config 3 --dense: a stream of --stream-frames fresh draws of the scene instead of one frame resubmitted")
    ap.add_argument("--source-hash", action="store_true", help="print the hash of the kernel sources and exit (scripts/pmc_traffic.sh)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer (PCIe-inclusive) runs")
    ap.add_argument("--dense", action="store_true", help="config 3: points drawn inside the ROI (the sort-stress variant); config 5: "
                    "points drawn inside the crop box on shared structure (the ranks' tables hold the same voxels: real merges)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5],
                    help="2: headline (BASELINE.json configs[1]); 3: configs[2]; 5: configs[4], the single fused cloud — the "
                         "16 sensors dealt to the ranks, partial tables all-gathered (RCCL), merged on every rank")
    ap.add_argument("--points-per-sensor", type=int, default=0, help="points per sensor (default: the configuration's own — 1 M for "
                    "config 2, 4 M for config 5); anything else is a rehearsal, not BASELINE.json's workload (said so in config.workload)")
    ap.add_argument("--check", action="store_true", help="config 5 only: compare the fused cloud with the CPU oracle on rank 0 "
                                                          "(builds all 16 sensors there: reduced sizes only)")
    ap.add_argument("--min-pts", type=int, default=2, help="min points per voxel (reference: 2, PCL default: 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --backend gloo; not a measurement)")
    ap.add_argument("--profile-frames", type=int, default=50, help="frames timed alone with HIP events (median t_device)")
    ap.add_argument("--outlier-radius", type=float, default=0.0,
                    help="also run pcl::RadiusOutlierRemoval (min 1 neighbour) on the fused cloud before VoxelGrid "
                         "(SURVEY 8f rank 2; off for the headline metric)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="frames in flight per GPU (independent frames on separate HIP streams/contexts)")
    return ap.parse_args()


def main_fused(args):
    """BASELINE.json configs[4]: 16 sensors x 4 M points, 1 cm voxel, crop x[-15,45] y[-5,5] z[-0.5,3]; ONE fused cloud.
    Rank r (one process per GPU) holds the sensors s = r mod world. A step = one frame: cm_merge_partial on the rank's
    sensors (bucket path -> per-voxel sums, threshold deferred), all-gather of the tables (lengths first) over
    torch.distributed — "nccl" = RCCL over xGMI; "gloo" + --single-device rehearses the same code on one GPU with the
    exchange on the host — and cm_merge_tables on every rank (every rank ends up with the whole fused cloud, as an
    all-gather implies). Reference: fusePointclouds + voxelgrid, pc_preprocessing_main.cpp:131-177."""
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host_x = args.backend == "gloo"
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from cloud_merger_amd import capi, fused, synth
    n_sensors = 16
    nps = args.points_per_sensor or 4_000_000
    gen5 = synth.config5_dense_shard if args.dense else synth.config5_shard
    sensors, params = gen5(rank, world, n_per_sensor=nps, n_sensors=n_sensors, min_pts=args.min_pts)
    n_rank = sum(s.n for s in sensors)
    n_total = n_sensors * nps
    dev_clouds = [torch.from_numpy(np.ascontiguousarray(s.data).view(np.uint8).reshape(-1)).to(dev) for s in sensors]
    torch.cuda.synchronize()
    # (cm_merge_tables sorts ALL ranks' entries: at most one per surviving point of the whole frame — the dense variant's tables
    # come close to that, the synthesised cfg5's hold a few hundred thousand entries)
    cm = capi.CloudMerger(max_points_total=max(n_total if args.dense else n_rank, 1 << 16), max_sensors=max(1, len(sensors)), device=local_rank,
                          flags=capi.FLAG_OCCUPANCY if args.check else 0)
    cm.set_stream(torch.cuda.current_stream().cuda_stream)
    for k, s in enumerate(sensors):
        cm.set_transform(k, s.q_xyzw, s.t_xyz)

    acc = {"partial_ms": 0.0, "exchange_ms": 0.0, "merge_ms": 0.0}
    last = {}

    def step():
        for k, s in enumerate(sensors):
            cm.submit_device(k, dev_clouds[k].data_ptr(), s.n, s.point_step, s.off_x, s.off_y, s.off_z, s.off_i)
        t = {}
        res = fused.fused_cloud(cm, params, dist if world > 1 else None, rank, world, dev, host_exchange=host_x, times=t)
        for k in acc:
            acc[k] += t[k]
        last.update(t)
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        res = step()
    for k in acc:
        acc[k] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if res.status != capi.OK:
        raise SystemExit(f"fused cloud status {capi.status_string(res.status)}")
    if world > 1:
        tmax = torch.tensor([elapsed] + [acc[k] for k in ("partial_ms", "exchange_ms", "merge_ms")], dtype=torch.float64,
                            device="cpu" if host_x else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        worst = {k: float(tmax[1 + i].item()) / args.steps for i, k in enumerate(("partial_ms", "exchange_ms", "merge_ms"))}
    else:
        worst = {k: acc[k] / args.steps for k in acc}
    n_out = int(res.n_out)
    ms = 1e3 * elapsed / args.steps
    # per GPU: the rank reads its own points once and writes the whole fused cloud once (SURVEY.md 8d's accounting
    # applied to what one GPU does); the exchange is reported beside it, in ms and in table bytes
    b_alg_gpu = 16.0 * n_rank + 16.0 * n_out
    out = {
        "metric": "fused-cloud merged+voxelized points/sec at 1 cm leaf, 16x4 M-pt inputs; HBM GB/s fraction per GPU",
        "value": args.steps * n_total / elapsed, "unit": "points/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfg5{' (dense variant: points drawn inside the crop box, 70 % on a road surface shared by all sensors)' if args.dense else ''}: "
                               f"{n_sensors} x {nps} XYZI float32 points, yaw-only SE(3), 1 cm voxel, crop x[-15,45] y[-5,5] "
                               "z[-0.5,3]; one fused cloud on every rank",
                   "points_per_frame": n_total, "points_per_rank": n_rank, "voxels_out": n_out,
                   "min_points_per_voxel": args.min_pts, "sharding": f"sensor s on rank s mod {world}",
                   "collective": ("all-gather of partial voxel tables (lengths, then padded tables) over torch.distributed "
                                  + ("nccl = RCCL" if args.backend == "nccl" else "gloo on host copies (rehearsal)")) if world > 1
                                 else "none (one rank)",
                   "step_ms_worst_rank": worst,
                   "table_entries_this_rank": int(last.get("table_entries", 0)),
                   "table_bytes_this_rank": 32 * int(last.get("table_entries", 0)),
                   "points_kept_this_rank": int(getattr(res, "n_merged", 0)),
                   "gathered_entries": int(last.get("gathered_entries", 0)),
                   "gathered_bytes": 32 * int(last.get("gathered_entries", 0)),
                   "rehearsal_single_device": bool(args.single_device)},
        "roofline": {"bound": "hbm", "achieved": b_alg_gpu / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": b_alg_gpu / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "scope": "per GPU: (16 B x this rank's input points + 16 B x fused voxels) / step time"},
    }
    if args.check and rank == 0:
        from oracle import oracle
        allsens = gen5(0, 1, n_per_sensor=nps, n_sensors=n_sensors, min_pts=args.min_pts)[0]   # all 16, in sensor order
        st, _, ref, rep = oracle.merge_voxelize(allsens, params, threads=6, stable=True, want_merged=False)
        cells, counts = cm.cells(n_out)
        got = cm.result(n_out)
        ok = st == oracle.OK and rep.n_out == n_out and np.array_equal(rep.cells, cells) and np.array_equal(rep.counts, counts)
        dx = float(max(np.abs(got[a].astype(np.float64) - ref[a].astype(np.float64)).max() for a in ("x", "y", "z"))) if ok and n_out else None
        out["parity"] = {"occupancy_bit_exact": bool(ok), "max_abs_dxyz_m": dx}
        if not ok or (dx is not None and dx > 1e-4):
            print(json.dumps(out))
            raise SystemExit("fused cloud differs from the oracle")
    if rank == 0:
        print(json.dumps(out))
    cm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    if "--source-hash" in sys.argv:
        print(kernel_source_hash())
        return
    args = parse()
    if args.config == 5:
        return main_fused(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from cloud_merger_amd import capi, synth

    # ---- this rank's sensor stream: K distinct frames (more bytes than the 256 MiB Infinity Cache holds, so that no
    # step finds its input there), every frame a fresh draw with slightly different poses and bounds; now and then a
    # frame that reaches 30 % further out than its predecessors (it leaves the predicted box: CM_PATH_REDONE).
    # --static: the round-1 loop (one frame resubmitted for ever), kept for comparison.
    K = 1 if args.static else max(1, args.stream_frames)
    moving = not args.static and args.config == 2
    frames = []
    if args.config == 2:
        nps2 = args.points_per_sensor or 1_000_000
        workload = "cfg2: 4 x 1M XYZI float32 points, random SE(3) per sensor, 5 cm voxel, no crop"
        if nps2 != 1_000_000:
            workload = f"REHEARSAL (not BASELINE.json's size): cfg2's scene with 4 x {nps2} points, random SE(3) per sensor, 5 cm voxel, no crop"
        for k in range(K):
            frames.append(synth.config2_stream(k + 97 * rank, n_per_sensor=nps2, min_pts=args.min_pts)[0] if moving else
                          synth.config2(n_per_sensor=nps2, min_pts=args.min_pts)[0])
        params = synth.config2(n_per_sensor=8, min_pts=args.min_pts)[1]
        wide = (synth.config2_stream(1000 + rank, n_per_sensor=nps2, min_pts=args.min_pts, wide=True)[0]
                if (moving and args.jump_every) else None)
    else:
        gen = synth.config3_dense if args.dense else synth.config3
        workload = ("cfg3 (dense variant): 8 x 2M XYZI float32 points drawn inside the reference ROI (86 % survive the crop), "
                    "yaw-only SE(3), 2 cm voxel" if args.dense else
                    "cfg3: 8 x 2M XYZI float32 points, yaw-only SE(3), 2 cm voxel, reference ROI crop")
        sensors0, params = gen(min_pts=args.min_pts)
        frames, wide, moving, K = [sensors0], None, False, 1
        if args.dense and args.moving:
            K = max(2, args.stream_frames)
            frames += [synth.config3_dense(min_pts=args.min_pts, draw=k + 97 * rank)[0] for k in range(1, K)]
            moving = True
    if args.outlier_radius > 0:
        params.outlier_radius, params.outlier_min_neighbors = args.outlier_radius, 1
        workload += f" + radius outlier removal r={args.outlier_radius} m, min 1 neighbour"
    sensors = frames[0]
    n_in = sum(s.n for s in sensors)
    n_sens = len(sensors)

    def to_dev(fr):
        return [torch.from_numpy(np.ascontiguousarray(s.data).view(np.uint8).reshape(-1)).to(dev) for s in fr]
    dev_frames = [to_dev(fr) for fr in frames]            # inputs resident in HBM before the timed region
    dev_wide = to_dev(wide) if wide is not None else None
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream()
    cparams = capi.make_params(params)
    # Frames are independent units (the reference node is stateless per frame): keep `inflight` of
    # them going on separate HIP streams so one frame's latency-bound phases overlap another's.
    inflight = max(1, args.inflight)
    streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(inflight - 1)]
    cms = []
    for q in range(inflight):
        c = capi.CloudMerger(max_points_total=n_in, max_sensors=n_sens, device=local_rank)
        c.set_stream(streams[q].cuda_stream)
        cms.append(c)

    def jumps(i):
        return dev_wide is not None and i % args.jump_every == args.jump_every - 1

    def pick(i):
        """frame of step i: (host descriptors, device tensors)"""
        if jumps(i):
            return wide, dev_wide
        return frames[i % K], dev_frames[i % K]

    def enqueue(c, i):
        fr, dv = pick(i)
        for k, s in enumerate(fr):
            c.set_transform(k, s.q_xyzw, s.t_xyz)
            c.submit_device(k, dv[k].data_ptr(), s.n, s.point_step, s.off_x, s.off_y, s.off_z, s.off_i)
        c.merge_voxelize_async(cparams)

    stats = {"redone": 0, "packed": 0, "quantile": 0, "jumped": 0, "lat": [], "done_t": []}

    def run_steps(n, first=0, record=False):
        """n complete frames; at most `inflight` enqueued at any time; every frame's result is waited for."""
        res, issued, done, t_enq = None, 0, 0, {}
        while done < n:
            while issued < n and issued - done < inflight:
                t_enq[issued] = time.perf_counter()
                enqueue(cms[issued % inflight], first + issued)
                issued += 1
            res = cms[done % inflight].wait()
            if res.status != capi.OK:
                raise SystemExit(f"frame status {capi.status_string(res.status)}")
            if record:
                stats["jumped"] += 1 if jumps(first + done) else 0
                now = time.perf_counter()
                stats["lat"].append(now - t_enq[done]); stats["done_t"].append(now)
                stats["redone"] += 1 if res.path_flags & capi.PATH_REDONE else 0
                stats["packed"] += 1 if res.path_flags & capi.PATH_PACKED else 0
                stats["quantile"] += 1 if res.path_flags & capi.PATH_QUANTILE else 0
            done += 1
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, not measurement: the library allocates its work buffers on a context's first frame and, without a
    # crop box, takes its first box from one min/max pass — run one frame per context so that --warmup 0 still
    # times steady-state frames only.
    for q, c in enumerate(cms):
        enqueue(c, q)
        if c.wait().status != capi.OK:
            raise SystemExit("set-up frame failed")
    if args.warmup:
        run_steps(args.warmup, first=inflight)
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps, first=inflight + args.warmup, record=True)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    n_out = int(res.n_out)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    lat = np.sort(np.asarray(stats["lat"])) if stats["lat"] else np.zeros(1)
    gaps = np.sort(np.diff(np.asarray(stats["done_t"]))) if len(stats["done_t"]) > 2 else np.zeros(1)

    out = {
        "metric": METRIC,
        "value": world * args.steps * n_in / elapsed,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "points_per_frame": n_in, "voxels_out": n_out,
                   "min_points_per_voxel": args.min_pts, "sharding": f"frame-sharded x{world}, no collective",
                   "inputs": "resident in HBM (16-byte XYZI records)",
                   "stream": (f"moving: {K} fresh draws of the scene per rank in turn ({K * n_in * 16 >> 20} MiB of input); a context "
                              f"sees every {inflight}th frame" if (moving and args.config != 2) else
                              f"moving: {K} distinct frames per rank ({K * n_in * 16 >> 20} MiB of input, more than the 256 MiB "
                              f"Infinity Cache), poses and bounds jittered from frame to frame; "
                              + (f"{stats['jumped']} of the {args.steps} timed frames reached 30 % further out than the box predicted from "
                                 f"their predecessors (every {args.jump_every}th frame of the stream does)" if stats["jumped"] else
                                 f"none of the {args.steps} timed frames left its predicted box (every {args.jump_every}th frame of the stream "
                                 "does: the timed window is shorter than that; what such a frame costs is config.box_miss_frame_ms)"
                                 if (dev_wide is not None) else "no frame leaves its predicted box (--jump-every 0)")) if moving else
                             "static: the same frame resubmitted every step",
                   "frames_in_flight": inflight,
                   "redone_frames": stats["redone"], "packed_frames": stats["packed"], "quantile_frames": stats["quantile"],
                   "frame_latency_ms": {"p50": 1e3 * float(lat[len(lat) // 2]), "p99": 1e3 * float(lat[min(len(lat) - 1, int(0.99 * len(lat)))]),
                                        "note": "enqueue -> result count back on the host, with the other frames in flight"},
                   "frame_interval_ms": {"p50": 1e3 * float(gaps[len(gaps) // 2]), "p99": 1e3 * float(gaps[min(len(gaps) - 1, int(0.99 * len(gaps)))])},
                   "radix_ranking": "lds-add (device probe passed)" if res.path_flags & 1 else "ballot-match",
                   "path": ("quantile passes (cm_kernels_v4.hip + cm_kernels_v3.hip): ONE global pass over point records into buckets cut "
                            "at the previous frame's quantiles, one LDS-local finish workgroup per bucket (k3_local + k3_compact); "
                            "box %s" % ("predicted from the previous frame's bounds" if res.path_flags & 4 else "= crop box"))
                           if res.path_flags & capi.PATH_QUANTILE else
                           ("bucket path (cm_kernels_v2.hip + cm_kernels_v3.hip): %d global passes over point records, LDS-local finish%s; box %s"
                            % (res.sort_passes, " (k3_local + k3_compact)" if res.path_flags & capi.PATH_SPLIT else " (k2_local)",
                               "predicted from the previous frame's bounds" if res.path_flags & 4 else "= crop box"))
                           if res.path_flags & 2 else
                           "general path (cm_kernels.hip): %d-pass LSD sort of (voxel, point) pairs + gather" % res.sort_passes},
    }

    if rank == 0:
        # One frame alone on the GPU, nothing else in flight: host-observed latency and, by a pair of events on the
        # stream, the device time from the first kernel's start to the last one's end (no per-kernel events here).
        c0 = cms[0]
        s1 = streams[1] if inflight > 1 else torch.cuda.Stream(device=dev)     # (an explicit stream: NULL means "the context's own")
        c0.set_stream(s1.cuda_stream)
        lat1, dev1 = [], []
        for it in range(60):                                # (median of >= 50 iterations after 5 warm-ups)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fr, dv = frames[it % K], dev_frames[it % K]
            for k, s in enumerate(fr):
                c0.set_transform(k, s.q_xyzw, s.t_xyz)
                c0.submit_device(k, dv[k].data_ptr(), s.n, s.point_step, s.off_x, s.off_y, s.off_z, s.off_i)
            ta = time.perf_counter()
            e0.record(s1)
            c0.merge_voxelize_async(cparams)
            e1.record(s1)
            r1 = c0.wait()
            tb = time.perf_counter()
            torch.cuda.synchronize()
            if it >= 5 and not (r1.path_flags & capi.PATH_REDONE):
                lat1.append(tb - ta); dev1.append(e0.elapsed_time(e1))
        out["config"]["latency_one_frame_ms"] = {"host_enqueue_to_result": 1e3 * float(np.median(lat1)),
                                                 "device_first_kernel_to_last": float(np.median(dev1))}
        if dev_wide is not None:
            # What a frame that leaves the predicted box costs (found out on the device, redone inside cm_wait in a box around
            # the bounds the first attempt measured) and what the frame behind it costs (its splitters are the wide frame's:
            # usually handed back once more): host-observed, nothing else in flight.
            miss, after = [], []
            for it in range(6):
                for fr, dv, sink in ((frames[0], dev_frames[0], None), (frames[1 % K], dev_frames[1 % K], None),
                                     (wide, dev_wide, miss), (frames[2 % K], dev_frames[2 % K], after)):
                    for k, s in enumerate(fr):
                        c0.set_transform(k, s.q_xyzw, s.t_xyz)
                        c0.submit_device(k, dv[k].data_ptr(), s.n, s.point_step, s.off_x, s.off_y, s.off_z, s.off_i)
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    c0.merge_voxelize_async(cparams)
                    rr = c0.wait()
                    if sink is not None and it >= 1:
                        sink.append((time.perf_counter() - ta, bool(rr.path_flags & capi.PATH_REDONE)))
            out["config"]["box_miss_frame_ms"] = {
                "frame_that_leaves_the_box": 1e3 * float(np.median([t for t, _ in miss])), "redone": all(r for _, r in miss),
                "frame_behind_it": 1e3 * float(np.median([t for t, _ in after])), "frame_behind_it_redone": sum(r for _, r in after) / len(after),
                "note": "enqueue -> result on the host, one frame at a time; compare latency_one_frame_ms.host_enqueue_to_result"}

        # Per-kernel HIP-event timing on the same stream, same inputs (separate frames so the event
        # records do not sit inside the throughput measurement above).
        cmp = capi.CloudMerger(max_points_total=n_in, max_sensors=n_sens, device=local_rank,
                               flags=capi.FLAG_PROFILE | capi.FLAG_OCCUPANCY)
        cmp.set_stream(stream.cuda_stream)
        for k, s in enumerate(sensors):
            cmp.set_transform(k, s.q_xyzw, s.t_xyz)
        acc, order, dev_ms = {}, [], []
        nprof = args.profile_frames + 5
        for it in range(nprof):
            # (the stream's frames in turn, ending on frame 0 — the one the parity gate below compares with the oracle)
            fi = (it - (nprof - 1)) % K
            for k, s in enumerate(frames[fi]):
                cmp.set_transform(k, s.q_xyzw, s.t_xyz)
                cmp.submit_device(k, dev_frames[fi][k].data_ptr(), s.n, s.point_step, s.off_x, s.off_y, s.off_z, s.off_i)
            cmp.merge_voxelize_async(cparams)
            r = cmp.wait()
            if it < 5 or (r.path_flags & capi.PATH_REDONE):
                continue
            dev_ms.append(r.device_ms)
            for name, ms in cmp.stage_times():
                if name not in acc:
                    acc[name] = [0.0, 0]
                    order.append(name)
                acc[name][0] += ms
                acc[name][1] += 1
        nf = max(1, len(dev_ms))
        t_device_ms = float(np.median(dev_ms))
        n_out0 = int(r.n_out)
        b_alg = 16.0 * n_in + 16.0 * n_out0             # read each point once, write each voxel once
        # avg_us is event-to-event: the kernel plus the dependent-launch gap behind it and the event record itself;
        # rocprofv3's kernel_stats.csv (profiles/) shows the kernels alone.
        kernels = [{"name": n, "launches_per_frame": acc[n][1] / nf, "avg_us": 1e3 * acc[n][0] / acc[n][1],
                    "us_per_frame": 1e3 * acc[n][0] / nf} for n in order]
        dom = max(kernels, key=lambda k: k["us_per_frame"])
        t_alone_ms = float(np.median(dev1)) if dev1 else t_device_ms
        alone = b_alg / (t_alone_ms * 1e-3) / 1e9
        # HBM traffic per frame from the PMC counters: collected by scripts/pmc_traffic.sh in separate
        # rocprofv3 --pmc passes of this same command and committed under profiles/ (counters cannot
        # be read from inside the process). gfx950's FETCH_SIZE counts wide reads at half their
        # bytes, so the corrected figure (fetch x2 + write) is quoted; raw kept beside it.
        traffic, traffic_src = None, None
        suffix = "_bucket" if res.path_flags & 2 else ""
        dense = "_dense" if (args.config == 3 and args.dense) else ""
        # A file is only quoted for the build it was measured on (its "source_hash" = kernel_source_hash() of that build).
        src_hash = kernel_source_hash()
        for rnd in ("r3", "r2", "r1"):
            tpath = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_cfg{args.config}{dense}_minpts{args.min_pts}{suffix}.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                same_route = ("quantile" in tj.get("path", "")) == bool(res.path_flags & capi.PATH_QUANTILE)
                if tj.get("source_hash") == src_hash and not same_route:
                    traffic_src = {"not_this_route": os.path.relpath(tpath, ROOT), "file_path": tj.get("path", "")[:60],
                                   "note": "the file was measured on the other route of the bucket path (quantile passes vs fixed grid): not quoted"}
                    break
                if tj.get("source_hash") != src_hash:
                    traffic_src = {"stale_file": os.path.relpath(tpath, ROOT), "file_source_hash": tj.get("source_hash"),
                                   "this_build": src_hash, "stale_traffic_high": tj["traffic_high"],
                                   "note": "measured on another build of the kernels: not quoted (regenerate with scripts/pmc_traffic.sh)"}
                    break
                traffic = tj["traffic_high"]
                traffic_src = {"file": os.path.relpath(tpath, ROOT), "fetch_raw": tj["fetch_raw"], "write": tj["write"],
                               "fetch_corrected_x2": tj["fetch_x2"], "frames_averaged": tj["frames"], "source_hash": src_hash}
                break
        # The path is a sequence of 6 (bucket path) to 11 dependent launches per frame, so the roofline is quoted for
        # the frame: algorithmic bytes (16*N_in + 16*M) over the time a frame takes in
        # the timed region above (independent frames overlap on separate streams).
        achieved = (16.0 * n_in + 16.0 * n_out) * args.steps / elapsed / 1e9
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "peak_measured_copy": HBM_COPY_GBS, "frac_of_measured_copy": achieved / HBM_COPY_GBS,
            "scope": f"whole frame pipeline over the timed region ({inflight} independent frames in flight): "
                     "16*N_in + 16*M algorithmic bytes per frame / (elapsed / steps)",
            "algorithmic_bytes_per_frame": 16.0 * n_in + 16.0 * n_out,
            "one_frame_alone": {
                "note": "one frame alone on the GPU (= --inflight 1): t_device_ms by one pair of events around the frame's "
                        "launches; the per-kernel durations by HIP events between the launches of a profiling context "
                        "(event-to-event: kernel + launch gap + the event), to compare with "
                        "profiles/*inflight1_kernel_stats.csv (rocprofv3 --kernel-trace --stats)",
                "t_device_ms": t_alone_ms, "t_device_ms_with_per_kernel_events": t_device_ms,
                "achieved": alone, "frac": alone / HBM_PEAK_GBS,
                "dominant_kernel": dom["name"],
                # the contract's literal per-kernel figure: the frame's algorithmic bytes over the time the dominant kernel
                # takes per frame (all its launches of a frame: us_per_frame below; the headline `achieved` above charges
                # the whole frame instead)
                "dominant_kernel_achieved": b_alg / (dom["us_per_frame"] * 1e-6) / 1e9,
                "dominant_kernel_frac": b_alg / (dom["us_per_frame"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "kernels": kernels},
        }
        out_gpu = cmp.result(r.n_out)
        cells_gpu, counts_gpu = cmp.cells(r.n_out)
        fs = cmp.frame_stats()
        out["config"]["frame_stats_frame0"] = {"n_in": fs["n_in"], "n_kept": fs["n_kept"], "bytes_algorithmic": fs["bytes_algorithmic"]}
        cmp.close()

        # PCIe-inclusive figures (never `value`): PointCloud2 payloads in host memory in, compact
        # result in host memory out, through cm_submit_cloud / cm_result_copy — from pageable memory
        # (what a ROS callback hands over) and from pinned staging buffers (cm_host_alloc).
        if world == 1 and not args.no_e2e:
            import copy
            def e2e(cloud_list, reps=7):
                cme = capi.CloudMerger(max_points_total=n_in, max_sensors=n_sens, device=local_rank)
                for k, s in enumerate(sensors):
                    cme.set_transform(k, s.q_xyzw, s.t_xyz)
                ts = []
                for it in range(reps):
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    for k, s in enumerate(cloud_list):
                        cme.submit(k, s)
                    re = cme.merge_voxelize(params)
                    cme.result(re.n_out)
                    ts.append(time.perf_counter() - ta)
                cme.close()
                return float(np.median(ts[2:]))
            t_page = e2e(sensors)
            Ke = min(K, 3)
            holders, pinned_frames = [], []
            for fr in frames[:Ke]:
                pf = []
                for s in fr:
                    raw = np.ascontiguousarray(s.data).view(np.uint8).reshape(-1)
                    h = capi.pinned_array(raw.nbytes)
                    h.array[:] = raw
                    holders.append(h)
                    ps = copy.copy(s)
                    ps.data = h.array
                    pf.append((ps, h.ptr.value))
                pinned_frames.append(pf)
            t_pin = e2e([p[0] for p in pinned_frames[0]])
            # pipelined: H2D of frame n+1 (cm_submit_cloud_async, the slots' own copy streams) beside the kernels of frame n
            # beside the D2H of frame n-1 (cm_result_copy_async), over P contexts; every payload from pinned memory,
            # every result into pinned memory
            P = 3
            ctxs = [capi.CloudMerger(max_points_total=n_in, max_sensors=n_sens, device=local_rank) for _ in range(P)]
            outs = [capi.pinned_array(16 * n_in) for _ in range(P)]
            def pipe(n):
                pend, busy = [], [False] * P
                for i in range(n + P):
                    if i < n:
                        q = i % P
                        if busy[q]:
                            ctxs[q].sync(); busy[q] = False          # its last result has landed, its inputs are free again
                        for k, (ps, ptr) in enumerate(pinned_frames[i % Ke]):
                            ctxs[q].set_transform(k, ps.q_xyzw, ps.t_xyz)
                            ctxs[q].submit_async(k, ps, host_ptr=ptr)
                        ctxs[q].merge_voxelize_async(cparams)
                        pend.append(q)
                    if pend and (len(pend) == P or i >= n):
                        q = pend.pop(0)
                        rr = ctxs[q].wait()
                        ctxs[q].result_async(outs[q].ptr.value, n_in)
                        busy[q] = True
                for q in range(P):
                    ctxs[q].sync()
            pipe(6)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            n_pipe = 40
            pipe(n_pipe)
            t_pipe = (time.perf_counter() - tp0) / n_pipe
            for cx in ctxs:
                cx.close()
            for h in holders + outs:
                h.free()
            out["config"]["e2e_host_buffers"] = {
                "pageable_ms_per_step": 1e3 * t_page, "pageable_points_per_s": n_in / t_page,
                "pinned_ms_per_step": 1e3 * t_pin, "pinned_points_per_s": n_in / t_pin,
                "pipelined_ms_per_step": 1e3 * t_pipe, "pipelined_points_per_s": n_in / t_pipe,
                "h2d_bytes_per_frame": sum(s.n * s.point_step for s in sensors),
                "note": "host PointCloud2 payloads -> HBM -> result in host memory (H2D + D2H over PCIe included); pageable / "
                        "pinned: one frame at a time through cm_submit_cloud + cm_merge_voxelize + cm_result_copy; pipelined: "
                        "cm_submit_cloud_async / cm_merge_voxelize_async / cm_result_copy_async over 3 contexts from and to pinned "
                        "memory (the copy of frame n+1 beside the kernels of frame n beside the copy-out of frame n-1). "
                        "Reported beside, never as, value"}

        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle
            reps, times = 5, []
            for it in range(reps):
                st, _, ref, rep = oracle.merge_voxelize(sensors, params, threads=6, stable=False, want_merged=False)
                times.append(rep.t_total_s)
            st1, _, _, rep1 = oracle.merge_voxelize(sensors, params, threads=1, stable=False, want_merged=False)
            t_cpu = float(np.median(times))
            # parity gate on the frame that was just timed (oracle with a stable sort: PCL leaves the order of
            # a voxel's points to std::sort; the stable order is the one the device path reproduces)
            st, _, ref, rep = oracle.merge_voxelize(sensors, params, threads=6, stable=True, want_merged=False)
            ok = (st == oracle.OK and rep.n_out == n_out0 and np.array_equal(rep.cells, cells_gpu)
                  and np.array_equal(rep.counts, counts_gpu))
            g4 = np.stack([out_gpu["x"], out_gpu["y"], out_gpu["z"], out_gpu["intensity"]], 1)
            r4 = np.stack([ref["x"], ref["y"], ref["z"], ref["intensity"]], 1)
            dx = float(np.abs(g4[:, :3].astype(np.float64) - r4[:, :3].astype(np.float64)).max()) if ok else None
            bits = bool(ok and np.array_equal(np.ascontiguousarray(g4, np.float32).view(np.uint32),
                                               np.ascontiguousarray(r4, np.float32).view(np.uint32)))
            out["cpu_baseline"] = {
                "value": n_in / t_cpu, "unit": "points/s", "cores": int(rep.threads_used), "kind": "port",
                "sample": f"{reps} full frames of the same workload (frame 0 of the stream, {n_in} points), median; CPU restatement "
                          "of PCL 1.8.1 semantics (oracle/), not libpcl; ingest+transform+crop on "
                          f"{int(rep.threads_used)} threads, concat+VoxelGrid on 1 thread like the reference",
                "ms_per_frame": 1e3 * t_cpu, "single_thread_value": n_in / rep1.t_total_s,
                "breakdown_ms": {"ingest": 1e3 * rep.t_ingest_s, "transform_crop": 1e3 * rep.t_transform_crop_s,
                                 "concat": 1e3 * rep.t_concat_s, "voxelgrid": 1e3 * rep.t_voxel_s},
                "host_cpus": os.cpu_count(),
                "gpu_over_cpu": (args.steps * n_in / elapsed) / (n_in / t_cpu),
            }
            out["parity"] = {"occupancy_bit_exact": bool(ok), "max_abs_dxyz_m": dx, "centroids_bit_exact": bits,
                             "frame": "frame 0 of the stream"}
        print(json.dumps(out))
    for c in cms:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
