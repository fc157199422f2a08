"""Where the pipelined loop of bench.py spends its time on the host: per-frame enqueue time (4 x set_transform +
4 x cm_submit_cloud_device + cm_merge_voxelize_async) and wait time, for several depths of frames in flight, with the
enqueues issued from the main thread or from one thread per context (ctypes releases the GIL inside the library)."""
import os, sys, time, threading, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cloud_merger_amd import capi, synth

dev = torch.device("cuda", 0)
K = 6
frames = [synth.config2_stream(k, min_pts=2)[0] for k in range(K)]
params = synth.config2(n_per_sensor=8, min_pts=2)[1]
cp = capi.make_params(params)
dv = [[torch.from_numpy(np.ascontiguousarray(s.data).view(np.uint8).reshape(-1)).to(dev) for s in fr] for fr in frames]
torch.cuda.synchronize()


def enqueue(c, i, static):
    fr, d = frames[0 if static else i % K], dv[0 if static else i % K]
    for k, s in enumerate(fr):
        c.set_transform(k, s.q_xyzw, s.t_xyz)
        c.submit_device(k, d[k].data_ptr(), s.n, s.point_step, s.off_x, s.off_y, s.off_z, s.off_i)
    c.merge_voxelize_async(cp)


def run(inflight, n, static, threaded):
    streams = [torch.cuda.Stream(device=dev) for _ in range(inflight)]
    cms = []
    for q in range(inflight):
        c = capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, device=0)
        c.set_stream(streams[q].cuda_stream)
        cms.append(c)
    for q, c in enumerate(cms):
        enqueue(c, q, static); c.wait()
    te, tw = [], []
    if not threaded:
        for rep in range(2):
            issued = done = 0
            te, tw = [], []
            t0 = time.perf_counter()
            while done < n:
                while issued < n and issued - done < inflight:
                    a = time.perf_counter()
                    enqueue(cms[issued % inflight], issued, static)
                    te.append(time.perf_counter() - a)
                    issued += 1
                a = time.perf_counter()
                r = cms[done % inflight].wait()
                tw.append(time.perf_counter() - a)
                assert r.status == capi.OK
                done += 1
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
    else:
        # one thread per context: each loops enqueue -> wait on its own frames (i = q, q + inflight, ...)
        for rep in range(2):
            def worker(q):
                c = cms[q]
                for i in range(q, n, inflight):
                    enqueue(c, i, static)
                    r = c.wait()
                    assert r.status == capi.OK
            ths = [threading.Thread(target=worker, args=(q,)) for q in range(inflight)]
            t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
    for c in cms:
        c.close()
    msg = "inflight %d %s %s: %.4f ms/frame" % (inflight, "static" if static else "moving", "threads" if threaded else "main", 1e3 * el / n)
    if te:
        msg += "  enqueue median %.1f us mean %.1f us; wait median %.1f us mean %.1f us" % (
            1e6 * np.median(te), 1e6 * np.mean(te), 1e6 * np.median(tw), 1e6 * np.mean(tw))
    print(msg, flush=True)


for static in (True, False):
    for inflight in (1, 2, 3, 4, 6):
        run(inflight, 300, static, False)
    for inflight in (3, 4, 6):
        run(inflight, 300, static, True)
