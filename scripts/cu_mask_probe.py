"""Frames in flight on streams that own disjoint sets of CUs (hipExtStreamCreateWithCUMask) against ordinary streams:
do a bandwidth-bound kernel of one frame and the instruction-bound finish of another overlap better when they cannot
take each other's CUs? usage: cu_mask_probe.py [n_streams]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cloud_merger_amd import capi, synth

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)
K = 6
frames = [synth.config2_stream(k, min_pts=2)[0] for k in range(K)]
params = synth.config2(n_per_sensor=8, min_pts=2)[1]
cp = capi.make_params(params)
dv = [[torch.from_numpy(np.ascontiguousarray(s.data).view(np.uint8).reshape(-1)).to(dev) for s in fr] for fr in frames]
torch.cuda.synchronize()
N_CU = 256


def masked_stream(cus):
    words = (ctypes.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return st


def enqueue(c, i):
    fr, d = frames[i % K], dv[i % K]
    for k, s in enumerate(fr):
        c.set_transform(k, s.q_xyzw, s.t_xyz)
        c.submit_device(k, d[k].data_ptr(), s.n, s.point_step, s.off_x, s.off_y, s.off_z, s.off_i)
    c.merge_voxelize_async(cp)


def run(name, streams, n=400):
    inflight = len(streams)
    cms = []
    for q in range(inflight):
        c = capi.CloudMerger(max_points_total=4_000_000, max_sensors=4, device=0)
        c.set_stream(streams[q])
        cms.append(c)
    for q, c in enumerate(cms):
        enqueue(c, q); c.wait()
    best = None
    for rep in range(3):
        issued = done = 0
        t0 = time.perf_counter()
        while done < n:
            while issued < n and issued - done < inflight:
                enqueue(cms[issued % inflight], issued); issued += 1
            r = cms[done % inflight].wait()
            assert r.status == capi.OK
            done += 1
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        best = el if best is None else min(best, el)
    for c in cms:
        c.close()
    print("%-44s %d in flight: %.4f ms/frame" % (name, inflight, 1e3 * best), flush=True)


ns = int(sys.argv[1]) if len(sys.argv) > 1 else 3
plain = [torch.cuda.Stream(device=dev) for _ in range(ns)]
run("ordinary streams", [s.cuda_stream for s in plain])
run("CU ranges (contiguous thirds)", [masked_stream(range(q * N_CU // ns, (q + 1) * N_CU // ns)).value for q in range(ns)])
run("CUs interleaved (c % n)", [masked_stream([c for c in range(N_CU) if c % ns == q]).value for q in range(ns)])
run("all CUs on every masked stream", [masked_stream(range(N_CU)).value for q in range(ns)])
if ns == 3:
    run("2 ordinary streams", [s.cuda_stream for s in plain[:2]])
    run("halves", [masked_stream(range(0, 128)).value, masked_stream(range(128, 256)).value])
    four = [masked_stream(range(q * 64, (q + 1) * 64)).value for q in range(4)]
    run("quarters", four)
    six = [masked_stream([c for c in range(N_CU) if (c // 8) % 3 == q % 3]).value for q in range(6)]
    run("6 streams on thirds (groups of 8 CUs)", six)
