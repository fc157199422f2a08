"""Single fused cloud across GPUs (BASELINE config 5).

One process per GPU. Every rank voxelises its share of the sensors into a partial table of
per-voxel sums (thresholding deferred), the tables are all-gathered — the one real exchange step of
the path, RCCL over xGMI through torch.distributed's "nccl" backend (gloo on CPU in the tests) —
and each rank merges them by voxel index, adds, thresholds and divides. xGMI is a full mesh of
point-to-point links, so an all-gather lets every GPU push its table on all seven links at once.

Frame-sharded operation (bench.py, cloudmerge_replay --shard) needs none of this.
"""
import numpy as np

from .capi import ENTRY_DTYPE

ENTRY_WORDS = 8          # cm_partial_entry = 32 bytes


def shard_sensors(n_sensors, rank, world):
    """Sensor s belongs to rank s % world (every rank gets a contiguous-in-time share of the work)."""
    return [s for s in range(n_sensors) if s % world == rank]


def allreduce_bounds(dist, mn, mx, n_valid, device="cpu"):
    """Global min/max of the fused cloud from the ranks' local ones (2 tiny all-reduces).
    Ranks without valid points contribute +inf/-inf. Returns None if no rank has any point."""
    import torch
    lo = torch.tensor(np.asarray(mn, dtype=np.float32) if n_valid else np.full(3, np.inf, np.float32), device=device)
    hi = torch.tensor(np.asarray(mx, dtype=np.float32) if n_valid else np.full(3, -np.inf, np.float32), device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    lo, hi = lo.cpu().numpy(), hi.cpu().numpy()
    if not np.all(np.isfinite(lo)):
        return None
    return np.concatenate([lo, hi]).astype(np.float32)


def allgather_tables(dist, table, n_entries, world):
    """table: int32 tensor [capacity >= n_entries, 8] (cm_partial_entry rows) on this rank's device.
    Exchanges the lengths first, pads to the longest table, all-gathers once.
    Returns (gathered [world, max_n, 8] tensor, list of lengths)."""
    import torch
    dev = table.device
    mine = torch.tensor([int(n_entries)], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, mine)
    counts = [int(v.item()) for v in lens]
    max_n = max(max(counts), 1)
    send = torch.zeros((max_n, ENTRY_WORDS), dtype=torch.int32, device=dev)
    if n_entries:
        send[:n_entries] = table[:n_entries]
    if dev.type == "cuda" and hasattr(dist, "all_gather_into_tensor"):
        gathered = torch.empty((world, max_n, ENTRY_WORDS), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(gathered.view(-1), send.view(-1))
    else:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        gathered = torch.stack(parts)
    return gathered, counts


def merge_tables_numpy(tables, min_pts):
    """CPU reference of the device merge (cm_merge_tables), used by the gloo tests: concatenate in
    rank order, stable sort by key, add per key (fp32, in that order), threshold, divide.
    tables: list of ENTRY_DTYPE arrays. Returns (keys, counts, centroids (n,4) float32)."""
    cat = np.concatenate([np.asarray(t, dtype=ENTRY_DTYPE) for t in tables]) if tables else np.zeros(0, ENTRY_DTYPE)
    if len(cat) == 0:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 4), np.float32)
    order = np.argsort(cat["key"], kind="stable")
    cat = cat[order]
    head = np.flatnonzero(np.r_[True, cat["key"][1:] != cat["key"][:-1]])
    cnt = np.add.reduceat(cat["count"].astype(np.uint64), head).astype(np.uint32)
    sums = np.zeros((len(head), 4), dtype=np.float32)
    ends = np.r_[head[1:], len(cat)]
    vals = np.stack([cat["sx"], cat["sy"], cat["sz"], cat["si"]], axis=1)
    for r, (a, b) in enumerate(zip(head, ends)):           # fp32 running sum in rank order
        acc = vals[a].copy()
        for k in range(a + 1, b):
            acc = (acc + vals[k]).astype(np.float32)
        sums[r] = acc
    keep = cnt >= max(1, int(min_pts))
    cent = (sums[keep] / cnt[keep, None].astype(np.float32)).astype(np.float32)
    return cat["key"][head][keep], cnt[keep], cent


def fused_cloud(cm, params, dist, rank, world, device, host_exchange=False, times=None):
    """Whole exchange on one rank whose sensors are already submitted to `cm` (a capi.CloudMerger on
    `device`). Returns the cm_result of the merged cloud; read it with cm.result()/cm.cells().

    host_exchange: the collectives run on CPU tensors (the gloo rehearsal: several ranks on one GPU); the
    tables go device -> host -> all-gather -> device. With "nccl" (= RCCL) they stay in HBM and travel over xGMI.
    times: optional dict that receives the wall time of the three steps in ms (partial, exchange, merge).
    A rank whose cm_merge_partial fails (its sensors not fresh, a bad argument ...) must not leave the others
    hanging in the all-gather: the ranks all-reduce a status word first and every rank raises together."""
    import time
    import torch
    from .capi import CloudMergeError
    xdev = torch.device("cpu") if host_exchange else device
    t0 = time.perf_counter()
    bounds, err = None, None
    try:
        if params.crop_min is None:
            mn, mx, n_valid = cm.local_bounds(params)
            bounds = allreduce_bounds(dist, mn, mx, n_valid, device=xdev) if world > 1 else \
                (np.concatenate([mn, mx]) if n_valid else None)
        part = cm.merge_partial(params, bounds)
        status = int(part.status)
    except CloudMergeError as e:               # (a rank-local failure: report it to everybody below)
        err, status, part = e, int(e.status), None
    bad = status not in (0, 1)                 # OK or EMPTY_INPUT (an empty share is a valid share)
    if world > 1:
        flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        any_bad = bool(flag.item())
    else:
        any_bad = bad
    if any_bad:
        raise err if err is not None else CloudMergeError(status if bad else -5, "fused_cloud: another rank's partial table failed")
    n = int(part.n_out) if part.status == 0 else 0
    # (torch.empty, and nothing of torch's pending on the buffer: the library copies on ITS stream, which need not be
    # torch's — a zero-fill still queued on torch's stream would race with that copy)
    table = torch.empty((max(n, 1), ENTRY_WORDS), dtype=torch.int32, device=device)
    torch.cuda.synchronize(device)
    if n:
        cm.partial_to_device(table.data_ptr(), n)
    else:
        table.zero_()
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    if world > 1:
        gathered, counts = allgather_tables(dist, table.to(xdev) if host_exchange else table, n, world)
        if host_exchange:
            gathered = gathered.to(device)
    else:
        gathered, counts = table.unsqueeze(0), [n]
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    ptrs = [gathered[r].data_ptr() for r in range(world)]
    res = cm.merge_tables(ptrs, counts, params)
    t3 = time.perf_counter()
    if times is not None:
        times.update(partial_ms=1e3 * (t1 - t0), exchange_ms=1e3 * (t2 - t1), merge_ms=1e3 * (t3 - t2),
                     table_entries=n, gathered_entries=int(sum(counts)))
    return res
