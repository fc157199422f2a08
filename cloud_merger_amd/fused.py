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


class _DeviceWords:
    """n int32 words of device memory at `ptr`, for torch.as_tensor (zero-copy through __cuda_array_interface__)."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def alias_device_words(ptr, n_words, device):
    """A torch int32 tensor over device memory the library owns (no copy); None if this torch build cannot do that."""
    import torch
    try:
        t = torch.as_tensor(_DeviceWords(ptr, n_words), device=device)
        return t if t.data_ptr() == int(ptr) else None
    except Exception:            # noqa: BLE001  (any refusal: the caller stages a copy instead)
        return None


def allgather_tables(dist, table, n_entries, world, send_view=None):
    """table: int32 tensor [capacity >= n_entries, 8] (cm_partial_entry rows) on this rank's device, or None when
    send_view is given. Exchanges the lengths first, then all-gathers the tables padded to the longest.
    send_view(max_n): a tensor of max_n rows that STARTS with this rank's table (the library's own buffer, aliased: no
    staging copy) or None when the buffer is shorter than max_n rows.
    Returns (gathered [world, max_n, 8] tensor, list of lengths)."""
    import torch
    dev = table.device if table is not None else send_view.device
    mine = torch.tensor([int(n_entries)], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, mine)
    counts = [int(v.item()) for v in lens]
    max_n = max(max(counts), 1)
    send = send_view(max_n) if send_view is not None else None
    if send is None:
        send = torch.zeros((max_n, ENTRY_WORDS), dtype=torch.int32, device=dev)
        if n_entries:
            send[:n_entries] = table[:n_entries] if table is not None else send_view.stage(n_entries)
    if dev.type == "cuda" and hasattr(dist, "all_gather_into_tensor"):
        gathered = torch.empty((world, max_n, ENTRY_WORDS), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(gathered.view(-1), send.reshape(-1))
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
    t1 = time.perf_counter()
    # The table stays where cm_merge_partial left it (the library's buffer: one entry per padded input slot at least). On a
    # device exchange it is sent from there — aliased as a torch tensor, no staging copy — and the collective is ordered
    # behind the library's kernels by the stream both are enqueued on (cm.set_stream(torch's current stream): the caller's
    # job, as in bench.py); nothing waits on the host between the table, the all-gather and the merge.
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if not host_exchange else None
    ptr, n_lib = cm.partial_device() if n else (0, 0)
    cap_rows = int(cm.max_points_total)
    if world > 1 and not host_exchange:
        class _View:
            device = torch.device(device)

            def __call__(self, max_n):
                if not n or max_n > cap_rows:
                    return None
                a = alias_device_words(ptr, max_n * ENTRY_WORDS, device)
                return None if a is None else a.view(max_n, ENTRY_WORDS)

            @staticmethod
            def stage(n_rows):
                tmp = torch.empty((n_rows, ENTRY_WORDS), dtype=torch.int32, device=device)
                cm.partial_to_device(tmp.data_ptr(), n_rows)
                return tmp
        ev[0].record()
        gathered, counts = allgather_tables(dist, None, n, world, send_view=_View())
        ev[1].record()
        ptrs = [gathered[r].data_ptr() for r in range(world)]
    elif world > 1:                                # the gloo rehearsal: device -> host -> all-gather -> device
        host = torch.from_numpy(cm.partial(n).view(np.int32).reshape(-1, ENTRY_WORDS).copy()) if n else \
            torch.zeros((1, ENTRY_WORDS), dtype=torch.int32)
        gathered, counts = allgather_tables(dist, host, n, world)
        gathered = gathered.to(device)
        torch.cuda.current_stream(device).synchronize()
        ptrs = [gathered[r].data_ptr() for r in range(world)]
    else:                                          # one rank: its own table, in place
        counts = [n]
        if n:
            ptrs = [ptr]
        else:
            gathered = torch.zeros((1, ENTRY_WORDS), dtype=torch.int32, device=device)
            torch.cuda.current_stream(device).synchronize()
            ptrs = [gathered.data_ptr()]
    t2 = time.perf_counter()
    res = cm.merge_tables(ptrs, counts, params)        # (blocking: it returns the merged cloud's counts)
    t3 = time.perf_counter()
    if times is not None:
        x_ms = ev[0].elapsed_time(ev[1]) if (ev is not None and world > 1) else 0.0   # the all-gather on the device
        times.update(partial_ms=1e3 * (t1 - t0), exchange_ms=1e3 * (t2 - t1) + x_ms,
                     merge_ms=max(0.0, 1e3 * (t3 - t2) - x_ms), table_entries=n, gathered_entries=int(sum(counts)))
    return res
