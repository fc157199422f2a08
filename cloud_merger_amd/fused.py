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


def fused_cloud(cm, params, dist, rank, world, device):
    """Whole exchange on one rank whose sensors are already submitted to `cm` (a capi.CloudMerger on
    `device`). Returns the cm_result of the merged cloud; read it with cm.result()/cm.cells()."""
    import torch
    bounds = None
    if params.crop_min is None:
        mn, mx, n_valid = cm.local_bounds(params)
        bounds = allreduce_bounds(dist, mn, mx, n_valid, device=device) if world > 1 else \
            (np.concatenate([mn, mx]) if n_valid else None)
    part = cm.merge_partial(params, bounds)
    n = int(part.n_out) if part.status == 0 else 0
    table = torch.zeros((max(n, 1), ENTRY_WORDS), dtype=torch.int32, device=device)
    if n:
        cm.partial_to_device(table.data_ptr(), n)
    if world > 1:
        gathered, counts = allgather_tables(dist, table, n, world)
    else:
        gathered, counts = table.unsqueeze(0), [n]
    torch.cuda.synchronize(device)
    ptrs = [gathered[r].data_ptr() for r in range(world)]
    return cm.merge_tables(ptrs, counts, params)
