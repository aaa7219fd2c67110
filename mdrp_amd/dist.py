"""Pair-level data parallelism over the GPUs of one node.

Image pairs are independent units — the reference itself only ever parallelises over pairs (a process pool,
/root/reference/eval.py:355-359; one SLURM job per scene, slurm_scripts/eval_mdrp_spawn_all.sh:78-80).  Each rank
(one process per GPU, torch.distributed, backend "nccl" = RCCL over xGMI) takes a contiguous block of ceil(P/G) pairs;
there is NO collective on the data path.  The only exchange is one all_gather of the fixed-size result records
(136 B per pair; 100k pairs = 13.6 MB, far below a single xGMI link), optionally the N-byte inlier masks.
"""
import numpy as np

from . import _capi


def shard_bounds(total, rank, world):
    """contiguous block [lo, hi) of `total` pairs owned by `rank`; every block has ceil(total/world) slots"""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total), per


def gather_results(local, total, group=None, device=None):
    """all_gather of RESULT_DTYPE records; local covers this rank's shard (len <= per).  Returns all `total` records."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local[:total].copy()
    _, _, per = shard_bounds(total, 0, world)
    buf = np.zeros(per, dtype=_capi.RESULT_DTYPE)
    buf[: len(local)] = local
    mine = torch.from_numpy(buf.view(np.uint8).reshape(per, -1).copy())
    if device is not None:
        mine = mine.to(device)
    out = torch.empty((world * per, mine.shape[1]), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    allrec = out.cpu().numpy().reshape(world, per, -1)
    parts = []
    for r in range(world):
        lo, hi, _ = shard_bounds(total, r, world)
        parts.append(np.ascontiguousarray(allrec[r, : hi - lo]).view(_capi.RESULT_DTYPE).reshape(-1))
    return np.concatenate(parts)


def gather_masks(local_mask, total, group=None, device=None):
    """all_gather of the (pairs, N) uint8 inlier masks"""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_mask[:total].copy()
    _, _, per = shard_bounds(total, 0, world)
    n = local_mask.shape[1]
    buf = np.zeros((per, n), dtype=np.uint8)
    buf[: len(local_mask)] = local_mask
    mine = torch.from_numpy(buf)
    if device is not None:
        mine = mine.to(device)
    out = torch.empty((world * per, n), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    allm = out.cpu().numpy().reshape(world, per, n)
    return np.concatenate([allm[r, : shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0]] for r in range(world)])


def estimate_sharded(kind, x1, x2, d1, d2, ransac_opt=None, bundle_opt=None, n_per_pair=None, cam1=None, cam2=None, group=None,
                     local_fn=None, want_mask=False):
    """Every rank passes the SAME full arrays (or at least its own block filled in); each estimates its block and all ranks
    receive all results.  local_fn(kind, x1, x2, d1, d2, ropt, bopt, n_per_pair, cam1, cam2) -> (results, mask) defaults to
    the HIP handle of this rank's device; tests inject a CPU function to exercise the sharding/gather logic under gloo."""
    import torch
    import torch.distributed as dist

    initialized = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if initialized else 0
    world = dist.get_world_size(group) if initialized else 1
    total = len(d1)
    lo, hi, _ = shard_bounds(total, rank, world)
    ro = ransac_opt if isinstance(ransac_opt, _capi.RansacOpt) else _capi.ransac_opt_from_dict(ransac_opt)
    bo = bundle_opt if isinstance(bundle_opt, _capi.BundleOpt) else _capi.bundle_opt_from_dict(bundle_opt)
    device = None
    if local_fn is None:
        dev_index = torch.cuda.current_device()
        handle = _capi.default_handle(dev_index)
        device = torch.device("cuda", dev_index)

        def local_fn(kind, a, b, c, d, ro, bo, npp, c1, c2):
            return handle.estimate_batch(kind, a, b, c, d, ro, bo, npp, c1, c2)

    sl = slice(lo, hi)
    if hi > lo:
        res, mask = local_fn(kind, x1[sl], x2[sl], d1[sl], d2[sl], ro, bo, None if n_per_pair is None else n_per_pair[sl],
                             None if cam1 is None else cam1[sl], None if cam2 is None else cam2[sl])
    else:
        res, mask = np.zeros(0, dtype=_capi.RESULT_DTYPE), np.zeros((0, d1.shape[1]), dtype=np.uint8)
    all_res = gather_results(res, total, group, device)
    if want_mask:
        return all_res, gather_masks(mask, total, group, device)
    return all_res
