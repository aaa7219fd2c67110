"""Pair-level data parallelism over the GPUs of one node (BASELINE.json configs[4]).

Image pairs are independent units — the reference itself only ever parallelises over pairs (a process pool,
/root/reference/eval.py:355-359; one SLURM job per scene, slurm_scripts/eval_mdrp_spawn_all.sh:78-80).  Each rank
(one process per GPU, torch.distributed, backend "nccl" = RCCL over xGMI) owns a contiguous block of ceil(P/G) pairs and
holds ONLY that block; there is NO collective on the data path.  The only exchange is one all_gather_into_tensor of the
fixed-size result records (136 B per pair; 100k pairs = 13.6 MB, far below a single xGMI link) — device to device when
the records are on the GPU — and, on request, of the N-byte inlier masks.
"""
import numpy as np

from . import _capi

RECORD_BYTES = _capi.RESULT_DTYPE.itemsize


def shard_bounds(total, rank, world):
    """contiguous block [lo, hi) of `total` pairs owned by `rank`; every block has ceil(total/world) slots"""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total), per


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _unpad(allbuf, total, world, per):
    """(world * per, ...) gathered blocks -> the `total` real rows in pair order"""
    parts = [allbuf[r * per: r * per + (shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0])] for r in range(world)]
    return np.concatenate(parts) if parts else allbuf[:0]


def gather_records(local, total, group=None, device=None, force_collective=False):
    """All `total` result records on every rank.  `local`: this rank's records, either a numpy RESULT_DTYPE array or a
    (rows, 136) uint8 torch tensor already on the GPU (then the all-gather runs device to device).  Blocks are padded to
    ceil(total / world) rows for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    rank, world = _world(group)
    per = (total + world - 1) // world
    single = world == 1 and not force_collective  # (force_collective: run the all-gather even with one rank — the tests' way to cover the RCCL path on one GPU)
    if isinstance(local, np.ndarray):
        if single:
            return local[:total].copy()
        mine = torch.zeros((per, RECORD_BYTES), dtype=torch.uint8)
        if len(local):
            mine[: len(local)] = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(len(local), RECORD_BYTES).copy())
        if device is not None:
            mine = mine.to(device)
    else:
        if single:
            return local[:total].cpu().numpy().reshape(-1).view(_capi.RESULT_DTYPE).copy()
        mine = local
        if mine.shape[0] != per:
            padded = torch.zeros((per, RECORD_BYTES), dtype=torch.uint8, device=mine.device)
            padded[: mine.shape[0]] = mine
            mine = padded
    out = torch.empty((world * per, RECORD_BYTES), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
    rows = _unpad(out.cpu().numpy(), total, world, per)
    return np.ascontiguousarray(rows).reshape(-1).view(_capi.RESULT_DTYPE)


gather_results = gather_records  # older name


def gather_masks(local_mask, total, group=None, device=None):
    """all_gather of the (pairs, N) uint8 inlier masks (numpy array or device tensor)"""
    import torch
    import torch.distributed as dist

    rank, world = _world(group)
    per = (total + world - 1) // world
    is_np = isinstance(local_mask, np.ndarray)
    if world == 1:
        return local_mask[:total].copy() if is_np else local_mask[:total].cpu().numpy()
    n = local_mask.shape[1]
    mine = torch.zeros((per, n), dtype=torch.uint8, device=(device if is_np else local_mask.device) or "cpu")
    if local_mask.shape[0]:
        mine[: local_mask.shape[0]] = torch.from_numpy(np.ascontiguousarray(local_mask)).to(mine.device) if is_np else local_mask
    out = torch.empty((world * per, n), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return _unpad(out.cpu().numpy(), total, world, per)


def estimate_local_shard(kind, total, x1, x2, d1, d2, ransac_opt=None, bundle_opt=None, n_per_pair=None, cam1=None, cam2=None,
                         group=None, local_fn=None, want_mask=False):
    """BASELINE.json configs[4]: `total` pairs over the ranks of `group`; THIS rank passes only its own block — the pairs
    shard_bounds(total, rank, world) names, (rows, N, 2) / (rows, N) arrays plus optional per-pair counts and cameras —
    estimates it on its GPU and receives all `total` records (and masks) back.
    local_fn(kind, x1, x2, d1, d2, ropt, bopt, n_per_pair, cam1, cam2) -> (records, mask) defaults to this rank's HIP
    handle; the CPU tests inject a function to exercise sharding and gather under gloo."""
    import torch

    rank, world = _world(group)
    lo, hi, per = shard_bounds(total, rank, world)
    rows = hi - lo
    if len(x1) != rows:   # d1 / d2 are None for the non-monodepth baselines (kind 3 / 5)
        raise ValueError(f"rank {rank} owns pairs [{lo}, {hi}) = {rows} rows, got {len(x1)}")
    ro = ransac_opt if isinstance(ransac_opt, _capi.RansacOpt) else _capi.ransac_opt_from_dict(ransac_opt)
    bo = bundle_opt if isinstance(bundle_opt, _capi.BundleOpt) else _capi.bundle_opt_from_dict(bundle_opt)
    device = None
    if local_fn is None:
        dev_index = torch.cuda.current_device()
        handle = _capi.default_handle(dev_index)
        device = torch.device("cuda", dev_index)

        def local_fn(kind, a, b, c, d, ro, bo, npp, c1, c2):
            return handle.estimate_batch(kind, a, b, c, d, ro, bo, npp, c1, c2)

    if rows > 0:
        res, mask = local_fn(kind, x1, x2, d1, d2, ro, bo, n_per_pair, cam1, cam2)
    else:
        res, mask = np.zeros(0, dtype=_capi.RESULT_DTYPE), np.zeros((0, np.shape(x1)[1] if np.ndim(x1) == 3 else 0), dtype=np.uint8)
    all_res = gather_records(res, total, group, device)
    if want_mask:
        return all_res, gather_masks(mask, total, group, device)
    return all_res


def estimate_local_shard_device(kind, total, x1, x2, d1, d2, ransac_opt=None, bundle_opt=None, n_per_pair=None, cam1=None, cam2=None,
                                group=None, handle=None, mask=None, force_collective=False):
    """The same on DEVICE-RESIDENT inputs, the way bench.py runs BASELINE configs[4]: x1, x2 (rows, N, 2) and d1, d2 (rows, N) are
    float64 CUDA tensors holding this rank's block; the estimate runs through mdrp_estimate_batch_async, the 136-byte records go
    device to device into the rank's slot (mdrp_copy_results_device) and from there through ONE all_gather_into_tensor over
    RCCL / xGMI — nothing crosses PCIe until the caller reads the gathered records.  `mask`: optional (rows, N) uint8 CUDA tensor
    that receives this rank's inlier masks.  Returns all `total` records as a numpy RESULT_DTYPE array."""
    import torch

    rank, world = _world(group)
    lo, hi, per = shard_bounds(total, rank, world)
    rows = hi - lo
    if x1.shape[0] != rows or not x1.is_cuda or x1.dtype != torch.float64:
        raise ValueError(f"rank {rank} owns pairs [{lo}, {hi}) = {rows} rows of float64 CUDA tensors, got {tuple(x1.shape)} {x1.dtype} on {x1.device}")
    ro = ransac_opt if isinstance(ransac_opt, _capi.RansacOpt) else _capi.ransac_opt_from_dict(ransac_opt)
    bo = bundle_opt if isinstance(bundle_opt, _capi.BundleOpt) else _capi.bundle_opt_from_dict(bundle_opt)
    dev = x1.device
    h = handle if handle is not None else _capi.default_handle(dev.index if dev.index is not None else torch.cuda.current_device())
    rec_local = torch.zeros((per, RECORD_BYTES), dtype=torch.uint8, device=dev)
    if mask is not None and (not mask.is_contiguous() or mask.dtype != torch.uint8 or tuple(mask.shape) != tuple(x1.shape[:2])):
        raise ValueError("mask must be a contiguous (rows, N) uint8 CUDA tensor")
    # Contiguous views FIRST, bound to locals that live until the results are copied out: .contiguous() of a strided tensor queues a copy
    # kernel on torch's stream and returns a temporary — the synchronize below must come after those copies, and the temporaries must not
    # be freed (and their memory reused by the caching allocator) while the handle's stream still reads them.
    x1c, x2c = x1.contiguous(), x2.contiguous()
    d1c = d1.contiguous() if d1 is not None else None
    d2c = d2.contiguous() if d2 is not None else None
    torch.cuda.current_stream(dev).synchronize()  # the inputs may have been produced on torch's stream; the handle runs on its own
    if rows > 0:
        n = x1c.shape[1]
        h.estimate_batch_device(kind, x1c.data_ptr(), x2c.data_ptr(), d1c.data_ptr() if d1c is not None else 0, d2c.data_ptr() if d2c is not None else 0,
                                rows, n, ro, bo, n_per_pair, cam1, cam2, mask.data_ptr() if mask is not None else None)
        h.copy_results_device(rec_local.data_ptr(), rows)  # returns after the handle's stream has drained
    del x1c, x2c, d1c, d2c  # (kept alive until here on purpose)
    return gather_records(rec_local, total, group, dev, force_collective=force_collective)


def estimate_sharded(kind, x1, x2, d1, d2, ransac_opt=None, bundle_opt=None, n_per_pair=None, cam1=None, cam2=None, group=None,
                     local_fn=None, want_mask=False):
    """Convenience for callers that hold the FULL arrays on every rank: slices this rank's block and calls
    estimate_local_shard.  (Large data sets should never be replicated: build each rank's block only.)"""
    rank, world = _world(group)
    total = len(x1)
    lo, hi, _ = shard_bounds(total, rank, world)
    sl = slice(lo, hi)
    return estimate_local_shard(kind, total, x1[sl], x2[sl], None if d1 is None else d1[sl], None if d2 is None else d2[sl], ransac_opt, bundle_opt,
                                None if n_per_pair is None else n_per_pair[sl], None if cam1 is None else cam1[sl],
                                None if cam2 is None else cam2[sl], group, local_fn, want_mask)


def device_identity(dev_index=None):
    """A string that names the PHYSICAL device behind a torch device ordinal (uuid when the runtime reports one, else PCI domain:bus:device
    + name): two ranks that print the same string share a GPU, whatever their ordinals say (ROCR_/HIP_VISIBLE_DEVICES masks renumber)."""
    import torch
    i = torch.cuda.current_device() if dev_index is None else int(dev_index)
    p = torch.cuda.get_device_properties(i)
    uuid = getattr(p, "uuid", None)
    pci = ":".join(str(getattr(p, k, "?")) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    return f"{uuid}|{pci}|{p.name}" if uuid is not None else f"{pci}|{p.name}"


def gather_device_identities(dev_index=None, group=None, force_collective=False):
    """[(rank, device_identity)] of every rank of the group (one all_gather_object), for the N > 1 bench line: the proof that N distinct GPUs took part
    (force_collective: run the collective even with one rank — the tests' way to cover the RCCL path on one GPU)"""
    import torch.distributed as dist
    rank, world = _world(group)
    mine = (rank, device_identity(dev_index))
    if world == 1 and not force_collective:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine, group=group)
    return out


def local_device_index(local_rank):
    """device ordinal of a rank: LOCAL_RANK modulo the devices this process can see — with one visible device per rank
    (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES set by the launcher) every rank uses ordinal 0"""
    import torch
    n = torch.cuda.device_count()
    if n <= 0:
        raise RuntimeError("no GPU visible to this rank")
    return int(local_rank) % n
