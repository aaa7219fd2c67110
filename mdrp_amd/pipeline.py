"""Several batches in flight on one GPU.

One call of the batched estimators walks a serial chain of phases (solve -> count -> bound -> score -> scan -> LO -> walk ->
final); its LM phases run at one or two wavefronts per SIMD and leave most issue slots of the chip idle.  Consecutive batches
are independent (the reference itself only ever parallelises over pairs, eval.py:355-359), so the next batch's solver and
sweeps can fill those slots: `BatchPipeline` keeps `depth` handles, each with its own streams and scratch buffers and its
own host thread (the C call releases the GIL), and hands batches to them round-robin.  Results come back in submission
order and are bit-identical to the sequential calls.  Measured on the benchmark shape (1024 pairs, N = 2000, 10^4
iterations): 88 k pairs/s with one batch in flight, 108 k with two, 105 k with three (tools/inflight_exp.py).
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _capi


class BatchPipeline:
    """pipe = BatchPipeline(depth=2); futures = [pipe.submit(kind, x1, x2, d1, d2, ropt, bopt, ...) for batch in batches];
    results = [f.result() for f in futures]   # (records, mask) per batch, in order"""

    def __init__(self, depth=2, device=0):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.depth, self.device = int(depth), int(device)
        self._local = threading.local()
        self._handles = []
        self._lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=self.depth, thread_name_prefix="mdrp-pipe")

    def _handle(self):
        h = getattr(self._local, "h", None)
        if h is None:
            h = self._local.h = _capi.Handle(self.device)
            with self._lock:
                self._handles.append(h)
        return h

    def submit(self, kind, x1, x2, d1, d2, ropt, bopt, n_per_pair=None, cam1=None, cam2=None, want_mask=True):
        """host (numpy) buffers; returns a Future of (records, mask) exactly as Handle.estimate_batch returns them"""
        ro = ropt if isinstance(ropt, _capi.RansacOpt) else _capi.ransac_opt_from_dict(ropt)
        bo = bopt if isinstance(bopt, _capi.BundleOpt) else _capi.bundle_opt_from_dict(bopt)
        return self._pool.submit(lambda: self._handle().estimate_batch(kind, x1, x2, d1, d2, ro, bo, n_per_pair, cam1, cam2, want_mask))

    def submit_device(self, kind, x1_ptr, x2_ptr, d1_ptr, d2_ptr, batch, n_max, ropt, bopt, n_per_pair=None, cam1=None, cam2=None,
                      mask_ptr=None):
        """device pointers (e.g. torch tensors' data_ptr()); returns a Future of the result records (numpy)"""
        ropt = _capi.ransac_opt_from_dict(ropt) if isinstance(ropt, dict) else ropt  # as submit() does
        bopt = _capi.bundle_opt_from_dict(bopt) if isinstance(bopt, dict) else bopt

        def run():
            h = self._handle()
            h.estimate_batch_device(kind, x1_ptr, x2_ptr, d1_ptr, d2_ptr, batch, n_max, ropt, bopt, n_per_pair, cam1, cam2, mask_ptr)
            return h.fetch_results(batch)
        return self._pool.submit(run)

    def map(self, kind, batches, ropt, bopt, **kw):
        """batches: iterable of (x1, x2, d1, d2) host arrays -> list of (records, mask) in order"""
        futs = [self.submit(kind, *b, ropt, bopt, **kw) for b in batches]
        return [f.result() for f in futs]

    def close(self):
        self._pool.shutdown(wait=True)
        with self._lock:
            for h in self._handles:
                h.close()
            self._handles = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# Large batches through the drop-in entry points.  Measured through the Python entry points (tools/entry_rate.py,
# profiles/r06_python_entry_points.txt; headline pairs, one MI355X):
#   resident tensors (estimate_batch_torch):   ONE call is always the fastest - 145 k pairs/s at 8192 and 16384 pairs (the LM tails amortise
#       inside a large call).
#   pageable host buffers (estimate_*_batch):  rounds 4-5 cut batches beyond 6144 pairs into 1024-pair chunks, two in flight, to hide the H2D
#       copies (110-115 k pairs/s at 8192-16384 against 109 k for one call then).  Since round 6 a host-buffer call copies in 256-pair slices on
#       its own copy stream beside the first kernels of the slices before (mdrp_capi.hip run_pass): ONE call is ahead there as well - 124.8 k
#       against 110.4 k at 8192 pairs, 120.9 k against 114.7 k at 16384.  The automatic chunking is therefore OFF by default
#       (PIPELINE_MIN = 0: never); MDRP_PIPELINE_MIN=<pairs> switches it on for batches beyond that size, BatchPipeline stays for callers that
#       have several independent batches to keep in flight.
# Pairs are independent units and every summation order depends on the record index and list position only, so chunked results are those of
# sequential chunk calls bit for bit (tests/test_gpu_boundary.py::test_batch_pipeline_equals_sequential_calls, tests/test_gpu_headline.py).
def _env_int(name, dflt):
    import os
    try:
        return int(os.environ.get(name, dflt))
    except ValueError:
        return dflt


PIPELINE_MIN = _env_int("MDRP_PIPELINE_MIN", 0)         # host batches beyond this size are chunked; 0 = never (the default)
PIPELINE_CHUNK = 1024                                    # pairs per chunk
PIPELINE_DEPTH = 2                                       # chunks in flight


def chunk_bounds(batch, chunk=None):
    """[lo, hi) of the chunks a batch of `batch` pairs is cut into: `chunk` pairs each, a short remainder (< chunk / 4) joins the last one"""
    chunk = PIPELINE_CHUNK if chunk is None else chunk
    if PIPELINE_MIN <= 0 or chunk <= 0 or batch <= max(PIPELINE_MIN, chunk):
        return [(0, batch)]
    cuts = list(range(0, batch, chunk)) + [batch]
    if len(cuts) > 2 and cuts[-1] - cuts[-2] < chunk // 4:
        del cuts[-2]
    return list(zip(cuts[:-1], cuts[1:]))


_auto_pipes = {}            # device -> BatchPipeline, shared by every calling thread (submissions go through the pipeline's own pool: thread-safe)
_auto_lock = threading.Lock()


def _auto_pipe(device):
    """ONE pipeline per device for the whole process (one per calling thread leaked two handles with full scratch and two worker
    threads for every short-lived thread of a server), closed at interpreter exit."""
    with _auto_lock:
        p = _auto_pipes.get(device)
        if p is None:
            if not _auto_pipes:
                import atexit
                atexit.register(close_auto_pipelines)
            p = _auto_pipes[device] = BatchPipeline(depth=PIPELINE_DEPTH, device=device)
        return p


def close_auto_pipelines():
    """release the handles (device scratch) and worker threads of the automatic pipelines; they are recreated on the next large batch"""
    with _auto_lock:
        pipes = list(_auto_pipes.values())
        _auto_pipes.clear()
    for p in pipes:
        p.close()


def estimate_host(kind, x1, x2, d1, d2, ro, bo, n_per_pair=None, cam1=None, cam2=None, device=0, want_mask=True):
    """Handle.estimate_batch for a batch of any size (host buffers): one call up to PIPELINE_MIN pairs, pipelined chunks beyond.
    Returns (records, mask) in pair order."""
    B = len(x1)
    bounds = chunk_bounds(B)
    if len(bounds) == 1:
        return _capi.default_handle(device).estimate_batch(kind, x1, x2, d1, d2, ro, bo, n_per_pair, cam1, cam2, want_mask)
    pipe = _auto_pipe(device)

    def cut(a, lo, hi):
        return None if a is None else a[lo:hi]
    futs = [pipe.submit(kind, x1[lo:hi], x2[lo:hi], cut(d1, lo, hi), cut(d2, lo, hi), ro, bo, cut(n_per_pair, lo, hi), cut(cam1, lo, hi), cut(cam2, lo, hi), want_mask)
            for lo, hi in bounds]
    parts = [f.result() for f in futs]
    res = np.concatenate([p[0] for p in parts])
    mask = np.concatenate([p[1] for p in parts]) if want_mask else None
    return res, mask


def estimate_device(kind, x1_ptr, x2_ptr, d1_ptr, d2_ptr, batch, n_max, ro, bo, n_per_pair=None, cam1=None, cam2=None, mask_ptr=None, device=0):
    """The same on device pointers ([batch][n_max][2] / [batch][n_max] float64, mask [batch][n_max] bytes): chunks are pointer offsets into the
    caller's buffers.  Only called for batches beyond PIPELINE_MIN; the caller has made sure the inputs are complete (stream synchronised).
    Returns the records (numpy) in pair order."""
    pipe = _auto_pipe(device)

    def cut(a, lo, hi):
        return None if a is None else a[lo:hi]
    futs = []
    for lo, hi in chunk_bounds(batch):
        futs.append(pipe.submit_device(kind, x1_ptr + 16 * n_max * lo, x2_ptr + 16 * n_max * lo, d1_ptr + 8 * n_max * lo if d1_ptr else 0,
                                       d2_ptr + 8 * n_max * lo if d2_ptr else 0, hi - lo, n_max, ro, bo, cut(n_per_pair, lo, hi), cut(cam1, lo, hi), cut(cam2, lo, hi),
                                       mask_ptr + n_max * lo if mask_ptr else None))
    return np.concatenate([f.result() for f in futs])
