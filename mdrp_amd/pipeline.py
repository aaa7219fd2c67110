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
