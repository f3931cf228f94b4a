"""Multi-GPU evaluation: the live-point batch is sharded row-wise across ranks (one
process per GPU), every rank evaluates its shard with no data-path collective, and the
shards' logL are exchanged with ONE all-gather per batch (RCCL over xGMI when the
backend is ``nccl``; ``gloo`` on CPU hosts for tests).

The reference's analogue is the schwimmbad MPIPool task farm
(``nmma/core/mpi_setup.py:651-667``): N ranks x 1 point becomes G ranks x B/G points.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced [lo, hi) of ``n`` rows for ``rank`` (first n % world ranks get +1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedEvaluator:
    """``evaluate(theta_full)`` -> logL for ALL rows on every rank.

    ``local_fn(theta_shard) -> logL_shard`` is the per-rank evaluation (e.g.
    ``likelihood.log_likelihood_batch``); tensors stay on the rank's device.
    """

    def __init__(self, local_fn, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.local_fn = local_fn
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def evaluate(self, theta):
        import torch
        n = theta.shape[0]
        lo, hi = shard_bounds(n, self.world, self.rank)
        local = self.local_fn(theta[lo:hi])
        if not isinstance(local, torch.Tensor):
            local = torch.as_tensor(np.asarray(local))
        if self.world == 1:
            return local
        # the collective runs where its backend works: RCCL ("nccl") on device buffers, gloo on host copies
        home = local.device
        backend = self.dist.get_backend(self.group)
        if backend == "gloo" and local.is_cuda:
            local = local.cpu()
        elif backend == "nccl" and not local.is_cuda:
            local = local.to(f"cuda:{torch.cuda.current_device()}")
        # equal-sized slots so a single all_gather_into_tensor suffices; pad the short shards
        slot = (n + self.world - 1) // self.world
        buf = torch.full((slot,), float("nan"), dtype=local.dtype, device=local.device)
        buf[: hi - lo] = local
        out = torch.empty(self.world * slot, dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(out, buf, group=self.group)
        out = out.to(home)
        pieces = []
        for r in range(self.world):
            a, b = shard_bounds(n, self.world, r)
            pieces.append(out[r * slot: r * slot + (b - a)])
        return torch.cat(pieces)


def pack_records(u, v, logl, counts):
    """``(u[n, D], v[n, D], logl[n], counts[n, 4] int32)`` -> ``rows[n, 2 D + 3]`` float64 in the layout of
    ``nmma_walk_queue::records_dev`` (include/nmma_hip.h): ``u | v | logl | counts``, the four counters in the bit patterns of two
    doubles -- copied, never converted, so a collective that moves bytes returns them exactly."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    n, ndim = u.shape
    rows = np.empty((n, 2 * ndim + 3))
    rows[:, :ndim], rows[:, ndim:2 * ndim], rows[:, 2 * ndim] = u, v, logl
    rows[:, 2 * ndim + 1:] = np.ascontiguousarray(counts, dtype=np.int32).reshape(n, 4).view(np.float64)
    return rows


def unpack_records(rows, ndim):
    """Inverse of :func:`pack_records` for host rows: ``(u, v, logl, counts[n, 4] int32)``."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    n = rows.shape[0]
    counts = np.ascontiguousarray(rows[:, 2 * ndim + 1:2 * ndim + 3]).view(np.int32).reshape(n, 4)
    return (np.ascontiguousarray(rows[:, :ndim]), np.ascontiguousarray(rows[:, ndim:2 * ndim]), np.ascontiguousarray(rows[:, 2 * ndim]), counts)


class ShardedQueue:
    """One queue of the nested sampler over ``world`` ranks (one process per GPU): ``run(live, u0, loglstar, keys, walks)`` returns
    ``(u, v, logl, counts)`` for ALL chains on every rank.  Rank r walks the chains ``shard_bounds(n, world, r)`` and the shards'
    records are exchanged with ONE all-gather of a packed ``[slot, 2 D + 3]`` float64 buffer per rank (:func:`pack_records`).  The
    reference spreads a queue's chains over its MPI ranks the same way (``nmma/core/mpi_setup.py:651-667, :679-683``); a chain's
    path depends on its key only (counter-based random numbers), so the records equal the single-rank queue's, bit for bit.

    Two forms of the rank's walk:

    * ``engine=`` an :class:`nmma_amd.engine.EMEngine` (+ ``table`` = ``sampler.device_prior_table(...)``, ``constraints`` = the
      likelihood's lowered Constraint program or None) -- the product form, DEVICE-RESIDENT: the library packs the shard's records
      into the send buffer on the GPU (``nmma_walk_queue::records_dev``; no download of the shard), the all-gather runs on the same
      stream (RCCL over xGMI with backend ``nccl``), and ONE download brings all records to the host.  With a ``gloo`` group (ranks
      sharing one GPU in the tests) the send buffer is staged through the host for the collective only.
    * ``local_fn(live, u0_shard, loglstar_shard, keys_shard, walks_or_shard) -> (u, v, logl, counts)`` -- any other walk (the CPU
      tests' stand-ins; the host walk of a likelihood without an engine form); its records are packed on the host.
    """

    def __init__(self, local_fn=None, group=None, device=None, engine=None, table=None, constraints=None):
        import torch.distributed as dist
        if (local_fn is None) == (engine is None):
            raise ValueError("ShardedQueue takes either local_fn or engine")
        self.dist = dist
        self.local_fn = local_fn
        self.engine, self.table, self.constraints = engine, table, constraints
        self.group = group
        self.device = device if device is not None else (f"cuda:{engine.device}" if engine is not None else None)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._bufs = {}               # (slot, width) -> (send, gathered) device buffers, reused from queue to queue
        self.last_gpu_ms = 0.0        # the rank's own walk, HIP events (engine form)
        self.last_collect_ms = 0.0    # host wall time from `begin` returning to the records on the host: rest of the walk + all-gather + download

    def _backend(self):
        return self.dist.get_backend(self.group) if self.dist.is_initialized() else None

    def run(self, live, u0, loglstar, keys, walks, table=None):
        import torch
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        n, ndim = u0.shape
        star = np.ascontiguousarray(np.broadcast_to(np.asarray(loglstar, dtype=np.float64), (n,)))
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        lo, hi = shard_bounds(n, self.world, self.rank)
        wl = walks if np.ndim(walks) == 0 else np.ascontiguousarray(np.asarray(walks)[lo:hi])
        slot, width = (n + self.world - 1) // self.world, 2 * ndim + 3
        rows_of = np.concatenate([np.arange(r * slot, r * slot + (b - a)) for r, (a, b) in
                                  enumerate(shard_bounds(n, self.world, r) for r in range(self.world))]) if n else np.zeros(0, dtype=int)
        if self.engine is not None:
            import time
            eng = self.engine
            tab = table if table is not None else self.table
            if tab is None:
                raise ValueError("ShardedQueue(engine=...) needs the device prior table (table=)")
            if (slot, width) not in self._bufs:
                self._bufs.clear()
                self._bufs[(slot, width)] = (torch.zeros((max(slot, 1), width), dtype=torch.float64, device=self.device),
                                             torch.empty((self.world * max(slot, 1), width), dtype=torch.float64, device=self.device))
            send, gathered = self._bufs[(slot, width)]
            token = None
            with torch.cuda.device(eng.device):       # (walk and collective on this device's current stream, in that order)
                if hi > lo:
                    token = eng.walk_queue_begin(tab, live, u0[lo:hi], star[lo:hi], keys[lo:hi], wl, constraints=self.constraints, records=send)
                try:
                    t0 = time.perf_counter()
                    if not self.dist.is_initialized():    # (no process group at all; a group of ONE rank still runs its collective)
                        full = send.cpu().numpy()
                    elif self._backend() == "gloo":       # (test mode: ranks share a GPU, the collective runs on host copies)
                        out = torch.empty((self.world * max(slot, 1), width), dtype=torch.float64)
                        self.dist.all_gather_into_tensor(out, send.cpu(), group=self.group)
                        full = out.numpy()
                    else:                                 # RCCL: same stream as the walk, no host round trip before the one download
                        self.dist.all_gather_into_tensor(gathered, send, group=self.group)
                        full = gathered.cpu().numpy()
                    self.last_collect_ms = 1e3 * (time.perf_counter() - t0)
                finally:
                    if token is not None:
                        eng.walk_queue_end(token)         # (the stream is idle by now: waits for nothing, checks the handle)
                        self.last_gpu_ms = eng.last_walk_gpu_ms
            return unpack_records(full[rows_of], ndim)
        if hi > lo:
            u, v, logl, counts = self.local_fn(live, u0[lo:hi], star[lo:hi], keys[lo:hi], wl)
        else:
            u, v, logl, counts = np.empty((0, ndim)), np.empty((0, ndim)), np.empty(0), np.empty((0, 4), dtype=np.int32)
        if self.world == 1:
            return u, v, logl, counts
        mine = np.zeros((slot, width))
        mine[: hi - lo] = pack_records(u, v, logl, counts)
        dev = self.device if self.device is not None else "cpu"
        buf = torch.as_tensor(mine).to(dev)
        out = torch.empty((self.world * slot, width), dtype=torch.float64, device=dev)
        self.dist.all_gather_into_tensor(out, buf, group=self.group)
        return unpack_records(out.cpu().numpy()[rows_of], ndim)


def send_command(dist, group, header, arrays, device=None):
    """Rank 0 of ``group`` hands a command to the others (the master / worker form of ``GPUPool``): ``header`` (a small picklable dict) and
    ``arrays`` (name -> numpy array) travel as ONE ``broadcast_object_list`` of the header + the arrays' shapes and ONE broadcast of their
    bytes (RCCL with backend ``nccl``: from ``device``; gloo: host tensors).  Returns what :func:`recv_command` returns on the workers."""
    import torch
    arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    meta = [(k, v.shape, v.dtype.str, v.nbytes) for k, v in arrays.items()]
    dist.broadcast_object_list([dict(header, _arrays=meta)], src=0, group=group)
    total = sum(m[3] for m in meta)
    if total:
        blob = np.concatenate([v.reshape(-1).view(np.uint8) for v in arrays.values()]) if len(arrays) > 1 else next(iter(arrays.values())).reshape(-1).view(np.uint8)
        t = torch.from_numpy(np.ascontiguousarray(blob))
        if dist.get_backend(group) == "nccl":
            t = t.to(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        dist.broadcast(t, src=0, group=group)
    return header, arrays


def recv_command(dist, group, device=None):
    """The workers' side of :func:`send_command`: ``(header, {name: numpy array})``."""
    import torch
    box = [None]
    dist.broadcast_object_list(box, src=0, group=group)
    header = dict(box[0])
    meta = header.pop("_arrays")
    total = sum(m[3] for m in meta)
    arrays = {}
    if total:
        on_dev = dist.get_backend(group) == "nccl"
        t = torch.empty(total, dtype=torch.uint8, device=(device if device is not None else f"cuda:{torch.cuda.current_device()}") if on_dev else "cpu")
        dist.broadcast(t, src=0, group=group)
        blob = t.cpu().numpy()
        off = 0
        for name, shape, dtype, nbytes in meta:
            arrays[name] = blob[off:off + nbytes].view(np.dtype(dtype)).reshape(shape).copy()
            off += nbytes
    else:
        for name, shape, dtype, _ in meta:
            arrays[name] = np.empty(shape, dtype=np.dtype(dtype))
    return header, arrays


class MultiDeviceEvaluator:
    """ONE process driving several GPUs ("single process x 8 devices (ctypes + streams) avoids MPI").

    ``engine_factory(device) -> engine`` builds one engine per entry of ``devices`` (an ``EMEngine``, a ``GWEngine`` -- anything
    with ``loglike(theta, out=..., stream=...)`` or a plain callable ``theta_shard -> logL_shard`` on that device).  Every
    ``evaluate`` splits the rows with :func:`shard_bounds`, copies each shard to its device (peer copy over xGMI when theta
    lives on a GPU), launches all shards asynchronously on per-device streams -- the C ABI only enqueues, so one host thread
    keeps every device busy -- and gathers the pieces on ``devices[0]``.  No collective library is involved."""

    def __init__(self, engine_factory, devices, engines=None):
        """``engines``: ready-made engines, one per entry of ``devices`` (they stay the caller's: ``close`` leaves them alone)."""
        import torch
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("MultiDeviceEvaluator needs at least one device")
        self._own = engines is None
        self.engines = [engine_factory(d) for d in self.devices] if engines is None else list(engines)
        if len(self.engines) != len(self.devices):
            raise ValueError("one engine per device")
        self.streams = [torch.cuda.Stream(device=d) for d in self.devices]

    def evaluate(self, theta):
        import torch
        if not isinstance(theta, torch.Tensor):
            theta = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64))
        n, world = theta.shape[0], len(self.devices)
        home = torch.device(f"cuda:{self.devices[0]}")
        out = torch.empty(n, dtype=torch.float64, device=home)
        src_stream = torch.cuda.current_stream(theta.device) if theta.is_cuda else None
        pieces = []
        for r, (d, eng, s) in enumerate(zip(self.devices, self.engines, self.streams)):
            lo, hi = shard_bounds(n, world, r)
            if hi == lo:
                continue
            with torch.cuda.device(d), torch.cuda.stream(s):
                if src_stream is not None:
                    s.wait_stream(src_stream)          # theta may still be in flight on its producer's stream
                shard = theta[lo:hi].to(f"cuda:{d}", non_blocking=True).contiguous()
                if hasattr(eng, "loglike"):
                    part = eng.loglike(shard, stream=s)
                elif hasattr(eng, "loglike_ratio"):
                    part = eng.loglike_ratio(shard, stream=s)
                else:
                    part = eng(shard)
                pieces.append((lo, hi, part, s))
        cur = torch.cuda.current_stream(home)
        for lo, hi, part, s in pieces:
            cur.wait_stream(s)
            out[lo:hi].copy_(part, non_blocking=True)
            part.record_stream(cur)        # allocated under `s`, read by the copy on `cur`: keep the block until that copy is done
        return out

    def close(self):
        if not self._own:
            return
        for eng in self.engines:
            if hasattr(eng, "close"):
                eng.close()
