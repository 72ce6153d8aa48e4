"""Multi-GPU: the trial-source list shards across ranks (one process per GPU); replicated read-only
state (GF tensor, receivers, references) needs no exchange.  The only collective is one all-gather of
per-source misfit scalars (RCCL over xGMI on GPUs, gloo in CPU tests): N_s floats in total, i.e.
latency bound.  Synthetics are never exchanged.

Trial ordering is the reference's grid ordering (first parameter slowest, source.py:119-164), cut into
contiguous ranges, so that concatenating the ranks' results restores `misfits_by_src` order
(seismosizer.py:709-713)."""
import numpy as np


def shard_range(nsrc, world, rank):
    """Contiguous [lo, hi) of rank's share; sizes differ by at most one."""
    base, rem = divmod(nsrc, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_misfits(local, dist=None, device_index=None, counts=None, force=False):
    """All-gather per-source values (1-D or [n, k]) from every rank, restoring global source order.
    `dist` is torch.distributed (initialised) or None for a single process.  `counts`: the ranks' share sizes when the
    caller knows them (shard_range), which saves the exchange of the counts -- one collective per call instead of two."""
    local = np.ascontiguousarray(local, np.float32)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return local                                  # force: run the collective even for one rank (self-test)
    import torch
    world = dist.get_world_size()
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", device_index if device_index is not None else torch.cuda.current_device()) \
        if use_cuda else torch.device("cpu")
    # shard sizes may differ by one: exchange counts, pad to the maximum
    if counts is None:
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        counts = [int(c.item()) for c in counts]
    elif len(counts) != world or counts[dist.get_rank()] != local.shape[0]:
        raise ValueError("counts do not match this rank's share")
    nmax = max(counts)
    width = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
    buf = torch.zeros((nmax, width), dtype=torch.float32, device=dev)
    buf[:local.shape[0]] = torch.from_numpy(local.reshape(local.shape[0], width)).to(dev)
    out = torch.empty((world * nmax, width), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, nmax, width)
    parts = [out[r, :counts[r]] for r in range(world)]
    res = np.concatenate(parts, 0)
    return res.reshape((-1,) + local.shape[1:])


class DeviceGather:
    """The sharded grid search's one collective, device to device: every rank's global misfits go from the engine's own device
    buffer (Engine.global_misfits_device) into one all-gather over RCCL; the gathered array stays on the device until somebody
    asks for it (`host()`).  No staging through the host per step (until round 3: device -> numpy -> device -> gather -> host).
    Shares may differ by one source: shards are padded to the largest."""

    def __init__(self, dist, device_index, counts):
        import torch
        self.torch, self.dist, self.counts = torch, dist, list(counts)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if len(self.counts) != self.world:
            raise ValueError("counts do not match the world size")
        self.dev = torch.device("cuda", device_index)
        self.nmax = max(self.counts)
        self.buf = torch.zeros(self.nmax, dtype=torch.float32, device=self.dev)
        self.out = torch.empty(self.world * self.nmax, dtype=torch.float32, device=self.dev)

    def gather(self, engine, isrc0=0):
        n = self.counts[self.rank]
        if n:
            src = self.torch.as_tensor(engine.global_misfits_device(isrc0, n), device=self.dev)   # (the call synchronises the engine's stream)
            self.buf[:n].copy_(src)
        self.dist.all_gather_into_tensor(self.out, self.buf)
        return self.out

    def host(self):
        o = self.out.cpu().numpy().reshape(self.world, self.nmax)
        return np.concatenate([o[r, :self.counts[r]] for r in range(self.world)])


def best_source(global_misfits):
    """MisfitGrid's argmin over the trial list (gridsearch.py:250-266): NaNs ignored."""
    g = np.asarray(global_misfits, np.float64)
    return int(np.nanargmin(g))


def sharded_misfits_for_sources(engine, sourcetype, params, dist=None, device_index=None):
    """`Seismosizer.make_misfits_for_sources` (seismosizer.py:682-722) over all ranks: this rank evaluates its
    contiguous share of `params` on its own GPU, then the [n, N_r, N_k] misfit and norm arrays are
    all-gathered so that every rank holds the full, ordered result (what MisfitGrid's outer norm and bootstrap
    need: N_s * N_r * N_k * 2 floats).  Returns (misfits_by_src, norms_by_src, failings) with
    `failings` in global source indices (seismosizer.py:716-717)."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_range(len(params), world, rank)
    if hi > lo:
        m, n, fails = engine.make_misfits_for_sources(sourcetype, params[lo:hi])
    else:
        nrec = len(engine.components)
        nk = max([len(c) for c in engine.components] + [1])
        m = n = np.zeros((0, nrec, nk))
        fails = []
    shape = m.shape[1:]
    failed = np.zeros((len(m), 1))
    failed[list(fails)] = 1.0
    both = np.concatenate([m.reshape(len(m), -1), n.reshape(len(n), -1), failed], 1)
    allb = gather_misfits(both, dist, device_index).astype(np.float64)
    k = (allb.shape[1] - 1) // 2
    failings = [int(i) for i in np.nonzero(allb[:, 2 * k])[0]]
    return allb[:, :k].reshape((-1,) + shape), allb[:, k:2 * k].reshape((-1,) + shape), failings
