"""Running observation normaliser (mirror of uhc/khrylib/utils/zfilter.py:7-73).

``RunningStat.push`` is Welford's update for one vector (reference semantics).  ``push_batch`` merges a
whole batch of observations (Chan et al. parallel update) with torch on the device holding the batch --
mathematically the same statistics as pushing the rows one by one.  ``ZFilter.__call__`` accepts either
a single numpy vector (reference behaviour) or a (B, dim) torch tensor (batched env)."""
import numpy as np
import torch

from uhc_amd import rollout_ops


class RunningStat:
    def __init__(self, shape):
        self._n = 0
        self._M = np.zeros(shape)
        self._S = np.zeros(shape)
        # device-resident copy used by the batched path: [n (0-d), M, S].  Once created the three tensors are only ever updated IN PLACE
        # (a captured HIP graph of the rollout step keeps their addresses); _stale = the device holds newer statistics than the numpy
        # arrays, _dirty = the numpy arrays were set from the host and must be copied down before the next device use
        self._dev, self._stale, self._dirty = None, False, False
        self._scratch = {}  # rows per batch -> partial-sum buffer of the library's push (kept: a captured graph holds its address)

    def _host_changed(self):
        self._stale, self._dirty = False, self._dev is not None

    def push(self, x):
        self._sync()
        x = np.asarray(x)
        assert x.shape == self._M.shape
        self._n += 1
        if self._n == 1:
            self._M[...] = x
        else:
            old = self._M.copy()
            self._M[...] = old + (x - old) / self._n
            self._S[...] = self._S + (x - old) * (x - self._M)
        self._host_changed()

    def _to_device(self, device):
        if self._dev is not None and self._dev[1].device != device:
            self._sync()
            self._dev = None
        if self._dev is None:
            self._dev = [torch.tensor(float(self._n), dtype=torch.float64, device=device), torch.as_tensor(self._M, dtype=torch.float64, device=device).clone(),
                         torch.as_tensor(self._S, dtype=torch.float64, device=device).clone()]
            self._dirty = False
        elif self._dirty:
            self._dev[0].fill_(float(self._n))
            self._dev[1].copy_(torch.as_tensor(self._M, dtype=torch.float64))
            self._dev[2].copy_(torch.as_tensor(self._S, dtype=torch.float64))
            self._dirty = False

    def push_batch(self, xb: torch.Tensor, weights: torch.Tensor = None):
        """Merge B rows at once: (n, M, S) <- merge((n, M, S), (B, mean_b, S_b)).  `weights` (0/1 per row, a device tensor) selects the
        rows that count, without the host having to know how many there are.  The statistics -- the count included -- stay on the
        device of `xb` and are updated in place (no host sync per rollout step, graph-capturable); the numpy views are refreshed lazily."""
        if xb.shape[0] == 0:
            return
        self._to_device(xb.device)
        n, M, Sd = self._dev
        if xb.dim() == 2 and M.dim() == 1 and rollout_ops.usable(xb) and (weights is None or (weights.dtype == torch.int32 and weights.is_contiguous())):
            # device float64: two launches of the library (uhc_filter_push: per-row-block partials, fixed-order Chan merges)
            key = xb.shape[0]
            scratch = self._scratch.get(key)
            if scratch is None or scratch.device != xb.device:
                scratch = self._scratch[key] = rollout_ops.filter_scratch(xb.shape[0], xb.shape[1], xb.device)
            rollout_ops.filter_push(xb, weights, n, M, Sd, scratch)
            self._stale = True
            return
        x = xb.double()
        if weights is None:
            B = torch.full((), float(xb.shape[0]), dtype=torch.float64, device=xb.device)
            mb = x.mean(0)
            Sb = ((x - mb) ** 2).sum(0)
        else:
            w = (weights != 0).double()
            B = w.sum()
            mb = (w[:, None] * x).sum(0) / B.clamp(min=1.0)
            Sb = (w[:, None] * (x - mb) ** 2).sum(0)
        tot = n + B
        ts = tot.clamp(min=1.0)
        delta = mb - M
        Sd.add_(Sb + delta * delta * (n * B / ts))
        M.add_(delta * (B / ts))
        n.copy_(tot)
        self._stale = True

    def mark_device_updated(self):
        """The device copy was advanced by a replayed graph (no Python ran): the numpy views are stale."""
        if self._dev is not None:
            self._stale = True

    def _sync(self):
        if self._stale:
            self._n = int(round(float(self._dev[0].item())))
            self._M[...] = self._dev[1].cpu().numpy()
            self._S[...] = self._dev[2].cpu().numpy()
            self._stale = False

    def device_mean_std(self, like: torch.Tensor):
        """(mean, std) as tensors on like.device without a host round trip."""
        self._to_device(like.device)
        n, M, Sd = self._dev
        var = torch.where(n > 1, Sd / (n - 1).clamp(min=1.0), M * M)
        return M.to(like.dtype), torch.sqrt(var).to(like.dtype)

    def merge(self, nb, mb, Sb):
        self._sync()
        n = self._n
        tot = n + nb
        delta = mb - self._M
        self._S[...] = self._S + Sb + delta * delta * (n * nb / tot)
        self._M[...] = self._M + delta * (nb / tot)
        self._n = tot
        self._host_changed()

    def __getstate__(self):  # checkpoints hold plain numpy statistics, like the reference's pickles
        self._sync()
        return {"_n": self._n, "_M": self._M, "_S": self._S}

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._dev, self._stale, self._dirty = None, False, False
        self._scratch = {}

    @property
    def n(self):
        self._sync()
        return self._n

    @property
    def mean(self):
        self._sync()
        return self._M

    shape = property(lambda self: self._M.shape)

    @property
    def var(self):
        self._sync()
        return self._S / (self._n - 1) if self._n > 1 else np.square(self._M)

    @property
    def std(self):
        return np.sqrt(self.var)


class ZFilter:
    """y = clip((x - mean) / (std + 1e-8), +-clip) with running estimates of mean, std."""

    def __init__(self, shape, demean=True, destd=True, clip=10.0):
        self.demean, self.destd, self.clip = demean, destd, clip
        self.rs = RunningStat(shape)

    def __call__(self, x, update=True, out=None, step_counter=None):
        """out / step_counter (batched device path): write the result into `out` and add one to the int64 device scalar `step_counter`
        (the rollout's step index; with the library's kernel both ride on the normalisation launch)."""
        if torch.is_tensor(x):
            if update:
                self.rs.push_batch(x)
            if x.dim() == 2 and self.rs._M.ndim == 1 and rollout_ops.usable(x) and (out is None or rollout_ops.usable(out)):
                self.rs._to_device(x.device)
                n, M, Sd = self.rs._dev
                y = torch.empty_like(x) if out is None else out
                rollout_ops.filter_apply(x, n, M, Sd, self.demean, self.destd, self.clip, y, step_counter)
                return y
            y = self._call_torch(x)
            if out is not None:
                out.copy_(y)
                y = out
            if step_counter is not None:
                step_counter.add_(1)
            return y
        if update:
            self.rs.push(x)
        if self.demean:
            x = x - self.rs.mean
        if self.destd:
            x = x / (self.rs.std + 1e-8)
        return np.clip(x, -self.clip, self.clip) if self.clip else x

    def _call_torch(self, x):
        mean, std = self.rs.device_mean_std(x)
        if self.demean:
            x = x - mean
        if self.destd:
            x = x / (std + 1e-8)
        return torch.clamp(x, -self.clip, self.clip) if self.clip else x

    def set_mean_std(self, mean, std, n):
        self.rs._n = n
        self.rs._M[...] = mean
        self.rs._S[...] = std
        self.rs._host_changed()
