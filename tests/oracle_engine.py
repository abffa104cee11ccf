"""Test-only compute engine with the interface of pfhubbenchmarks_amd.solver.HipSlabEngine, backed by the CPU
oracle (oracle/ch_fd.c) and CPU torch tensors, so the slab / halo-exchange logic of SlabSolver can run under the
gloo backend without a GPU.  Never imported by product code."""
import contextlib

import numpy as np
import torch

from oracle import ch_fd
from pfhubbenchmarks_amd.solver import slab_partition


class OracleSlabEngine:
    ghost = 2

    def __init__(self, n, h, nranks, rank, **params):
        nx, ny, nz = (n, n, n) if isinstance(n, int) else n
        self.nx, self.ny, self.nz_global = nx, ny, nz
        self.h = h
        self.z0, self.nz = slab_partition(nz, nranks, rank)
        self.buffers = [torch.zeros((self.nz + 4, ny, nx), dtype=torch.float64) for _ in range(2)]
        self._cur = 0
        self.rank_lo = (rank - 1) % nranks
        self.rank_hi = (rank + 1) % nranks
        self._open = None

    @property
    def cur(self):
        return self._cur

    def stream_context(self):
        return contextlib.nullcontext()

    def set_local(self, arr):
        self.buffers[self._cur][2:2 + self.nz] = torch.from_numpy(np.ascontiguousarray(arr))

    def get_local(self):
        return self.buffers[self._cur][2:2 + self.nz].numpy().copy()

    def _launch(self, dt, zlo, zhi):
        if zhi <= zlo:
            return
        ch_fd.fd_step(self.buffers[self._cur].numpy(), dt, h=self.h, ghost=2, zwrap=0, zlo=zlo, zhi=zhi,
                      out=self.buffers[1 - self._cur].numpy())

    def step_begin(self, dt):
        assert self._open is None
        self._launch(dt, 2, self.nz - 2)
        self._open = dt

    def step_finish(self):
        dt, nz = self._open, self.nz
        if nz - 2 > 2:
            self._launch(dt, 0, 2)
            self._launch(dt, nz - 2, nz)
        else:
            self._launch(dt, 0, nz)
        self._cur ^= 1
        self._open = None

    def diag_local(self):
        F, C, E = ch_fd.diagnostics(self.buffers[self._cur].numpy(), h=self.h, dim=3, ghost=2, zwrap=0)
        return [F, C, E]

    def sync(self):
        pass
