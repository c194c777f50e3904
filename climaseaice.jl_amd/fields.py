"""Field: an Oceananigans-style staggered field whose parent array lives in GPU memory.

Storage is a torch tensor (device memory plumbing only) of shape (nj, ni), C order, i.e. the
same bytes as Oceananigans' column-major (ni, nj) parent: element (i, j) (1-based) sits at
flat offset (i + Hx - 1) + (j + Hy - 1) * ni.
"""
import numpy as np
import torch

from .grids import Center, Face


class Field:
    def __init__(self, loc, grid, device=None, name=""):
        self.LX, self.LY = loc[0], loc[1]
        self.grid = grid
        self.name = name
        ni, nj = grid.field_size(self.LX, self.LY)
        self.ni, self.nj = ni, nj
        self.data = torch.zeros((nj, ni), dtype=torch.float64, device=device)

    @property
    def location(self):
        return (self.LX, self.LY)

    def parent(self):
        return self.data

    def interior(self):
        g = self.grid
        nx, ny = g.interior_size(self.LX, self.LY)
        return self.data[g.Hy:g.Hy + ny, g.Hx:g.Hx + nx]

    def numpy(self):
        return self.data.detach().cpu().numpy()

    def interior_numpy(self):
        return self.interior().detach().cpu().numpy()

    def set(self, value):
        """set!(field, value): a number, an (ny, nx) array, or a function f(x, y) of the field's nodes."""
        g = self.grid
        dst = self.interior()
        if callable(value):
            x = g.xnodes(self.LX)
            y = g.ynodes(self.LY)
            arr = np.asarray(value(x[None, :], y[:, None]), dtype=np.float64)
            arr = np.broadcast_to(arr, (len(y), len(x)))
            dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(dst.device))
        elif np.isscalar(value):
            dst.fill_(float(value))
        else:
            arr = np.asarray(value, dtype=np.float64)
            dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(dst.device))
        # torch wrote on ITS stream, the library works on its own (non-blocking) one: without this a library call issued right after
        # a set could read the field before the copy had landed (seen once in 30-odd runs of tests/test_gpu_steps.py under load)
        if dst.is_cuda:
            torch.cuda.current_stream(dst.device).synchronize()
        return self

    def fill_parent(self, value):
        self.data.fill_(float(value))
        if self.data.is_cuda:
            torch.cuda.current_stream(self.data.device).synchronize()
        return self


def CenterField(grid, device=None, name=""):
    return Field((Center, Center), grid, device, name)


def XFaceField(grid, device=None, name=""):
    return Field((Face, Center), grid, device, name)


def YFaceField(grid, device=None, name=""):
    return Field((Center, Face), grid, device, name)


def CornerField(grid, device=None, name=""):
    return Field((Face, Face), grid, device, name)
