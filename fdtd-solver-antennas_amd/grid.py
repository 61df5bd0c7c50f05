"""Rectilinear Yee grid: node lines plus the primal/dual metric the EC operator needs.

Restates the mesh conventions of the external engine the reference drives ([EXT] openEMS
Operator::GetDiscDelta): N node lines per axis; the primal delta of the last line repeats the
last cell, the dual delta of an interior line is half the distance between its neighbours and
the dual delta of a boundary line is the full adjacent cell."""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np


def _primal(lines: np.ndarray) -> np.ndarray:
    d = np.empty_like(lines)
    d[:-1] = np.diff(lines)
    d[-1] = d[-2]
    return d


def _dual(lines: np.ndarray) -> np.ndarray:
    d = np.empty_like(lines)
    d[1:-1] = 0.5 * (lines[2:] - lines[:-2])
    d[0] = lines[1] - lines[0]
    d[-1] = lines[-1] - lines[-2]
    return d


@dataclass
class RectGrid:
    """Node lines in metres, strictly increasing."""
    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    lines: tuple = field(init=False)
    d: tuple = field(init=False)     # primal edge lengths per axis, len n (last repeats)
    dd: tuple = field(init=False)    # dual edge lengths per axis, len n

    def __post_init__(self):
        ls = []
        for name in ("x", "y", "z"):
            a = np.ascontiguousarray(getattr(self, name), dtype=np.float64)
            if a.ndim != 1 or a.size < 2 or not np.all(np.diff(a) > 0):
                raise ValueError(f"{name} lines must be 1-D, >= 2 and strictly increasing")
            setattr(self, name, a)
            ls.append(a)
        self.lines = tuple(ls)
        self.d = tuple(_primal(a) for a in ls)
        self.dd = tuple(_dual(a) for a in ls)

    @property
    def shape(self):
        """(nx, ny, nz) node counts."""
        return (self.x.size, self.y.size, self.z.size)

    @property
    def ncells(self) -> int:
        """Cell count as the Mcells/s metric counts it (product of node-line counts)."""
        nx, ny, nz = self.shape
        return nx * ny * nz

    def centers(self, axis: int) -> np.ndarray:
        l = self.lines[axis]
        return 0.5 * (l[:-1] + l[1:])

    def snap(self, axis: int, coord: float) -> int:
        """Index of the node line nearest to coord."""
        l = self.lines[axis]
        return int(np.argmin(np.abs(l - coord)))

    def flat(self, i, j, k):
        nx, ny, _ = self.shape
        return (np.asarray(k, dtype=np.int64) * ny + np.asarray(j, dtype=np.int64)) * nx + np.asarray(i, dtype=np.int64)

    def courant_dt(self, safety: float = 0.99) -> float:
        """Rigorous bound: the 1-D difference operators are bounded by 2/min spacing per axis."""
        from .constants import C0
        s = sum(1.0 / float(np.min(np.diff(l))) ** 2 for l in self.lines)
        return safety / (C0 * np.sqrt(s))
