"""Rectilinear mesh generation: edge hints ("thirds rule") and graded smoothing.

Restates what the reference asks of the external toolkit through
``FDTD.AddEdges2Grid(dirs, properties, metal_edge_res)`` and ``mesh.SmoothMeshLines('all', res, 1.4)``
(antenna_sim/solver_fdtd_openems_fixed.py:193,210,217): [EXT] openEMS automesh / CSXCAD
SmoothMeshLines.  Their source is not available here, so line positions are NOT pinned against
them; the invariants are (tests/test_host_logic_cpu.py::test_mesher_*): every hint line is kept (an isolated pair closer than max_res / 100
becomes one line at its middle: merge_close_lines), no cell exceeds max_res, neighbouring cells differ by at most `ratio` wherever
the hints allow it.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence
import numpy as np


def unique_lines(lines: Iterable[float], rel_tol: float = 1e-9) -> np.ndarray:
    a = np.sort(np.asarray(list(lines), dtype=np.float64))
    if a.size == 0:
        return a
    span = max(a[-1] - a[0], 1e-30)
    keep = [a[0]]
    for v in a[1:]:
        if v - keep[-1] > rel_tol * span:
            keep.append(v)
    return np.array(keep)


def mesh_hint_from_box(start: Sequence[float], stop: Sequence[float], dirs: Sequence[int],
                       metal_edge_res: Optional[float] = None) -> List[Optional[List[float]]]:
    """Hint lines for a box.  With metal_edge_res the metal edge is put 1/3 of a cell inside the
    metal (lines at edge + res/3 inside and edge - 2 res/3 outside): the "thirds rule" that places
    the singular edge field correctly on a Yee grid.  Degenerate extents give a single line."""
    lo = np.minimum(start, stop).astype(float)
    hi = np.maximum(start, stop).astype(float)
    hints: List[Optional[List[float]]] = [None, None, None]
    for a in range(3):
        if a not in dirs:
            continue
        ext = hi[a] - lo[a]
        if metal_edge_res is not None and ext > metal_edge_res:
            r = float(metal_edge_res)
            hints[a] = [lo[a] + r / 3.0, lo[a] - 2.0 * r / 3.0, hi[a] - r / 3.0, hi[a] + 2.0 * r / 3.0]
        elif ext > 0:
            hints[a] = [lo[a], hi[a]]
        else:
            hints[a] = [lo[a]]
    return hints


def _graded_fill(a: float, b: float, left: float, right: float, max_res: float, ratio: float) -> np.ndarray:
    """Interior lines for the gap (a, b): cells grow by `ratio` from the neighbouring cell sizes
    `left`/`right` towards max_res and are then scaled down uniformly to fit the gap exactly."""
    gap = b - a
    sl = min(max(left, 1e-30) * ratio, max_res)
    sr = min(max(right, 1e-30) * ratio, max_res)
    if gap <= min(max_res, max(sl, sr)) * (1 + 1e-9):
        return np.empty(0)
    ls, rs, tot = [], [], 0.0
    while tot < gap * (1 - 1e-12):
        if sl <= sr:
            ls.append(sl); tot += sl; sl = min(sl * ratio, max_res)
        else:
            rs.append(sr); tot += sr; sr = min(sr * ratio, max_res)
    cells = np.array(ls + rs[::-1])
    if cells.size < 2:
        return np.empty(0)
    cells *= gap / cells.sum()
    return a + np.cumsum(cells)[:-1]


def merge_close_lines(lines: np.ndarray, min_gap: float, lone: float = 0.125, record: Optional[list] = None) -> np.ndarray:
    """A pair of hint lines closer than min_gap AND closer than `lone` x the gaps on either side of it becomes ONE line at its middle (the
    outermost two lines stay where they are).

    Independent hint sets — a port's [start, centre, stop], the nine bounding-box lines the multi-patch scene adds per metal
    sheet, edge thirds — can land micrometres apart by accident: the reference's 2 x 2 array at a 61.2 mm pitch puts two y lines
    6.8 um apart (their neighbours 78 and 106 um away) on a 3.4 mm mesh.  The toolkit the reference calls keeps such a pair and pays with
    the timestep (Courant on the smallest cell: 2.3e-14 s, the excitation pulse alone then needs 104 000 timesteps, more than the scene's
    NrTS of 92 758 — the run ends before the pulse does); here the pair is one line, moved by 3.4 um.  Lines that are close ON PURPOSE
    — the five lines across a 0.254 mm substrate under a 900 MHz mesh are 63 um apart, max_res / 100 there is 110 um — have equally close
    neighbours and stay."""
    a = np.asarray(lines, dtype=np.float64)
    if min_gap <= 0:
        return a
    while a.size >= 4:
        d = np.diff(a)
        left = np.concatenate([[np.inf], d[:-1]])
        right = np.concatenate([d[1:], [np.inf]])
        cand = np.nonzero((d < min_gap) & (d < lone * np.minimum(left, right)))[0]
        if cand.size == 0:
            break
        i = int(cand[np.argmin(d[cand])])
        if record is not None:
            record.append((float(a[i]), float(a[i + 1])))
        mid = 0.5 * (a[i] + a[i + 1])
        if i == 0:
            mid = a[0]
        elif i + 1 == a.size - 1:
            mid = a[-1]
        a = np.concatenate([a[:i], [mid], a[i + 2:]])
    return a


# isolated pairs of lines closer than max_res / MERGE_FRACTION are merged before the fill-in (a 3.4 mm mesh: 34 um)
MERGE_FRACTION = 100.0


def smooth_mesh_lines(lines: Iterable[float], max_res: float, ratio: float = 1.5, merge_fraction: Optional[float] = None,
                      merged: Optional[list] = None) -> np.ndarray:
    """The hint lines (isolated pairs closer than max_res / merge_fraction merged: merge_close_lines), plus graded fill-in so that no cell
    is larger than max_res.  merge_fraction: None = MERGE_FRACTION (100; $FDTD_MESH_MERGE_FRACTION overrides), 0 = keep EVERY hint line, as
    the toolkit the reference calls does (a deliberate deviation of this package, DESIGN.md: out of scope / deviations).  `merged`: a list
    that receives the (line, line) pairs that became one line, so that a caller can tell its user that the mesh differs."""
    out = unique_lines(lines)
    if out.size < 2:
        return out
    max_res = float(max_res)
    if merge_fraction is None:
        import os
        merge_fraction = float(os.environ.get("FDTD_MESH_MERGE_FRACTION", MERGE_FRACTION))
    if merge_fraction > 0:
        out = merge_close_lines(out, max_res / merge_fraction, record=merged)
    for _ in range(10 * out.size + 1000):
        d = np.diff(out)
        big = np.nonzero(d > max_res * (1 + 1e-9))[0]
        if big.size == 0:
            break
        i = int(big[np.argmax(d[big])])
        left = d[i - 1] if i > 0 else max_res
        right = d[i + 1] if i + 1 < d.size else max_res
        new = _graded_fill(out[i], out[i + 1], min(left, max_res), min(right, max_res), max_res, ratio)
        if new.size == 0:   # cannot grade: split evenly
            n = int(np.ceil(d[i] / max_res))
            new = out[i] + d[i] * np.arange(1, n) / n
        out = np.concatenate([out[:i + 1], new, out[i + 1:]])
    return out
