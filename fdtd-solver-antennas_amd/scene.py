"""Scene description + voxeliser: what the reference builds through the CSXCAD/openEMS Python API
(``CSX.AddMaterial/AddMetal(...).AddBox``, ``FDTD.AddLumpedPort``, ``FDTD.CreateNF2FFBox`` —
antenna_sim/solver_fdtd_openems_fixed.py:176-220) kept as plain data, and its mapping onto the
Yee grid (what [EXT] openEMS does when ``FDTD.Run`` sets up its operator):

  * material boxes -> per-cell eps_r / kappa (highest priority box containing the cell centre);
  * metal boxes (PEC, any thickness incl. zero) -> every edge with both end nodes inside is PEC;
  * lumped port  -> per-edge conductance, soft-source edges, voltage line and current loop.

Coordinates are in drawing units (``unit`` metres per unit, 1e-3 in every reference scene).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple
import numpy as np

from .grid import RectGrid
from .ecoperator import LumpedEdge


@dataclass
class Box:
    start: Tuple[float, float, float]
    stop: Tuple[float, float, float]
    priority: int = 0
    matrix: Optional[np.ndarray] = None   # 4x4 local->world (drawing units), None = identity


@dataclass
class Material:
    name: str
    eps_r: float = 1.0
    kappa: float = 0.0
    boxes: List[Box] = field(default_factory=list)

    def add_box(self, start, stop, priority=0):
        self.boxes.append(Box(tuple(map(float, start)), tuple(map(float, stop)), int(priority)))
        return self


@dataclass
class Metal:
    name: str
    boxes: List[Box] = field(default_factory=list)

    def add_box(self, start, stop, priority=0):
        self.boxes.append(Box(tuple(map(float, start)), tuple(map(float, stop)), int(priority)))
        return self


@dataclass
class LumpedPort:
    """AddLumpedPort(port_nr, R, start, stop, p_dir, excite, priority) as plain data."""
    number: int
    R: float
    start: Tuple[float, float, float]
    stop: Tuple[float, float, float]
    direction: int            # 0,1,2
    excite: float = 1.0
    priority: int = 5
    delay_steps: int = 0


@dataclass
class Scene:
    unit: float = 1e-3
    materials: List[Material] = field(default_factory=list)
    metals: List[Metal] = field(default_factory=list)
    ports: List[LumpedPort] = field(default_factory=list)

    def add_material(self, name, eps_r=1.0, kappa=0.0) -> Material:
        m = Material(name, float(eps_r), float(kappa))
        self.materials.append(m)
        return m

    def add_metal(self, name) -> Metal:
        m = Metal(name)
        self.metals.append(m)
        return m

    def add_lumped_port(self, number, R, start, stop, direction, excite=1.0, priority=5) -> LumpedPort:
        d = {"x": 0, "y": 1, "z": 2}.get(direction, direction)
        p = LumpedPort(int(number), float(R), tuple(map(float, start)), tuple(map(float, stop)), int(d),
                       float(excite), int(priority))
        self.ports.append(p)
        return p


@dataclass
class PortOnGrid:
    port: LumpedPort
    lumped: List[LumpedEdge]
    src_idx: np.ndarray       # global flat node indices
    src_comp: np.ndarray
    src_amp: np.ndarray
    v_idx: np.ndarray
    v_comp: np.ndarray
    v_w: np.ndarray
    i_idx: np.ndarray
    i_comp: np.ndarray
    i_w: np.ndarray


@dataclass
class VoxelScene:
    eps_r: np.ndarray         # [nz-1][ny-1][nx-1]
    kappa: np.ndarray
    pec: np.ndarray           # bool [3][nz][ny][nx]
    ports: List[PortOnGrid]

    @property
    def lumped(self) -> List[LumpedEdge]:
        return [le for p in self.ports for le in p.lumped]


def _tol(grid: RectGrid) -> float:
    return 1e-6 * min(float(np.min(np.diff(l))) for l in grid.lines)


def _index_range(lines: np.ndarray, a: float, b: float, tol: float):  # kept for tools/tests
    """Node indices whose coordinate lies in [min(a,b), max(a,b)] (with tolerance)."""
    lo, hi = (a, b) if a <= b else (b, a)
    idx = np.nonzero((lines >= lo - tol) & (lines <= hi + tol))[0]
    return (int(idx[0]), int(idx[-1])) if idx.size else (0, -1)


def _inside_mask(bx: Box, u: float, tol: float, coords: Sequence[np.ndarray]):
    """Boolean block over the sub-grid of points `coords` (one 1-D array per axis, metres) that lie
    inside the (possibly rotated/translated) box.  Returns (mask[z][y][x], index offsets) or None."""
    lo = np.minimum(bx.start, bx.stop) * u
    hi = np.maximum(bx.start, bx.stop) * u
    if bx.matrix is None or np.allclose(bx.matrix, np.eye(4)):
        sel = [np.nonzero((c >= lo[a] - tol) & (c <= hi[a] + tol))[0] for a, c in enumerate(coords)]
        if any(s_.size == 0 for s_ in sel):
            return None
        off = [int(s_[0]) for s_ in sel]
        shape = [int(s_[-1]) - int(s_[0]) + 1 for s_ in sel]
        return np.ones((shape[2], shape[1], shape[0]), bool), off
    M = np.array(bx.matrix, dtype=float)
    M[:3, 3] *= u
    corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    w = (M @ corners.T).T[:, :3]
    sel = [np.nonzero((c >= w[:, a].min() - tol) & (c <= w[:, a].max() + tol))[0] for a, c in enumerate(coords)]
    if any(s_.size == 0 for s_ in sel):
        return None
    off = [int(s_[0]) for s_ in sel]
    sub = [coords[a][sel[a][0]:sel[a][-1] + 1] for a in range(3)]
    Z, Y, X = np.meshgrid(sub[2], sub[1], sub[0], indexing="ij")
    Minv = np.linalg.inv(M)
    P = np.stack([X, Y, Z, np.ones_like(X)], axis=-1) @ Minv.T
    mask = np.ones(X.shape, bool)
    for a in range(3):
        mask &= (P[..., a] >= lo[a] - tol) & (P[..., a] <= hi[a] + tol)
    return mask, off


def voxelize(scene: Scene, grid: RectGrid) -> VoxelScene:
    nx, ny, nz = grid.shape
    u = scene.unit
    tol = _tol(grid)
    eps = np.ones((nz - 1, ny - 1, nx - 1))
    kap = np.zeros_like(eps)
    prio = np.full(eps.shape, -(1 << 30), dtype=np.int64)
    centers = [grid.centers(a) for a in range(3)]
    for mat in scene.materials:
        for bx in mat.boxes:
            r = _inside_mask(bx, u, -tol, centers)     # strict: a cell centre on the surface is outside
            if r is None:
                continue
            mask, off = r
            sl = tuple(slice(off[a], off[a] + mask.shape[2 - a]) for a in (2, 1, 0))
            win = mask & (prio[sl] <= bx.priority)
            e = eps[sl]; k = kap[sl]; p = prio[sl]
            e[win] = mat.eps_r; k[win] = mat.kappa; p[win] = bx.priority
    pec = np.zeros((3, nz, ny, nx), dtype=bool)
    for met in scene.metals:
        for bx in met.boxes:
            r = _inside_mask(bx, u, tol, grid.lines)
            if r is None:
                continue
            node, off = r
            for c in range(3):
                npa = 2 - c                                   # numpy axis of direction c
                if node.shape[npa] < 2:
                    continue
                a = [slice(None)] * 3; b = [slice(None)] * 3
                a[npa] = slice(0, -1); b[npa] = slice(1, None)
                edge = node[tuple(a)] & node[tuple(b)]        # both end nodes inside
                sl = [slice(off[2], off[2] + edge.shape[0]), slice(off[1], off[1] + edge.shape[1]),
                      slice(off[0], off[0] + edge.shape[2])]
                pec[c][tuple(sl)] |= edge
    ports = [_port_on_grid(p, grid, u) for p in scene.ports]
    return VoxelScene(eps, kap, pec, ports)


def _port_on_grid(port: LumpedPort, grid: RectGrid, u: float) -> PortOnGrid:
    """Lumped port = series/parallel resistor network on the edges inside the port box, a soft
    voltage source on the same edges, a voltage line through the box centre and a current loop
    around the box at mid-length ([EXT] openEMS ports.LumpedPort: AddLumpedElement + AddExcitation +
    two AddProbe; called from solver_fdtd_openems_fixed.py:215)."""
    d = port.direction
    a1, a2 = (d + 1) % 3, (d + 2) % 3
    lo = [grid.snap(a, min(port.start[a], port.stop[a]) * u) for a in range(3)]
    hi = [grid.snap(a, max(port.start[a], port.stop[a]) * u) for a in range(3)]
    if hi[d] <= lo[d]:
        raise ValueError("lumped port has zero length along its direction on this mesh")
    sign = 1.0 if port.stop[d] >= port.start[d] else -1.0
    n_ser = hi[d] - lo[d]
    n_par = (hi[a1] - lo[a1] + 1) * (hi[a2] - lo[a2] + 1)
    length = grid.lines[d][hi[d]] - grid.lines[d][lo[d]]
    lumped, s_idx, s_amp = [], [], []
    for p1 in range(lo[a1], hi[a1] + 1):
        for p2 in range(lo[a2], hi[a2] + 1):
            for q in range(lo[d], hi[d]):
                pos = [0, 0, 0]
                pos[d], pos[a1], pos[a2] = q, p1, p2
                if port.R > 0:
                    lumped.append(LumpedEdge(d, pos[0], pos[1], pos[2], n_ser / (port.R * n_par)))
                s_idx.append(grid.flat(*pos))
                # field of -excite/length across the port => unit port voltage for excite = 1
                s_amp.append(-sign * port.excite * grid.d[d][q] / length)
    # voltage: U = -dir * sum of edge voltages along the centre line
    c1 = grid.snap(a1, 0.5 * (port.start[a1] + port.stop[a1]) * u)
    c2 = grid.snap(a2, 0.5 * (port.start[a2] + port.stop[a2]) * u)
    v_idx = []
    for q in range(lo[d], hi[d]):
        pos = [0, 0, 0]
        pos[d], pos[a1], pos[a2] = q, c1, c2
        v_idx.append(grid.flat(*pos))
    # current: loop of dual edges around the port cross-section at the middle edge
    qm = lo[d] + (n_ser - 1) // 2
    acc = {}

    def add(comp, pos, w):
        key = (comp, tuple(pos))
        acc[key] = acc.get(key, 0.0) + w

    for p1 in range(lo[a1], hi[a1] + 1):
        for p2 in range(lo[a2], hi[a2] + 1):
            pos = [0, 0, 0]
            pos[d], pos[a1], pos[a2] = qm, p1, p2
            pm1 = list(pos); pm1[a1] -= 1
            pm2 = list(pos); pm2[a2] -= 1
            add(a2, pos, +1.0); add(a2, pm1, -1.0)
            add(a1, pos, -1.0); add(a1, pm2, +1.0)
    items = [(k, w) for k, w in acc.items() if abs(w) > 0]
    i_idx = np.array([grid.flat(*k[1]) for k, _ in items], np.int64)
    i_comp = np.array([k[0] for k, _ in items], np.int8)
    i_w = np.array([sign * w for _, w in items], np.float32)
    return PortOnGrid(
        port=port, lumped=lumped,
        src_idx=np.array(s_idx, np.int64), src_comp=np.full(len(s_idx), d, np.int8),
        src_amp=np.array(s_amp, np.float32),
        v_idx=np.array(v_idx, np.int64), v_comp=np.full(len(v_idx), d, np.int8),
        v_w=np.full(len(v_idx), -sign, np.float32),
        i_idx=i_idx, i_comp=i_comp, i_w=i_w)
