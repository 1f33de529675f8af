"""Host-side graph preparation for the fused spatial kernel.

`edge_index` is the reference's (2,E) int64 COO with [0] = source j, [1] = target i
(graph_constructor.py:141; PyG flow source_to_target).  GATv2Conv strips self loops and re-adds one
per node (modules.py:335), so self loops are dropped here and handled implicitly by the kernel.
The result is a CSR by target plus, per tile of `tile_nodes` consecutive targets, the node-id
window [lo,hi) that covers the tile and all of its sources -- what the kernel stages in LDS.
Built once per distinct edge_index tensor and cached (the graph is static during training).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch

LDS_BYTES = 160 * 1024
C_FEAT = 22


@dataclass
class GraphMeta:
    num_nodes: int
    num_edges: int
    max_deg: int
    tile_nodes: int
    num_tiles: int
    win_max: int
    tile_edges_max: int
    rowptr: torch.Tensor   # int32 (N+1) device
    colidx: torch.Tensor   # int32 (E') device
    tile_lo: torch.Tensor  # int32 (num_tiles) device
    tile_hi: torch.Tensor  # int32 (num_tiles) device
    src_ptr: torch.Tensor      # int32: per tile, W+1 offsets into the tile's edge segment, grouped by source row
    src_col: torch.Tensor      # int32 (E'): ((target - n0) << 16) | slot, tile segments aligned with colidx
    src_ptr_off: torch.Tensor  # int32 (num_tiles): start of each tile's src_ptr run


def csr_by_target(edge_index: np.ndarray, num_nodes: int) -> Tuple[np.ndarray, np.ndarray]:
    src, dst = edge_index[0].astype(np.int64), edge_index[1].astype(np.int64)
    if src.size and (src.min() < 0 or dst.min() < 0 or src.max() >= num_nodes or dst.max() >= num_nodes):
        raise ValueError(f"edge_index has node ids outside [0, {num_nodes})")
    keep = src != dst
    src, dst = src[keep], dst[keep]
    order = np.argsort(dst, kind="stable")          # keeps the given edge order inside each target
    src, dst = src[order], dst[order]
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), src.astype(np.int32)


def tile_windows(rowptr: np.ndarray, colidx: np.ndarray, num_nodes: int, tile_nodes: int):
    num_tiles = (num_nodes + tile_nodes - 1) // tile_nodes
    lo = np.empty(num_tiles, dtype=np.int32)
    hi = np.empty(num_tiles, dtype=np.int32)
    for k in range(num_tiles):
        n0, n1 = k * tile_nodes, min(num_nodes, (k + 1) * tile_nodes)
        cols = colidx[rowptr[n0]:rowptr[n1]]
        lo[k] = min(n0, int(cols.min())) if cols.size else n0
        hi[k] = max(n1, int(cols.max()) + 1) if cols.size else n1
    return lo, hi


CP = 24


def by_source_lists(rowptr: np.ndarray, colidx: np.ndarray, num_nodes: int, tile_nodes: int, lo: np.ndarray,
                    hi: np.ndarray):
    """Per tile, the tile's by-target edge segment regrouped by SOURCE (the backward gathers d x_l per source
    row instead of scattering it with atomics).  Returns (src_ptr, src_col, src_ptr_off) as in include/tecmollm.h."""
    deg = np.diff(rowptr).astype(np.int64)
    tgt = np.repeat(np.arange(num_nodes, dtype=np.int64), deg)           # target of every CSR entry
    src_col = np.zeros(max(colidx.size, 1), dtype=np.int32)
    ptrs, offs, pos = [], [], 0
    for k in range(lo.size):
        n0, n1 = k * tile_nodes, min(num_nodes, (k + 1) * tile_nodes)
        e0, e1 = int(rowptr[n0]), int(rowptr[n1])
        w = colidx[e0:e1].astype(np.int64) - int(lo[k])
        order = np.argsort(w, kind="stable")
        if e1 - e0 >= (1 << 16):
            raise ValueError("a tile has more than 65535 in-edges")
        # (tile target << 16) | position of the edge inside the tile's by-target segment: the kernel's per-edge LDS
        # array is indexed by that position, so the by-source pass needs no second lookup
        src_col[e0:e1] = (((tgt[e0:e1][order] - n0) << 16) | order).astype(np.int32)
        W = int(hi[k] - lo[k])
        p = np.zeros(W + 1, dtype=np.int64)
        np.add.at(p, w + 1, 1)
        ptrs.append(np.cumsum(p).astype(np.int32))
        offs.append(pos)
        pos += W + 1
    return np.concatenate(ptrs), src_col, np.asarray(offs, dtype=np.int32)


MAXI = 40           # items a persistent block may own (csrc/spatial_fwd.hip, spatial_bwd.hip)
SCR_FWD, SCR_BWD = 1192, 2888
RED_FLOATS = 8 * 8 * 4 * 64
MAX_WINDOW = 512    # the backward's by-source pass: two rounds of 256 rows


def _r4(n: int) -> int:
    return (n + 3) & ~3


def lds_bytes_fwd(win: int, tile_nodes: int, tile_edges: int = 0) -> int:
    """Dynamic LDS of spatial_fwd_kernel (must match csrc/spatial_fwd.hip:tecm_spatial_fwd_lds): two item slots."""
    P, wm4 = win | 1, _r4(win)
    return 4 * (2 * (_r4(C_FEAT * P) + wm4 * CP + tile_nodes * CP) + MAXI * 96 + tile_nodes + 1 + tile_edges + SCR_FWD)


def lds_bytes_bwd(win: int, tile_nodes: int, demb: int = 16, tile_edges: int = 0) -> int:
    """Dynamic LDS of spatial_bwd_kernel (must match csrc/spatial_bwd.hip:make_map / red_offset)."""
    P, wm4 = win | 1, _r4(win)
    T, E = tile_nodes, tile_edges
    scr = (_r4((C_FEAT + 1) * P) + wm4 * CP + 4 * T * CP + (E + T) * 4 + _r4(demb * P) + MAXI * 32 + MAXI * 4)
    total = scr + SCR_BWD + (T + 1) + E + (wm4 + 1) + E
    red = 0 if scr >= RED_FLOATS else total
    return 4 * max(total, red + RED_FLOATS)


def build(edge_index: torch.Tensor, num_nodes: int, device: torch.device, demb: int = 16) -> GraphMeta:
    ei = edge_index.detach().cpu().numpy()
    rowptr, colidx = csr_by_target(ei, num_nodes)
    deg = np.diff(rowptr)
    # Largest tile (<= 128 target nodes: two threads per (node, head)) whose kernels fit the LDS.  Windows of at most 256
    # rows are preferred -- the backward's by-source pass then is one round per item and keeps its d x_l sums in half
    # the registers -- as long as that costs less than a quarter of the tile (the 41 x 71 grid: 110 nodes, 256 rows).
    chosen, wide = None, None
    for tn in (128, 120, 112, 110, 104, 96, 80, 64, 48, 32, 16, 8, 4, 2, 1):
        lo, hi = tile_windows(rowptr, colidx, num_nodes, tn)
        wmax = int((hi - lo).max())
        bounds = np.minimum(np.arange(lo.size + 1) * tn, num_nodes)
        emax = int(np.diff(rowptr[bounds]).max())
        if (max(lds_bytes_bwd(wmax, tn, demb, emax), lds_bytes_fwd(wmax, tn, emax)) <= LDS_BYTES
                and wmax <= MAX_WINDOW):
            if wide is None:
                wide = (tn, lo, hi, wmax, emax)
            if wmax <= 256:
                chosen = (tn, lo, hi, wmax, emax)
                break
            if tn < 0.75 * wide[0]:
                break
    if chosen is None or chosen[0] < 0.75 * wide[0]:
        chosen = wide
    if chosen is None:
        raise ValueError(
            "graph bandwidth too large for the 160 KiB LDS neighbour window even with 1-node tiles; "
            "renumber the nodes (e.g. reverse Cuthill-McKee) so that neighbours have nearby ids")
    tn, lo, hi, wmax, emax = chosen
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
    sp, sc, so = by_source_lists(rowptr, colidx, num_nodes, tn, lo, hi)
    return GraphMeta(num_nodes=num_nodes, num_edges=int(colidx.size), max_deg=int(deg.max()) if deg.size else 0,
                     tile_nodes=tn, num_tiles=int(lo.size), win_max=wmax, tile_edges_max=emax,
                     rowptr=to(rowptr),
                     colidx=to(colidx if colidx.size else np.zeros(1, np.int32)), tile_lo=to(lo), tile_hi=to(hi),
                     src_ptr=to(sp), src_col=to(sc), src_ptr_off=to(so))


_cache: Dict[Tuple[int, int, int, str], Tuple[torch.Tensor, GraphMeta]] = {}


def get(edge_index: torch.Tensor, num_nodes: int, device: torch.device, demb: int = 16) -> GraphMeta:
    key = (edge_index.data_ptr(), int(edge_index.shape[1]), num_nodes, str(device))
    hit = _cache.get(key)
    if hit is not None and hit[0] is edge_index:
        return hit[1]
    meta = build(edge_index, num_nodes, device, demb)
    _cache[key] = (edge_index, meta)
    if len(_cache) > 16:
        _cache.pop(next(iter(_cache)))
    return meta
