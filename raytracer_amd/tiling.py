"""Row-tile ownership for multi-GPU renders: tile t -> rank t mod G.

Pure integer host logic shared by the GPU path and the gloo CPU tests; mirrors the arithmetic of
rt_render_rows / rt_scatter_rows in csrc/rt355.hip."""
from __future__ import annotations

import numpy as np

DEFAULT_TILE_ROWS = 8     # one workgroup tile high: ranks interleave at workgroup granularity


def n_tiles(height: int, tile_rows: int) -> int:
    return (height + tile_rows - 1) // tile_rows


def local_rows(height: int, tile_rows: int, tile_first: int, tile_stride: int) -> np.ndarray:
    """Image rows owned by the rank, in the order rt_render_rows packs them."""
    if tile_rows < 1 or tile_stride < 1 or not (0 <= tile_first < tile_stride):
        raise ValueError("bad tiling")
    rows = []
    for t in range(tile_first, n_tiles(height, tile_rows), tile_stride):
        rows.extend(range(t * tile_rows, min((t + 1) * tile_rows, height)))
    return np.asarray(rows, np.int64)


def local_row_count(height, tile_rows, tile_first, tile_stride) -> int:
    """len(local_rows(...)) in closed form (called once per launch)."""
    if tile_rows < 1 or tile_stride < 1 or not (0 <= tile_first < tile_stride):
        raise ValueError("bad tiling")
    nt = n_tiles(height, tile_rows)
    if tile_first >= nt:
        return 0
    k = (nt - tile_first + tile_stride - 1) // tile_stride
    rows = k * tile_rows
    if (nt - 1 - tile_first) % tile_stride == 0:      # this rank owns the (possibly partial) last tile
        rows -= nt * tile_rows - height
    return rows


def max_local_rows(height: int, tile_rows: int, n_ranks: int) -> int:
    """Slab height every rank pads to for the fixed-size gather (rank 0 owns the most rows)."""
    return max(local_row_count(height, tile_rows, r, n_ranks) for r in range(n_ranks))
