"""TEST INFRASTRUCTURE: numpy restatement of the sparse tile exchange's data format (include/rt355.h,
rt_compact_tiles / rt_place_tiles), used as the checker of the HIP kernels and as the stand-in for them in the
CPU (gloo) tests of BatchExchange's pipeline.  Never imported by the product."""
import numpy as np

ENTRY = 196


def local_row_counts(height, tile_rows, world):
    from raytracer_amd import tiling
    return [tiling.local_row_count(height, tile_rows, r, world) for r in range(world)]


def compact(slab, n_rows, width):
    """slab uint8 [F, rows_alloc, W, 3] (rows_alloc >= n_rows) -> (entries [n, 196], n): the non-black 8x8 tiles of
    the first n_rows rows of every frame, in tile-id order."""
    F = slab.shape[0]
    tx_n, ty_n = (width + 7) // 8, (n_rows + 7) // 8
    pitch = tx_n | 1
    out = []
    for f in range(F):
        for ty in range(ty_n):
            for tx in range(tx_n):
                t = slab[f, ty * 8:min(ty * 8 + 8, n_rows), tx * 8:min(tx * 8 + 8, width)]
                if t.any():
                    e = np.zeros(ENTRY, np.uint8)
                    e[:4] = np.frombuffer(np.uint32((f * ty_n + ty) * pitch + tx).tobytes(), np.uint8)
                    pay = np.zeros((8, 8, 3), np.uint8)
                    pay[:t.shape[0], :t.shape[1]] = t
                    e[4:] = pay.reshape(-1)
                    out.append(e)
    return (np.stack(out) if out else np.zeros((0, ENTRY), np.uint8)), len(out)


def place(entries, counts, n_frames, width, height, tile_rows):
    """entries [world, cap, 196], counts [world] -> frames uint8 [F, H, W, 3] (black where no entry lands)."""
    from raytracer_amd import tiling
    world = entries.shape[0]
    out = np.zeros((n_frames, height, width, 3), np.uint8)
    tx_n = (width + 7) // 8
    pitch = tx_n | 1
    for r in range(world):
        rows = tiling.local_rows(height, tile_rows, r, world)
        ty_n = (len(rows) + 7) // 8
        for k in range(int(counts[r])):
            e = entries[r, k]
            tid = int(np.frombuffer(e[:4].tobytes(), np.uint32)[0])
            f, rem = divmod(tid, ty_n * pitch)
            ty, tx = divmod(rem, pitch)
            pay = e[4:].reshape(8, 8, 3)
            for i in range(8):
                lr = ty * 8 + i
                if lr < len(rows):
                    c = min(8, width - tx * 8)
                    out[f, rows[lr], tx * 8:tx * 8 + c] = pay[i, :c]
    return out
