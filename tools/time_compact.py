"""GPU box: the exchange kernels alone, on config 2's slabs of 256 frames (events, best of 5): rt_compact_tiles of a whole-frame
slab (N = 1) and of rank 0's eighth (N = 8); rt_place_tiles of eight ranks' entries, against a plain fill of the same frames
and against the same call with no entries at all (every tile black).      RT355_LIB=... python tools/time_compact.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from raytracer_amd import tiling
from raytracer_amd._lib import TILE_ENTRY_BYTES
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import Camera

r = Renderer(0)
w, h, n, _, _, D, desc = bench.WORKLOADS["c2"]
F, tr = 256, 8
r.upload_scene(bench.build_scene("c2"))
cams = r.pack_cameras([Camera()] * F)
for G in (1, 8):
    rows_max = tiling.max_local_rows(h, tr, G)
    cap = F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8)
    slab = torch.empty((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0")
    entries = torch.empty((cap, TILE_ENTRY_BYTES), dtype=torch.uint8, device="cuda:0")
    count = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slab, tile_rows=tr, tile_first=0, tile_stride=G)
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r.compact_tiles(slab, entries, count); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"N={G}: compact {min(ts[1:]):.3f} ms, {int(count.item())} tiles of {cap} ids")
# the root's side at N = 8: rt_place_tiles of eight ranks' entries, against a plain fill of the same frames
G = 8
rows_max = tiling.max_local_rows(h, tr, G)
cap = F * (((w + 7) // 8) | 1) * ((rows_max + 7) // 8)
slab = torch.empty((F, rows_max, w, 3), dtype=torch.uint8, device="cuda:0")
entries = torch.empty((G, cap, TILE_ENTRY_BYTES), dtype=torch.uint8, device="cuda:0")
counts = torch.zeros(G, dtype=torch.int32, device="cuda:0")
for rank in range(G):
    r.render_frames(cams, range(F), w, h, 1, max_depth=D, out=slab, tile_rows=tr, tile_first=rank, tile_stride=G)
    r.compact_tiles(slab, entries[rank], counts[rank:rank + 1])
torch.cuda.synchronize()
nmax = int(counts.max().item())
gathered = entries[:, :nmax].contiguous()
out = torch.empty((F, h, w, 3), dtype=torch.uint8, device="cuda:0")
table = {}
def timed(fn):
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:])
def place():
    table["t"] = r.place_tiles(gathered, counts, w, h, tr, out, table.get("t"))
tp = timed(place)
tf = timed(lambda: out.zero_())
print(f"place {tp:.3f} ms ({out.numel() / tp / 1e9:.2f} TB/s of frames), fill {tf:.3f} ms ({out.numel() / tf / 1e9:.2f} TB/s)")
zero = torch.zeros_like(counts)
def place_black():
    table["t"] = r.place_tiles(gathered, zero, w, h, tr, out, table.get("t"))
tb = timed(place_black)
print(f"place with no entries (all black) {tb:.3f} ms")
