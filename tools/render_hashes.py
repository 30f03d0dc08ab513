"""Debug aid (GPU box): SHA-1 of frames rendered by the library named by RT355_LIB, for byte-level A/B of two builds."""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracer_amd.render import Renderer
from raytracer_amd.scenes import cube_scene, default_scene, generate_scene, orbit_camera
r = Renderer(0)
cases = [("default", default_scene(), 640, 480, 5), ("gen8", generate_scene(8, 1, keep_default=True), 1920, 1080, 5),
         ("gen64", generate_scene(64, 2), 1920, 1080, 5), ("gen64_d8", generate_scene(64, 2), 1280, 720, 8),
         ("gen200", generate_scene(200, 5), 960, 540, 5), ("gen1024", generate_scene(1024, 3), 1280, 720, 5),
         ("cubes", cube_scene(), 640, 360, 5)]
for name, sc, w, h, D in cases:
    r.upload_scene(sc)
    for cam in (None, orbit_camera(37, 256), orbit_camera(150, 256)):
        img, hit = r.render(w, h, 1, max_depth=D, rng_mode="K", seed=9, camera=cam, return_hit_index=True)
        print(name, "cam" if cam else "def", hashlib.sha1(img.cpu().numpy().tobytes()).hexdigest()[:16],
              hashlib.sha1(hit.cpu().numpy().tobytes()).hexdigest()[:16])
