"""Presentation of a rendered frame: what the reference does with the array
``render`` returns — ``glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB, w, h, 0, GL_RGB, GL_UNSIGNED_BYTE,
image)`` in ``main`` (raytracer.py:338-348) — needs exactly a C-contiguous ``uint8 [H, W, 3]`` with
row 0 at the top.  These helpers keep that contract honest on the host side; they never touch the GPU
path (a ``torch.Tensor`` is brought to the host first)."""
from __future__ import annotations

import numpy as np


def as_host_rgb(image) -> np.ndarray:
    """``uint8 [H, W, 3]``, C-contiguous, from a render() result (numpy array or torch tensor)."""
    if hasattr(image, "detach"):                       # torch.Tensor, possibly on the GPU
        image = image.detach().cpu().numpy()
    a = np.asarray(image)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected a uint8 [H, W, 3] frame, got {a.dtype} {a.shape}")
    return np.ascontiguousarray(a)


def texture_bytes(image) -> tuple[int, int, bytes]:
    """(width, height, tightly packed RGB bytes, top row first): the arguments glTexImage2D takes
    (raytracer.py:346; GL_UNPACK_ALIGNMENT must be 1 when 3 * width is not a multiple of 4)."""
    a = as_host_rgb(image)
    return a.shape[1], a.shape[0], a.tobytes()


def save_image(image, path: str) -> str:
    """Write the frame as PNG/PPM/... by extension.  PPM (P6) needs nothing; other formats use Pillow
    and raise ImportError when it is absent."""
    a = as_host_rgb(image)
    if path.lower().endswith(".ppm"):
        with open(path, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
            f.write(a.tobytes())
        return path
    from PIL import Image                               # optional dependency
    Image.fromarray(a, mode="RGB").save(path)
    return path
