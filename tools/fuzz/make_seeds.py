"""Seed corpus for the parser fuzzer: the JPEG variants the loader meets (and the ones it hands to Pillow)."""
import io, os, sys
import numpy as np
from PIL import Image

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(0)
k = 0
for (w, h) in ((64, 128), (17, 9), (8, 8)):
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for kw in (dict(quality=75), dict(quality=95, subsampling=0), dict(quality=50, subsampling=1), dict(quality=75, optimize=True),
               dict(quality=75, progressive=True), dict(quality=75, restart_marker_blocks=1), dict(quality=75, restart_marker_rows=1)):
        buf = io.BytesIO()
        try:
            Image.fromarray(a).save(buf, "JPEG", **kw)
        except TypeError:
            continue
        open(os.path.join(out, f"seed{k:02d}.jpg"), "wb").write(buf.getvalue()); k += 1
    buf = io.BytesIO(); Image.fromarray(a[..., 0]).save(buf, "JPEG"); open(os.path.join(out, f"seed{k:02d}.jpg"), "wb").write(buf.getvalue()); k += 1
print(k, "seeds")
