#!/usr/bin/env python3
"""Convert the reference's JPEG textures to the binary PPM files ImageTexture::new reads.

    python tools/make_textures.py /root/reference/source assets/

The C++ host layer has no JPEG decoder (none is installed in this image); PIL decodes, and the
texel values may differ by +-1 LSB from the reference's `jpeg-decoder 0.1.22` (parity unpinned,
DESIGN.md §2). Without these files the scene builders use a procedural stand-in of the same size.
"""
import os
import sys

from PIL import Image


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for stem in ("earthmap", "Saturn", "Jupiter", "Mars"):
        path = os.path.join(src, stem + ".jpg")
        if not os.path.exists(path):
            print("missing", path)
            continue
        im = Image.open(path).convert("RGB")
        with open(os.path.join(dst, stem + ".ppm"), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % im.size)
            f.write(im.tobytes())
        print(stem, im.size)


if __name__ == "__main__":
    main()
