"""The reference's own mesh / environment-map DATA files for row f4 (scenes.rs:627-808), committed as fixtures.

tests/golden/assets/material/ holds data/material/models/Mesh000.obj and Mesh001.obj of the reference checkout
(gzip-compressed OBJ text: data, not source) and data/material/textures/envmap.hdr (data/teapot/textures/envmap.hdr is
the same file).  Mesh002.obj and the two teapot meshes are not in the checkout.  `material_dir(tmp)` unpacks them into the
directory layout the presets read (models/*.obj, textures/envmap.hdr), so the GPU box -- which has no /root/reference --
renders the real meshes too.
"""
import gzip
import os
import shutil

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, "golden", "assets", "material")


def material_dir(tmp_path):
    out = os.path.join(str(tmp_path), "material")
    os.makedirs(os.path.join(out, "models"), exist_ok=True)
    os.makedirs(os.path.join(out, "textures"), exist_ok=True)
    for name in ("Mesh000.obj", "Mesh001.obj"):
        with gzip.open(os.path.join(SRC, "models", name + ".gz"), "rb") as fi, open(os.path.join(out, "models", name), "wb") as fo:
            shutil.copyfileobj(fi, fo)
    shutil.copy(os.path.join(SRC, "textures", "envmap.hdr"), os.path.join(out, "textures", "envmap.hdr"))
    return out
