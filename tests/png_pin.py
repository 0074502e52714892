"""Shared by tools/make_reference_png_fixture.py and tests/test_reference_png_pin.py.

The only artefacts the reference holds for the hot path are its example renders.  Of those, only
examples/cornell_statue.png and examples/cornell_statue_metal.png were made by a scene the checkout still
contains: `cornell_box_statue()` (src/scenes.rs:200-307, metal line :244-246, yellow two-sided emitter :277-284,
output name "cornell_statue.png" :302).  The dragon / sphere pictures come from older code: their floors show a
checker that Texture::Checkered as committed (material.rs:553-562, keyed on uv with f = 0.1 / 10000) cannot
produce, two_dragons.png is ~10x brighter than the committed un-flipped emitter allows, and
full_sphere_test.png is the commented-out make_world() scene (scenes.rs:16), not sphere_roughness().

data/statue.obj is not in the checkout, so the statue is replaced by a crude proxy of the same placement and
size (pedestal box + bumpy ellipsoid, metal like the statue) and the comparison is made on regions of the
picture that the statue influences only through weak inter-reflection: wall, ceiling, floor-corner and back-wall
patches, the emitter, and the black frame around the box opening (which pins the camera model).
"""
import numpy as np

from tests import scenekit as K

# regions as fractions of the square image: name -> (x0, y0, x1, y1), image x to the right, y down
REGIONS = {
    "frame_left": (0.0, 0.0, 0.018, 1.0),
    "frame_top": (0.0, 0.0, 1.0, 0.018),
    "emitter": (0.43, 0.138, 0.57, 0.158),
    "left_wall_upper": (0.05, 0.18, 0.16, 0.42),
    "left_wall_lower": (0.05, 0.58, 0.16, 0.80),
    "right_wall_upper": (0.84, 0.18, 0.95, 0.42),
    "right_wall_lower": (0.84, 0.58, 0.95, 0.80),
    "ceiling_left": (0.16, 0.06, 0.36, 0.15),
    "ceiling_right": (0.64, 0.06, 0.84, 0.15),
    "back_wall_upper_left": (0.24, 0.24, 0.40, 0.36),
    "back_wall_upper_right": (0.64, 0.24, 0.76, 0.36),
    "floor_front_left": (0.10, 0.90, 0.30, 0.955),
    "floor_front_right": (0.72, 0.90, 0.90, 0.955),
}


def statue_proxy_obj(path):
    """OBJ (object space of the preset's transform translate(374,435,130) * rot_z(pi) * 0.86) of a pedestal box
    x 175..375, y 0..45, z 140..270 and a bumpy ellipsoid body of semi-axes (62, 190, 42) centred at
    (268, 238, 205) -- world extents read off examples/cornell_statue.png with the preset's camera."""
    def to_obj(p):
        p = np.asarray(p, dtype=np.float64)
        return np.stack([(374.0 - p[:, 0]) / 0.86, (435.0 - p[:, 1]) / 0.86, (p[:, 2] - 130.0) / 0.86], axis=1)

    x0, x1, y0, y1, z0, z1 = 175.0, 375.0, 0.5, 45.0, 140.0, 270.0
    c = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1],
                  [x0, y1, z1]])
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    box_p, box_i = [], []
    for q in quads:  # one vertex set per face; slight skew so that no triangle's box is flat (never hit, Q5)
        base = len(box_p)
        for k, vi in enumerate(q):
            box_p.append(c[vi] + 0.01 * np.array([(k * 7) % 3, (k * 5) % 3, (k * 3) % 3]))
        box_i += [base, base + 1, base + 2, base, base + 2, base + 3]
    P, IND, _, _ = K.bumpy_sphere(18, 24, radius=1.0, centre=(0.0, 0.0, 0.0), bump=0.12, normals=False, uvs=False)
    body = P * np.array([62.0, 190.0, 42.0]) + np.array([268.0, 238.0, 205.0])
    allp = np.concatenate([np.array(box_p), body])
    alli = np.concatenate([np.array(box_i, dtype=np.uint32), IND + len(box_p)])
    K.write_obj(path, to_obj(allp), alli)
    return path


def region_means(img, regions=REGIONS):
    """img: (H, W, 3) array -> {name: [r, g, b] mean over the region}."""
    H, W = img.shape[:2]
    out = {}
    for name, (x0, y0, x1, y1) in regions.items():
        a = img[int(round(y0 * H)):max(int(round(y1 * H)), int(round(y0 * H)) + 1),
                int(round(x0 * W)):max(int(round(x1 * W)), int(round(x0 * W)) + 1)]
        out[name] = [float(v) for v in a.reshape(-1, 3).astype(np.float64).mean(axis=0)]
    return out


def inverse_tone_map(v8):
    """8-bit value of util.rs:441-471 (x*0.6 -> ACES approximation -> gamma 2.2 -> *256) back to linear radiance
    (monotone, inverted by bisection; saturated values map to the clamp point)."""
    v = np.clip(np.asarray(v8, dtype=np.float64) / 256.0, 0.0, 1.0) ** 2.2
    lo, hi = np.zeros_like(v), np.full_like(v, 64.0)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        x = mid * 0.6
        y = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0, 1.0)
        hi = np.where(y > v, mid, hi)
        lo = np.where(y > v, lo, mid)
    return 0.5 * (lo + hi)
