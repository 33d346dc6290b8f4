"""Writes the small synthetic COLMAP model of tests/golden/colmap_small/ in both layouts (bin/ and txt/ describe the SAME model)
and the arrays the oracle (oracle/colmap.py) extracts from it (expected.npz).  The reference ships no model files
(SURVEY.md §4): the fixture exercises the cases its ingestion code distinguishes (src/CalibrationData/ImagePoints/Images.cpp:29-101,
IntrinsicOrientation.cpp:51-71, CalibrationData.cpp:492-538): outlier points (point3D id -1), the same 3D point twice in one image,
sparse / unordered ids, an image without points, a second camera, non-unit quaternions, first Euler angle on both sides of 0.

    python tools/make_colmap_fixture.py
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "colmap_small")
INVALID = 2 ** 64 - 1


def main():
    rs = np.random.default_rng(20241022)
    cams = {2: (1, 640, 480, [500.0, 505.0, 320.0, 240.0]),
            1: (4, 1024, 1024, [3181.25, 3184.75, 511.3, 513.9, -0.0123, 0.0045, 1.5e-4, -2.5e-4])}
    point_ids = [101, 7, 55, 1024, 3, 77, 300, 12, 9, 4096, 65, 18]
    pts = [(pid, tuple(float(v) for v in rs.uniform(-400, 400, 3) + np.array([0, 0, 1500]))) for pid in point_ids]
    imgs = []
    for k, iid in enumerate([7, 3, 12, 5, 9]):
        ang = rs.uniform(-0.15, 0.15, 3)
        if k == 1:
            ang[0] = -0.12      # first Euler angle negative: Eigen's eulerAngles takes its other branch
        if k == 2:
            ang[0] = 0.2
        h = 0.5 * ang
        qx = np.array([np.cos(h[0]), np.sin(h[0]), 0, 0]); qy = np.array([np.cos(h[1]), 0, np.sin(h[1]), 0]); qz = np.array([np.cos(h[2]), 0, 0, np.sin(h[2])])
        def mul(a, b):
            return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                             a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3], a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])
        q = mul(mul(qx, qy), qz) * (1.0 + (1e-9 if k % 2 else 0.0))   # not exactly unit: the reader normalises
        t = rs.uniform(-100, 100, 3)
        p2 = []
        if k != 3:                                                     # image 5 has no points at all
            vis = rs.permutation(len(point_ids))[: 8 + k]
            for j in vis:
                p2.append((float(rs.uniform(0, 1024)), float(rs.uniform(0, 1024)), point_ids[j]))
                if rs.uniform() < 0.35:
                    p2.append((float(rs.uniform(0, 1024)), float(rs.uniform(0, 1024)), INVALID))       # outlier
            p2.append((float(rs.uniform(0, 1024)), float(rs.uniform(0, 1024)), p2[0][2]))              # the first 3D point again: neglected
        imgs.append((iid, tuple(float(v) for v in q), tuple(float(v) for v in t), 1, f"frame_{iid:04d}.png", p2))
    os.makedirs(os.path.join(OUT, "bin"), exist_ok=True); os.makedirs(os.path.join(OUT, "txt"), exist_ok=True)
    # ---- binary (COLMAP's little-endian layout) ----
    with open(os.path.join(OUT, "bin", "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, (model, w, h, params) in cams.items():
            f.write(struct.pack("<IiQQ", cid, model, w, h)); f.write(struct.pack("<%dd" % len(params), *params))
    with open(os.path.join(OUT, "bin", "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(imgs)))
        for iid, q, t, cid, name, p2 in imgs:
            f.write(struct.pack("<I4d3dI", iid, *q, *t, cid)); f.write(name.encode() + b"\x00"); f.write(struct.pack("<Q", len(p2)))
            for x, y, pid in p2:
                f.write(struct.pack("<ddQ", x, y, pid))
    with open(os.path.join(OUT, "bin", "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(pts)))
        for n, (pid, c) in enumerate(pts):
            track = [(imgs[(n + a) % len(imgs)][0], a) for a in range(n % 3 + 2)]
            f.write(struct.pack("<Q3d3BdQ", pid, *c, 10, 20, 30, 0.5 + 0.01 * n, len(track)))
            for a, b in track:
                f.write(struct.pack("<II", a, b))
    # ---- text (the same model; repr() round-trips doubles exactly) ----
    names = ["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV"]
    with open(os.path.join(OUT, "txt", "cameras.txt"), "w") as f:
        f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n# Number of cameras: %d\n" % len(cams))
        for cid, (model, w, h, params) in cams.items():
            f.write(f"{cid} {names[model]} {w} {h} " + " ".join(repr(p) for p in params) + "\n")
    with open(os.path.join(OUT, "txt", "images.txt"), "w") as f:
        f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
        for iid, q, t, cid, name, p2 in imgs:
            f.write(f"{iid} " + " ".join(repr(v) for v in q + t) + f" {cid} {name}\n")
            f.write(" ".join(f"{x!r} {y!r} {-1 if pid == INVALID else pid}" for x, y, pid in p2) + "\n")
    with open(os.path.join(OUT, "txt", "points3D.txt"), "w") as f:
        f.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n")
        for n, (pid, c) in enumerate(pts):
            track = [(imgs[(n + a) % len(imgs)][0], a) for a in range(n % 3 + 2)]
            f.write(f"{pid} " + " ".join(repr(v) for v in c) + f" 10 20 30 {0.5 + 0.01 * n!r} " + " ".join(f"{a} {b}" for a, b in track) + "\n")
    from oracle import colmap as oc
    eb, et = oc.calib_data(os.path.join(OUT, "bin")), oc.calib_data(os.path.join(OUT, "txt"))
    for k in eb:
        if k != "binary":
            assert np.array_equal(np.asarray(eb[k]), np.asarray(et[k])), k      # both layouts hold the same model
    np.savez(os.path.join(OUT, "expected.npz"), **{k: np.asarray(v) for k, v in eb.items()})
    print("wrote", OUT, {k: np.asarray(v).shape for k, v in eb.items()})


if __name__ == "__main__":
    main()
