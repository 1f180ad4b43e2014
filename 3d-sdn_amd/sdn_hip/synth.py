"""Procedural stand-ins for the ShapeNet CAD templates (numpy only).

The reference loads eight ShapeNet car/bus/van OBJ files (geometric/derender3d/models/__init__.py:49-58;
31.5k-72.5k triangles each, two of them absent from the mount); neither they nor any dataset exist on the
benchmark machine, so tests and bench.py build meshes of the same size class here: closed, outward-wound
(counter-clockwise seen from outside) triangle soups with a realistic spread of triangle sizes, small
detail parts and a few exactly degenerate faces, normalised like `ShapenetObj` (unit extent per axis).
"""
import numpy as np


def cube():
    """8 vertices at (+-0.5)^3, 12 outward-wound triangles."""
    v = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = []
    for a, b, c, d in quads:
        f += [(a, b, c), (a, c, d)]
    return v, np.asarray(f, np.int32)


def uv_sphere(n_lat, n_lon):
    """Unit-radius UV sphere; (n_lat-1)*n_lon + 2 vertices, 2*n_lon*(n_lat-1) triangles, outward winding."""
    th = np.linspace(0, np.pi, n_lat + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing='ij')
    ring = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    v = np.concatenate([[[0, 1, 0]], ring, [[0, -1, 0]]]).astype(np.float32)
    south = len(v) - 1
    f = []
    idx = lambda i, j: 1 + i * n_lon + (j % n_lon)
    for j in range(n_lon):
        f.append((0, idx(0, j + 1), idx(0, j)))
        f.append((south, idx(n_lat - 2, j), idx(n_lat - 2, j + 1)))
    for i in range(n_lat - 2):
        for j in range(n_lon):
            a, b, c, d = idx(i, j), idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)
            f.append((a, b, c))
            f.append((a, c, d))
    return v, np.asarray(f, np.int32)


def _superellipsoid(v, ex):
    return np.sign(v) * np.abs(v) ** ex


def car_like(n_tris=45000, seed=0, degenerate=24):
    """Car-sized template: boxy body + cabin + 4 wheels + detail blobs, about `n_tris` triangles.

    Returns (vertices [V,3] float32 normalised to unit extent per axis like ShapenetObj, faces [F,3] int32)."""
    rng = np.random.default_rng(seed)
    parts = []

    def add(v, f, scale, shift):
        parts.append((v * np.asarray(scale, np.float32) + np.asarray(shift, np.float32), f))

    budget = max(n_tris, 2000)
    # body: 55 % of the triangles
    nb = int(np.sqrt(budget * 0.55 / 2))
    v, f = uv_sphere(max(nb, 6), max(nb, 6))
    v = _superellipsoid(v, 0.35)
    v = v + 0.01 * rng.standard_normal(v.shape).astype(np.float32)
    add(v, f, (1.0, 0.32, 0.42), (0, 0, 0))
    # cabin: 15 %
    nc = int(np.sqrt(budget * 0.15 / 2))
    v, f = uv_sphere(max(nc, 5), max(nc, 5))
    add(_superellipsoid(v, 0.5), f, (0.5, 0.25, 0.36), (-0.05, 0.38, 0))
    # wheels: 4 x 5 %
    nw = int(np.sqrt(budget * 0.05 / 2))
    for sx in (-0.6, 0.6):
        for sz in (-0.42, 0.42):
            v, f = uv_sphere(max(nw, 4), max(nw, 4))
            add(_superellipsoid(v, 0.7), f, (0.2, 0.2, 0.08), (sx, -0.3, sz))
    # detail blobs: 10 % in ~40 tiny parts (mirrors, handles, lights)
    nd = int(np.sqrt(budget * 0.10 / 40 / 2))
    for _ in range(40):
        v, f = uv_sphere(max(nd, 3), max(nd, 4))
        c = rng.uniform(-1, 1, 3) * (1.0, 0.3, 0.42)
        add(v, f, rng.uniform(0.01, 0.05, 3), c)
    verts, faces, off = [], [], 0
    for v, f in parts:
        verts.append(v.astype(np.float32))
        faces.append(f + off)
        off += len(v)
    verts = np.concatenate(verts)
    faces = np.concatenate(faces).astype(np.int32)
    if degenerate:
        # exactly degenerate faces (repeated vertex), as CAD exports contain
        pick = rng.integers(0, len(faces), degenerate)
        extra = faces[pick].copy()
        extra[:, 2] = extra[:, 1]
        faces = np.concatenate([faces, extra])
    verts = verts / np.ptp(verts, axis=0)
    return verts.astype(np.float32), faces
