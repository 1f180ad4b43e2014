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


def _lognormal_partition(rng, n, sigma, lo, hi):
    """n cell widths covering [lo, hi] whose logarithms are normal with standard deviation `sigma` octaves"""
    w = np.exp2(rng.normal(0.0, sigma, n))
    edges = np.concatenate([[0.0], np.cumsum(w)]) / w.sum()
    return lo + edges * (hi - lo)


def _patch_grid(rng, n_lat, n_lon, sigma):
    """Unit superellipsoid-parameter grid with log-normally uneven spacing: (theta [n_lat+1], phi [n_lon+1])."""
    # (the first / last ring sits 0.6 rows from the pole: cap triangles of ordinary aspect ratio.  Rings at 1e-3 rad made
    # ~3.4 % of all faces needle-thin, nine times the band-path share of the CAD files: profiles/cad_mesh_stats.json)
    cap = 0.6 * np.pi / n_lat
    th = _lognormal_partition(rng, n_lat, sigma, cap, np.pi - cap)
    ph = _lognormal_partition(rng, n_lon, sigma, 0.0, 2 * np.pi)
    return th, ph


def _shell(rng, n_tris, exponent, sigma):
    """A closed superellipsoid shell of ~n_tris triangles on an unevenly spaced (theta, phi) grid (triangle areas spread over
    ~4 sigma octaves; long slivers where a narrow row meets a wide column, as CAD exports have)."""
    n_lat = max(3, int(np.sqrt(n_tris / 4.0)))
    n_lon = max(4, int(n_tris / (2.0 * n_lat)))
    th, ph = _patch_grid(rng, n_lat, n_lon, sigma)
    T, P = np.meshgrid(th, ph[:-1], indexing='ij')
    v = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    v = _superellipsoid(v, exponent).astype(np.float32)
    idx = lambda i, j: i * n_lon + (j % n_lon)
    f = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d = idx(i, j), idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)
            f.append((a, b, c))
            f.append((a, c, d))
    return v, np.asarray(f, np.int32)


def cad_like(n_tris=46000, seed=0, degenerate_share=0.002):
    """Car-class template with the STATISTICS of the ShapeNet CAD files the reference loads (profiles/cad_mesh_stats.json,
    measured by tools/cad_mesh_stats.py on the six OBJs under the reference's geometric/assets -- 31.5k-72.5k triangles):
      * triangle sizes spread over ~19 octaves at the configs[1] pose (median ~4 internal pixels, 1 % above ~500, the largest
        several thousand): coarse body panels next to densely tessellated wheels and trim -- 3/4 of the projected area belongs
        to faces above 64 pixels while half of the faces are below 4;
      * depth complexity ~8: shells inside shells (seats, floor, engine bay, both sides of every panel with fill_back);
      * ~0.2 % exactly degenerate faces; with the needle-thin ones ~0.5 % of the faces are so thin that the rasterizer
        sends them down its band path (`band_path_share`; the CAD files: 0.03 % - 1.35 %, mean 0.39 %).
    car_like() -- near-uniform small triangles, depth complexity ~3 -- is the rasterizer's best case; this is the mesh class
    the reference actually renders.  tests/test_cad_like.py compares mesh_stats(cad_like) with the stored statistics.
    Returns (vertices [V,3] float32 normalised to unit extent per axis like ShapenetObj, faces [F,3] int32)."""
    rng = np.random.default_rng(seed)
    parts = []

    def add(v, f, scale, shift):
        parts.append((v * np.asarray(scale, np.float32) + np.asarray(shift, np.float32), f))
    budget = float(max(n_tris, 4000))
    # outer body: few, large panels
    v, f = _shell(rng, budget * 0.06, 0.35, 1.4)
    add(v, f, (1.0, 0.32, 0.42), (0, 0, 0))
    # cabin / glass
    v, f = _shell(rng, budget * 0.03, 0.5, 1.0)
    add(v, f, (0.5, 0.25, 0.36), (-0.05, 0.38, 0))
    # interior: shells inside the body (floor pan, seats, dashboard, engine bay) -- depth complexity
    for k, (sc, sh) in enumerate([((0.95, 0.29, 0.4), (0, -0.01, 0)), ((0.88, 0.25, 0.36), (0.02, -0.03, 0)),
                                  ((0.8, 0.22, 0.33), (0.03, -0.04, 0)), ((0.4, 0.2, 0.32), (0.5, -0.02, 0)),
                                  ((0.35, 0.24, 0.15), (-0.15, 0.05, 0.17)), ((0.35, 0.24, 0.15), (-0.15, 0.05, -0.17)),
                                  ((0.6, 0.12, 0.34), (-0.25, 0.12, 0)), ((0.9, 0.06, 0.38), (0, -0.2, 0)),
                                  ((0.7, 0.18, 0.3), (0.05, 0.0, 0))]):
        v, f = _shell(rng, budget * 0.045, 0.45, 1.7)
        add(v, f, sc, sh)
    # wheels: densely tessellated tyres with hub shells inside
    for sx in (-0.6, 0.6):
        for sz in (-0.42, 0.42):
            v, f = _shell(rng, budget * 0.065, 0.7, 0.8)
            add(v, f, (0.2, 0.2, 0.08), (sx, -0.3, sz))
            v, f = _shell(rng, budget * 0.02, 0.8, 0.8)
            add(v, f, (0.12, 0.12, 0.07), (sx, -0.3, sz))
    # trim: lights, grille bars, mirrors, handles -- many tiny parts
    n_small = 60
    for _ in range(n_small):
        v, f = _shell(rng, budget * 0.15 / n_small, 1.0, 0.7)
        c = rng.uniform(-1, 1, 3) * (1.0, 0.3, 0.42)
        add(v, f, np.exp(rng.uniform(np.log(0.0005), np.log(0.04))) * rng.uniform(0.6, 1.4, 3), c)   # bolts ... mirrors
    verts, faces, off = [], [], 0
    for v, f in parts:
        verts.append(v.astype(np.float32))
        faces.append(f + off)
        off += len(v)
    verts = np.concatenate(verts)
    faces = np.concatenate(faces).astype(np.int32)
    n_deg = int(round(degenerate_share * len(faces)))
    if n_deg:
        pick = rng.integers(0, len(faces), n_deg)
        extra = faces[pick].copy()
        extra[:, 2] = extra[:, 1]          # exactly degenerate (repeated vertex), as CAD exports contain
        faces = np.concatenate([faces, extra])
    verts = verts / np.ptp(verts, axis=0)
    return verts.astype(np.float32), faces
