"""Triangle soups of simple convex shapes (object-frame (ntri, 3, 3) float64 arrays), the input format of `compile_mjcf(meshes=)` and
`add_free_bodies(hulls=)`.  The reference's objects are convex meshes appended to the humanoid (uhc/smpllib/smpl_robot.py:1205-1224);
`bench.py --workload ball_objects` and the tests build their boxes here."""
import numpy as np


def box_triangles(hx, hy, hz, center=(0, 0, 0)):
    c = np.array(center, dtype=np.float64)
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) + c
    # vertex index = 4*ix + 2*iy + iz
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, cc, d in quads:
        tris.append([v[a], v[b], v[cc]])
        tris.append([v[a], v[cc], v[d]])
    return np.array(tris)


def prism_triangles(r, h):
    """Upright prism over an equilateral triangle (circumradius r, half height h): its three bottom corners are mutual hull
    neighbours, so it rests level on exactly three contacts."""
    a = np.array([[r * np.cos(t), r * np.sin(t)] for t in (np.pi / 2, np.pi / 2 + 2 * np.pi / 3, np.pi / 2 + 4 * np.pi / 3)])
    lo = [np.r_[p, -h] for p in a]
    hi = [np.r_[p, h] for p in a]
    tris = [[lo[0], lo[2], lo[1]], [hi[0], hi[1], hi[2]]]
    for i in range(3):
        j = (i + 1) % 3
        tris += [[lo[i], lo[j], hi[j]], [lo[i], hi[j], hi[i]]]
    return np.array(tris)


def hull_triangles(verts):
    """Outward-oriented triangles of the convex hull of a vertex set (the compiled Model keeps hull vertices, not faces)."""
    from scipy.spatial import ConvexHull
    verts = np.asarray(verts, dtype=np.float64)
    h = ConvexHull(verts)
    tris = verts[h.simplices].copy()
    n = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    flip = np.einsum("ij,ij->i", n, h.equations[:, :3]) < 0
    tris[flip] = tris[flip][:, ::-1]  # qhull's simplices are not consistently wound
    return tris
