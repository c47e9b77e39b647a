"""Mesh readers for the robot link meshes the hot path consumes.

The reference loads every link with ``trimesh.load(path, force='mesh')``
(/root/reference/easyhec/modeling/models/rb_solve/rb_solver.py:23-28): binary STL for xArm7, Collada for Franka,
with trimesh's default ``process=True`` vertex merge.  trimesh is not available here, so the three formats the
reference's configs name are read directly and merged the same way (coincident vertices collapse to one index so
that dr.antialias sees shared edges; face order is preserved).
"""
import struct
import xml.etree.ElementTree as ET

import numpy as np

__all__ = ["load_stl", "load_ply", "load_dae", "load_mesh", "merge_vertices", "merge_corners"]

_MERGE_DIGITS = 8  # trimesh tol.merge = 1e-8


def merge_vertices(vertices, faces, digits=_MERGE_DIGITS):
    """Collapse coincident vertices (|delta| < 10**-digits after rounding), keeping first-occurrence order.

    Returns (vertices float64 [V,3], faces int32 [T,3]).  Unreferenced vertices are dropped.
    """
    vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    key = np.round(vertices * (10.0 ** digits)).astype(np.int64)
    _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inverse = inverse.reshape(-1)
    # renumber unique rows by first occurrence so the result does not depend on the sort order of the keys
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    new_faces = rank[inverse[faces]]
    new_vertices = vertices[first[order]]
    used = np.zeros(new_vertices.shape[0], dtype=bool)
    used[new_faces.reshape(-1)] = True
    if not used.all():
        remap = np.cumsum(used) - 1
        new_faces = remap[new_faces]
        new_vertices = new_vertices[used]
    return new_vertices, new_faces.astype(np.int32)


def merge_corners(positions, normals=None, digits=_MERGE_DIGITS, digits_norm=2):
    """trimesh's ``merge_vertices`` for a triangle soup that carries per-corner normals: corners collapse only when
    position (1e-8) AND normal (1e-2) agree (``merge_norm=False``, ``digits_norm=2`` are trimesh's defaults).  A mesh
    exported with flat per-face normals therefore stays (almost) a vertex soup -- which is what the reference renders
    for the Franka Collada meshes, and what makes ``dr.antialias`` treat nearly every edge as a silhouette edge.
    Returns (vertices [V,3] float64, faces [T,3] int32), first-occurrence order."""
    positions = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
    cols = [np.round(positions * (10.0 ** digits))]
    if normals is not None:
        cols.append(np.round(np.asarray(normals, dtype=np.float64).reshape(-1, 3) * (10.0 ** digits_norm)))
    key = np.column_stack(cols).astype(np.int64)
    _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inverse = inverse.reshape(-1)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    faces = rank[inverse].reshape(-1, 3)
    return positions[first[order]], faces.astype(np.int32)


def load_stl(path, merge=True):
    """Binary (or ASCII) STL -> (vertices [V,3] float64, faces [T,3] int32)."""
    with open(path, "rb") as f:
        data = f.read()
    ntri = struct.unpack_from("<I", data, 80)[0] if len(data) >= 84 else 0
    if len(data) == 84 + 50 * ntri:
        rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=ntri,
                            offset=84)
        verts = rec["v"].reshape(-1, 3).astype(np.float64)
    else:  # ASCII
        verts = []
        for line in data.decode("ascii", errors="ignore").splitlines():
            s = line.split()
            if len(s) == 4 and s[0] == "vertex":
                verts.append([float(s[1]), float(s[2]), float(s[3])])
        verts = np.asarray(verts, dtype=np.float64)
        ntri = verts.shape[0] // 3
    faces = np.arange(3 * ntri, dtype=np.int64).reshape(-1, 3)
    if merge:
        return merge_vertices(verts, faces)
    return verts, faces.astype(np.int32)


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4",
              "uint32": "u4", "float32": "f4", "float64": "f8"}


def load_ply(path, merge=False):
    """PLY (binary little-endian or ASCII; triangles only) -> (vertices, faces)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode("ascii").splitlines()
    fmt = None
    elements = []  # (name, count, [(prop name, type or ('list', ct, it))])
    for line in header:
        s = line.split()
        if not s:
            continue
        if s[0] == "format":
            fmt = s[1]
        elif s[0] == "element":
            elements.append((s[1], int(s[2]), []))
        elif s[0] == "property":
            if s[1] == "list":
                elements[-1][2].append((s[4], ("list", s[2], s[3])))
            else:
                elements[-1][2].append((s[2], s[1]))
    verts = faces = None
    if fmt == "ascii":
        tokens = data[end:].split()
        pos = 0
        for name, count, props in elements:
            rows = []
            for _ in range(count):
                row = {}
                for pname, ptype in props:
                    if isinstance(ptype, tuple):
                        n = int(tokens[pos]); pos += 1
                        row[pname] = [float(t) for t in tokens[pos:pos + n]]; pos += n
                    else:
                        row[pname] = float(tokens[pos]); pos += 1
                rows.append(row)
            if name == "vertex":
                verts = np.array([[r["x"], r["y"], r["z"]] for r in rows], dtype=np.float64)
            elif name == "face":
                k = [p for p, t in props if isinstance(t, tuple)][0]
                faces = np.array([r[k] for r in rows], dtype=np.int64)
    else:
        endian = "<" if fmt == "binary_little_endian" else ">"
        off = end
        for name, count, props in elements:
            if all(not isinstance(t, tuple) for _, t in props):
                dt = np.dtype([(p, endian + _PLY_TYPES[t]) for p, t in props])
                arr = np.frombuffer(data, dtype=dt, count=count, offset=off)
                off += dt.itemsize * count
                if name == "vertex":
                    verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)
            else:
                # assume a single list property of constant length 3 (+ optional scalar props are not supported)
                (pname, (_, ct, it)), = [(p, t) for p, t in props if isinstance(t, tuple)]
                dt = np.dtype([("n", endian + _PLY_TYPES[ct]), ("v", endian + _PLY_TYPES[it], 3)])
                arr = np.frombuffer(data, dtype=dt, count=count, offset=off)
                off += dt.itemsize * count
                if not (arr["n"] == 3).all():
                    raise ValueError("load_ply: only triangle faces are supported")
                if name == "face":
                    faces = arr["v"].astype(np.int64)
    if verts is None or faces is None:
        raise ValueError(f"load_ply: {path} has no vertex/face elements")
    if merge:
        return merge_vertices(verts, faces)
    return verts, faces.astype(np.int32)


def _dae_floats(text):
    return np.array(text.split(), dtype=np.float64)


def load_dae(path, merge=True):
    """Collada -> one triangle mesh with the scene-graph node transforms applied (as trimesh's ``force='mesh'`` does;
    e.g. franka link1.dae carries a <matrix> translation).  Vertices are merged per primitive the way trimesh does
    when normals are present: position AND normal must agree (:func:`merge_corners`).  Unit scaling
    (<unit meter=..>) is NOT applied, like trimesh."""
    root = ET.parse(path).getroot()
    ns = root.tag[:root.tag.index("}") + 1] if root.tag.startswith("{") else ""

    def q(tag):
        return ns + tag

    geoms = {}
    for g in root.iter(q("geometry")):
        gid = g.get("id")
        mesh = g.find(q("mesh"))
        if mesh is None:
            continue
        sources = {}
        for s in mesh.findall(q("source")):
            fa = s.find(q("float_array"))
            if fa is None or fa.text is None:
                continue
            acc = s.find(q("technique_common") + "/" + q("accessor"))
            stride = int(acc.get("stride", "3")) if acc is not None else 3
            sources[s.get("id")] = _dae_floats(fa.text).reshape(-1, stride)
        vmap = {}
        for vs in mesh.findall(q("vertices")):
            for inp in vs.findall(q("input")):
                if inp.get("semantic") == "POSITION":
                    vmap[vs.get("id")] = inp.get("source").lstrip("#")
        parts = []
        for prim in list(mesh.findall(q("triangles"))) + list(mesh.findall(q("polylist"))) + list(
                mesh.findall(q("polygons"))):
            inputs = prim.findall(q("input"))
            nin = max(int(i.get("offset", "0")) for i in inputs) + 1
            voff, vsrc, noff, nsrc = 0, None, 0, None
            for i in inputs:
                if i.get("semantic") == "VERTEX":
                    voff = int(i.get("offset", "0"))
                    vsrc = vmap[i.get("source").lstrip("#")]
                elif i.get("semantic") == "NORMAL":
                    noff = int(i.get("offset", "0"))
                    nsrc = i.get("source").lstrip("#")
            if vsrc is None:
                continue
            pos = sources[vsrc][:, :3]
            if prim.tag == q("polygons"):
                rows = [np.array(p.text.split(), dtype=np.int64).reshape(-1, nin) for p in prim.findall(q("p"))]
            else:
                p = prim.find(q("p"))
                if p is None or p.text is None:
                    continue
                allidx = np.array(p.text.split(), dtype=np.int64).reshape(-1, nin)
                if prim.tag == q("triangles"):
                    rows = None
                    corners = allidx
                else:
                    vc = np.array(prim.find(q("vcount")).text.split(), dtype=np.int64)
                    rows, o = [], 0
                    for c in vc:
                        rows.append(allidx[o:o + c])
                        o += c
            if prim.tag != q("triangles"):
                tl = []
                for pl in rows:
                    for k in range(1, len(pl) - 1):
                        tl += [pl[0], pl[k], pl[k + 1]]
                corners = np.array(tl, dtype=np.int64).reshape(-1, nin)
            cpos = pos[corners[:, voff]]
            cnrm = sources[nsrc][:, :3][corners[:, noff]] if (nsrc is not None and nsrc in sources) else None
            parts.append((cpos, cnrm))
        geoms[gid] = parts

    def node_matrix(node):
        m = np.eye(4)
        for ch in node:
            if ch.tag == q("matrix"):
                m = m @ _dae_floats(ch.text).reshape(4, 4)
            elif ch.tag == q("translate"):
                t = np.eye(4); t[:3, 3] = _dae_floats(ch.text); m = m @ t
            elif ch.tag == q("scale"):
                s = np.eye(4); s[0, 0], s[1, 1], s[2, 2] = _dae_floats(ch.text); m = m @ s
            elif ch.tag == q("rotate"):
                ax, ay, az, ang = _dae_floats(ch.text)
                a = np.array([ax, ay, az]); a = a / (np.linalg.norm(a) + 1e-30)
                th = np.deg2rad(ang)
                K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                r = np.eye(4); r[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K); m = m @ r
        return m

    all_v, all_f, nv = [], [], 0

    def walk(node, parent):
        nonlocal nv
        m = parent @ node_matrix(node)
        for ig in node.findall(q("instance_geometry")):
            for cpos, cnrm in geoms.get(ig.get("url").lstrip("#"), []):
                # trimesh processes (merges) every primitive when it is loaded, in local coordinates, then applies the
                # scene-graph transform and concatenates without merging across primitives
                if merge:
                    pv, pf = merge_corners(cpos, cnrm)
                else:
                    pv, pf = cpos, np.arange(cpos.shape[0], dtype=np.int32).reshape(-1, 3)
                all_v.append(pv @ m[:3, :3].T + m[:3, 3])
                all_f.append(pf.astype(np.int64) + nv)
                nv += pv.shape[0]
        for ch in node.findall(q("node")):
            walk(ch, m)

    scenes = list(root.iter(q("visual_scene")))
    if scenes:
        for sc in scenes:
            for node in sc.findall(q("node")):
                walk(node, np.eye(4))
    if not all_v:  # no scene graph: take every geometry untransformed
        for parts in geoms.values():
            for cpos, cnrm in parts:
                pv, pf = merge_corners(cpos, cnrm) if merge else (cpos, np.arange(cpos.shape[0]).reshape(-1, 3))
                all_v.append(pv)
                all_f.append(np.asarray(pf, dtype=np.int64) + nv)
                nv += pv.shape[0]
    verts = np.concatenate(all_v, axis=0)
    faces = np.concatenate(all_f, axis=0)
    return verts, faces.astype(np.int32)


def load_mesh(path, merge=True):
    """Dispatch on extension (the three formats the reference's configs use)."""
    low = path.lower()
    if low.endswith(".stl"):
        return load_stl(path, merge=merge)
    if low.endswith(".ply"):
        return load_ply(path, merge=merge)
    if low.endswith(".dae"):
        return load_dae(path, merge=merge)
    raise ValueError(f"unsupported mesh format: {path}")
