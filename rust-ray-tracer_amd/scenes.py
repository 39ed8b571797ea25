"""Benchmark and test scenes.

* ``cornell_box``          -- the reference's programmatic Cornell box (``src/scene.rs:192-246,495-517``).
* ``write_gltf_compat``    -- emits a triangle soup as a glTF file the reference's own loader accepts
                              (``src/gltf.rs``: base64 buffers, u16 indices, <= 65536 vertices per mesh, one
                              bufferView per accessor, every serde-required field present, camera + point light).
* ``synthetic_mesh``       -- fixed-seed displaced height field, 2 * nx * nz triangles (default 1,048,576).
* ``synthetic_instanced``  -- one mesh instanced on a jittered grid as ``Translate(RotateY(mesh))``.
* ``synthetic_flattened``  -- the same scene with every instance written out: ten million unique triangles.
* ``metal_rough_spheres``  -- ``assets/MetalRoughSpheres`` through the extended loader.

All generators are deterministic (integer hashing, no global RNG state).
"""
from __future__ import annotations

import base64
import json
import math
import os
import tempfile

import numpy as np


# --------------------------------------------------------------------------------------------- helpers
def _hash_u32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def _value_noise(ix: np.ndarray, iz: np.ndarray, period: int, seed: int) -> np.ndarray:
    """Smooth value noise on an integer lattice of the given period (bilinear, smoothstep)."""
    fx, fz = ix / period, iz / period
    x0, z0 = np.floor(fx).astype(np.int64), np.floor(fz).astype(np.int64)
    tx, tz = fx - x0, fz - z0
    tx, tz = tx * tx * (3 - 2 * tx), tz * tz * (3 - 2 * tz)

    def lat(a, b):
        with np.errstate(over="ignore"):
            h = _hash_u32((a * 73856093) ^ (b * 19349663) ^ (seed * 83492791))
        return h.astype(np.float64) / 4294967295.0

    v00, v10, v01, v11 = lat(x0, z0), lat(x0 + 1, z0), lat(x0, z0 + 1), lat(x0 + 1, z0 + 1)
    return (v00 * (1 - tx) + v10 * tx) * (1 - tz) + (v01 * (1 - tx) + v11 * tx) * tz


def look_at_c2w(eye, target, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """Camera-to-world 3x4 affine of a camera at `eye` looking down its -z axis towards `target`."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    m = np.zeros((3, 4), np.float32)
    m[:, 0], m[:, 1], m[:, 2], m[:, 3] = r, u, -f, eye
    return m


def _quat_from_c2w(m: np.ndarray):
    """Unit quaternion (x, y, z, w) of the rotation part of a camera-to-world matrix."""
    r = np.asarray(m[:, :3], np.float64)
    t = r[0, 0] + r[1, 1] + r[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        w, x, y, z = 0.25 * s, (r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s
    elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
        s = math.sqrt(1.0 + r[0, 0] - r[1, 1] - r[2, 2]) * 2
        w, x, y, z = (r[2, 1] - r[1, 2]) / s, 0.25 * s, (r[0, 1] + r[1, 0]) / s, (r[0, 2] + r[2, 0]) / s
    elif r[1, 1] > r[2, 2]:
        s = math.sqrt(1.0 + r[1, 1] - r[0, 0] - r[2, 2]) * 2
        w, x, y, z = (r[0, 2] - r[2, 0]) / s, (r[0, 1] + r[1, 0]) / s, 0.25 * s, (r[1, 2] + r[2, 1]) / s
    else:
        s = math.sqrt(1.0 + r[2, 2] - r[0, 0] - r[1, 1]) * 2
        w, x, y, z = (r[1, 0] - r[0, 1]) / s, (r[0, 2] + r[2, 0]) / s, (r[1, 2] + r[2, 1]) / s, 0.25 * s
    return [x, y, z, w]


# --------------------------------------------------------------------------------------------- height field
def height_field(nx: int, nz: int, cell: float, amplitude: float, seed: int = 1234):
    """Vertices ((nx+1)*(nz+1), 3) float32 of a displaced grid in the xz plane, centred on the origin."""
    ix, iz = np.meshgrid(np.arange(nx + 1), np.arange(nz + 1), indexing="ij")
    h = np.zeros(ix.shape, np.float64)
    amp, period = 1.0, 256
    octave = 0
    while period >= 2:
        h += amp * _value_noise(ix, iz, period, seed + octave)
        amp *= 0.5
        period //= 2
        octave += 1
    h = (h / 2.0 - 0.5) * amplitude
    x = (ix - nx / 2.0) * cell
    z = (iz - nz / 2.0) * cell
    return np.stack([x, h, z], axis=-1).astype(np.float32)


def grid_triangles(verts: np.ndarray) -> np.ndarray:
    """(nx+1, nz+1, 3) vertex grid -> (2*nx*nz, 3, 3) triangles whose front faces (the side e1 x e2 points to,
    the only side the reference's culling test accepts, triangle.rs:42) look towards +y."""
    a, b = verts[:-1, :-1], verts[1:, :-1]
    c, d = verts[:-1, 1:], verts[1:, 1:]
    t1 = np.stack([a, c, b], axis=-2)  # (c-a) x (b-a) = z x x = +y
    t2 = np.stack([b, c, d], axis=-2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)]).astype(np.float32)


# --------------------------------------------------------------------------------------------- glTF writer
def write_gltf_compat(path: str, blocks, materials, camera_c2w: np.ndarray, yfov: float, aspect: float,
                      light_pos, light_intensity: float = 1000.0):
    """blocks: list of (vertices (n,3) float32 with n <= 65536, indices (m,) uint16 with m % 3 == 0, material index).
    materials: list of (base colour rgb, metallic, roughness)."""
    buffers, views, accessors, meshes, nodes = [], [], [], [], []
    for bi, (verts, idx, mat) in enumerate(blocks):
        verts = np.ascontiguousarray(verts, np.float32)
        idx = np.ascontiguousarray(idx, np.uint16)
        assert len(verts) <= 65536 and len(idx) % 3 == 0
        for payload, kind in ((verts.tobytes(), "VEC3"), (idx.tobytes(), "SCALAR")):
            buffers.append({"byteLength": len(payload),
                            "uri": "data:application/octet-stream;base64," + base64.b64encode(payload).decode()})
            views.append({"buffer": len(buffers) - 1, "byteLength": len(payload), "byteOffset": 0})
            acc = {"bufferView": len(views) - 1, "componentType": 5126 if kind == "VEC3" else 5123,
                   "count": len(verts) if kind == "VEC3" else len(idx), "type": kind}
            if kind == "VEC3":
                acc["min"], acc["max"] = verts.min(0).tolist(), verts.max(0).tolist()
            accessors.append(acc)
        meshes.append({"name": f"block{bi}", "primitives": [
            {"attributes": {"POSITION": len(accessors) - 2}, "indices": len(accessors) - 1, "material": int(mat)}]})
        nodes.append({"mesh": bi, "name": f"block{bi}"})
    mesh_nodes = list(range(len(nodes)))
    # light and camera: a parent with the translation/rotation and one child carrying the payload, as Blender exports them
    nodes.append({"extensions": {"KHR_lights_punctual": {"light": 0}}, "name": "Light_Orientation"})
    nodes.append({"children": [len(nodes) - 1], "name": "Light", "translation": [float(v) for v in light_pos]})
    light_node = len(nodes) - 1
    nodes.append({"camera": 0, "name": "Camera_Orientation"})
    nodes.append({"children": [len(nodes) - 1], "name": "Camera", "rotation": _quat_from_c2w(camera_c2w),
                  "translation": [float(v) for v in camera_c2w[:, 3]]})
    cam_node = len(nodes) - 1
    doc = {
        "asset": {"generator": "rrt scenes.write_gltf_compat", "version": "2.0"},
        "extensionsUsed": ["KHR_lights_punctual"], "extensionsRequired": ["KHR_lights_punctual"],
        "extensions": {"KHR_lights_punctual": {"lights": [
            {"color": [1, 1, 1], "intensity": light_intensity, "type": "point", "name": "Light"}]}},
        "scene": 0,
        "scenes": [{"name": "Scene", "nodes": mesh_nodes + [light_node, cam_node]}],
        "nodes": nodes,
        "cameras": [{"name": "Camera", "perspective": {"aspectRatio": aspect, "yfov": yfov, "zfar": 100, "znear": 0.1},
                     "type": "perspective"}],
        "materials": [{"doubleSided": True, "name": f"mat{i}", "pbrMetallicRoughness": {
            "baseColorFactor": [float(c[0]), float(c[1]), float(c[2]), 1.0], "metallicFactor": float(m),
            "roughnessFactor": float(r)}} for i, (c, m, r) in enumerate(materials)],
        "meshes": meshes, "accessors": accessors, "bufferViews": views, "buffers": buffers,
    }
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        json.dump(doc, f)
    os.replace(tmp, path)


def _grid_blocks(verts: np.ndarray, block: int, material_of_block):
    """Cuts an (nx+1, nz+1, 3) vertex grid into blocks of `block` x `block` cells with u16 indices."""
    nx, nz = verts.shape[0] - 1, verts.shape[1] - 1
    out = []
    for bx in range(0, nx, block):
        for bz in range(0, nz, block):
            ex, ez = min(bx + block, nx), min(bz + block, nz)
            sub = verts[bx:ex + 1, bz:ez + 1]
            w = sub.shape[1]
            i, j = np.meshgrid(np.arange(ex - bx), np.arange(ez - bz), indexing="ij")
            a, b, c, d = i * w + j, (i + 1) * w + j, i * w + j + 1, (i + 1) * w + j + 1
            idx = np.concatenate([np.stack([a, c, b], -1).reshape(-1, 3), np.stack([b, c, d], -1).reshape(-1, 3)])
            out.append((sub.reshape(-1, 3), idx.reshape(-1).astype(np.uint16), material_of_block(bx // block, bz // block)))
    return out


def synthetic_mesh_gltf(path: str, nx: int = 1024, nz: int = 512, cell: float = 0.05, amplitude: float = 6.0, seed: int = 1234):
    """Writes the synthetic displaced-mesh scene (2*nx*nz triangles) as a reference-loadable glTF file."""
    verts = height_field(nx, nz, cell, amplitude, seed)
    mats = [((0.75, 0.72, 0.68), 0.0, 0.5), ((0.85, 0.65, 0.25), 1.0, 0.15), ((0.35, 0.55, 0.30), 0.0, 0.5),
            ((0.80, 0.80, 0.85), 1.0, 0.4)]
    blocks = _grid_blocks(verts, 128, lambda i, j: int(_hash_u32(np.array([i * 131 + j * 17 + seed]))[0] % 4))
    half_x, half_z = nx * cell / 2, nz * cell / 2
    eye = (0.0, amplitude * 1.6, half_z * 1.35)
    c2w = look_at_c2w(eye, (0.0, -amplitude * 0.3, -half_z * 0.2))
    write_gltf_compat(path, blocks, mats, c2w, yfov=0.6, aspect=16.0 / 9.0, light_pos=(half_x * 0.3, amplitude * 2.2, 0.0),
                      light_intensity=4000.0)


def _cache_dir() -> str:
    d = os.environ.get("RRT_SCENE_CACHE", os.path.join(tempfile.gettempdir(), "rrt_scene_cache_%d" % os.getuid()))
    os.makedirs(d, exist_ok=True)
    return d


def synthetic_mesh(rrt, assets_dir: str, subdiv: int = 512, seed: int = 1234):
    """The 2 * (2*subdiv) * subdiv triangle scene, loaded back through the glTF front end in compat mode."""
    path = os.path.join(_cache_dir(), f"synthetic_{subdiv}_{seed}.gltf")
    if not os.path.exists(path):
        # several ranks of one job may get here at the same time: each writes a file of its own and moves it into
        # place in one step, so that nobody ever reads a half-written scene
        tmp = f"{path}.{os.getpid()}.tmp"
        synthetic_mesh_gltf(tmp, nx=2 * subdiv, nz=subdiv, cell=0.05 * 512 / subdiv, seed=seed)
        os.replace(tmp, path)
    return rrt.Scene.from_gltf_file(path, path)


def blob_mesh(n_lat: int, n_lon: int, radius: float, bump: float, seed: int) -> np.ndarray:
    """A closed bumpy blob (displaced lat/long sphere) with outward-facing front faces: 2*n_lat*n_lon triangles
    (minus the degenerate ones at the poles, which are kept out)."""
    th = np.linspace(0.0, math.pi, n_lat + 1)
    ph = np.linspace(0.0, 2 * math.pi, n_lon + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    ii, jj = np.meshgrid(np.arange(n_lat + 1), np.arange(n_lon + 1) % n_lon, indexing="ij")
    disp = 1.0 + bump * (_value_noise(ii * 4, jj * 4, 64, seed) - 0.5) * np.sin(T) ** 2
    r = radius * disp
    v = np.stack([r * np.sin(T) * np.cos(P), r * np.cos(T), r * np.sin(T) * np.sin(P)], -1).astype(np.float32)
    a, b, c, d = v[:-1, :-1], v[1:, :-1], v[:-1, 1:], v[1:, 1:]
    t1 = np.stack([a, c, b], -2).reshape(-1, 3, 3)
    t2 = np.stack([b, c, d], -2).reshape(-1, 3, 3)
    tris = np.concatenate([t1, t2])
    area2 = np.linalg.norm(np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]), axis=1)
    return tris[area2 > 1e-4]


def synthetic_instanced(rrt, assets_dir: str, n_lat: int = 250, n_lon: int = 251, grid: int = 10, seed: int = 99):
    """One ~100k-triangle blob instanced grid*grid times (Translate(RotateY(mesh))) above a ground quad."""
    s = rrt.Scene()
    grey = s.add_material(rrt.Lambertian(0.7, 0.7, 0.7))
    metal = s.add_material(rrt.Metal((0.8, 0.6, 0.3), 0.2))
    ground = s.add_material(rrt.Lambertian(0.45, 0.5, 0.45))
    light_m = s.add_material(rrt.DiffuseLight(4000.0, 4000.0, 4000.0))
    tris = blob_mesh(n_lat, n_lon, 1.0, 0.5, seed)
    mesh = s.new_mesh()
    half = len(tris) // 2
    s.add_triangles(tris[:half], grey, mesh=mesh)
    s.add_triangles(tris[half:], metal, mesh=mesh)
    mi = s.finish_mesh(mesh)
    pitch = 3.0
    ext = grid * pitch
    g = np.array([[[-ext, -1.3, -ext], [-ext, -1.3, ext], [ext, -1.3, -ext]],
                  [[ext, -1.3, -ext], [-ext, -1.3, ext], [ext, -1.3, ext]]], np.float32)
    s.add_triangles(g, ground)
    k = 0
    for i in range(grid):
        for j in range(grid):
            h = _hash_u32(np.array([seed * 7919 + k * 3, seed * 7919 + k * 3 + 1, seed * 7919 + k * 3 + 2])).astype(np.float64) / 4294967295.0
            ox = (i - (grid - 1) / 2) * pitch + (h[0] - 0.5) * 0.8
            oz = (j - (grid - 1) / 2) * pitch + (h[1] - 0.5) * 0.8
            s.add_instance(mi, angle_deg=float(h[2] * 360.0), offset=(ox, 0.0, oz))
            k += 1
    lp = (0.0, 14.0, 0.0)
    s.add_sphere(lp, 0.6, light_m)
    s.add_sphere_light(lp, 0.6)
    s.camera = rrt.Camera(16.0 / 9.0, 40.0, look_at_c2w((0.0, 9.0, ext * 0.75), (0.0, 0.0, 0.0)))
    s.background = (0.051, 0.051, 0.051)
    return s


def synthetic_flattened(rrt, assets_dir: str, n_lat: int = 250, n_lon: int = 251, grid: int = 10, seed: int = 99):
    """The scene of `synthetic_instanced` with every instance written out as triangles of its own: grid*grid copies of
    the ~100k-triangle blob, rotated about y and translated on the host (transform.rs:101-124 gives the point map
    x' = cos*x + sin*z, z' = -sin*x + cos*z), about ten million unique triangles in one flat BVH."""
    src = synthetic_instanced(rrt, assets_dir, n_lat, n_lon, grid, seed)
    s = rrt.Scene()
    grey = s.add_material(rrt.Lambertian(0.7, 0.7, 0.7))
    metal = s.add_material(rrt.Metal((0.8, 0.6, 0.3), 0.2))
    ground = s.add_material(rrt.Lambertian(0.45, 0.5, 0.45))
    light_m = s.add_material(rrt.DiffuseLight(4000.0, 4000.0, 4000.0))
    tris = blob_mesh(n_lat, n_lon, 1.0, 0.5, seed).astype(np.float32)
    half = len(tris) // 2
    ext = grid * 3.0
    g = np.array([[[-ext, -1.3, -ext], [-ext, -1.3, ext], [ext, -1.3, -ext]],
                  [[ext, -1.3, -ext], [-ext, -1.3, ext], [ext, -1.3, ext]]], np.float32)
    s.add_triangles(g, ground)
    for _, sn, cs, off in src.instances:
        v = np.empty_like(tris)
        v[..., 0] = np.float32(cs) * tris[..., 0] + np.float32(sn) * tris[..., 2] + np.float32(off[0])
        v[..., 1] = tris[..., 1] + np.float32(off[1])
        v[..., 2] = -np.float32(sn) * tris[..., 0] + np.float32(cs) * tris[..., 2] + np.float32(off[2])
        s.add_triangles(v[:half], grey)
        s.add_triangles(v[half:], metal)
    lp = (0.0, 14.0, 0.0)
    s.add_sphere(lp, 0.6, light_m)
    s.add_sphere_light(lp, 0.6)
    s.camera = src.camera
    s.background = src.background
    return s


GLASS_EVERY = 7  # material sidecar of metal_rough_spheres: materials 3, 10, 17, ... (14 of the 98) become glass


def metal_rough_spheres(rrt, assets_dir: str, scale: float = 3000.0, dielectric: bool = False):
    """assets/MetalRoughSpheres through the extended loader (external .bin, accessor byteOffset, node translations
    baked, uniform scale), lit by one emissive sphere, viewed by a camera of its own.

    The file has no glass: the reference's glTF route maps a material to Lambertian or Metal only
    (``src/gltf.rs:147-168``).  ``dielectric=True`` applies a material sidecar on top of the loaded scene: every
    material whose index is 3 modulo GLASS_EVERY (one sphere per row of seven) becomes ``Dielectric(1.5)``
    (``src/material/mod.rs:110-148``), so that all four material kinds run on the million-triangle scene."""
    path = os.path.join(assets_dir, "MetalRoughSpheres", "MetalRoughSpheres.gltf")
    s = rrt.Scene.from_gltf_file(path, path, extended=True, scale=scale)
    if dielectric:
        for i in range(3, len(s.materials), GLASS_EVERY):
            s.materials[i] = rrt.Dielectric(1.5)
    r = 0.00035 * scale
    lo, hi = np.array([-r, -r, -0.003 * scale - r]), np.array([0.006 * scale + r, 0.006 * scale + r, r])
    centre = (lo + hi) / 2
    light_m = s.add_material(rrt.DiffuseLight(20000.0, 20000.0, 20000.0))
    lp = (float(centre[0] + 6 * r), float(hi[1] + 10 * r), float(hi[2] + 12 * r))
    s.add_sphere(lp, 2 * r, light_m)
    s.add_sphere_light(lp, 2 * r)
    eye = (float(hi[0] + 14 * r), float(hi[1] + 4 * r), float(hi[2] + 16 * r))
    s.camera = rrt.Camera(16.0 / 9.0, 38.0, look_at_c2w(eye, tuple(float(c) for c in centre)))
    return s


# --------------------------------------------------------------------------------------------- Cornell box
def glam_look_at_rh(eye, center, up) -> np.ndarray:
    """glam 0.20 ``Affine3A::look_at_rh`` (= look_to_lh(eye, eye - center, up)), returned as the 3x4 affine.
    The reference passes this VIEW matrix as ``camera_to_world`` (src/scene.rs:499-513); kept as is."""
    eye, center, up = (np.asarray(v, np.float32) for v in (eye, center, up))

    def norm(v):
        return (v / np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1]) + np.float32(v[2] * v[2]))).astype(np.float32)

    f = norm(eye - center)
    s = norm(np.cross(up, f).astype(np.float32))
    u = np.cross(f, s).astype(np.float32)
    m = np.zeros((3, 4), np.float32)
    m[:, 0] = (s[0], u[0], f[0])
    m[:, 1] = (s[1], u[1], f[1])
    m[:, 2] = (s[2], u[2], f[2])
    m[:, 3] = (-np.dot(s, eye), -np.dot(u, eye), -np.dot(f, eye))
    return m


def cornell_box(rrt, aspect_ratio: float = 1.0, faithful_camera: bool = False):
    """The reference's Cornell box: five walls, a flipped XZ light, two rotated + translated cuboids, one XZ-rect
    light for sampling (src/scene.rs:192-246,495-517).  With ``faithful_camera`` the camera is the reference's
    (a look-at VIEW matrix used as camera_to_world, which looks away from the box); otherwise a camera that
    actually sees the box."""
    s = rrt.Scene()
    red = s.add_material(rrt.Lambertian(0.65, 0.05, 0.05))
    green = s.add_material(rrt.Lambertian(0.12, 0.45, 0.15))
    white = s.add_material(rrt.Lambertian(0.73, 0.73, 0.73))
    light = s.add_material(rrt.DiffuseLight(15.0, 15.0, 15.0))
    s.add_rect(0, 0.0, 555.0, 0.0, 555.0, 555.0, green)
    s.add_rect(0, 0.0, 555.0, 0.0, 555.0, 0.0, red)
    s.add_rect(1, 213.0, 343.0, 227.0, 332.0, 554.0, light, flip_face=True)
    s.add_rect(1, 0.0, 555.0, 0.0, 555.0, 0.0, white)
    s.add_rect(1, 0.0, 555.0, 0.0, 555.0, 555.0, white)
    s.add_rect(2, 0.0, 555.0, 0.0, 555.0, 555.0, white)
    for size, angle, off in (((165.0, 330.0, 165.0), 15.0, (265.0, 0.0, 295.0)), ((165.0, 165.0, 165.0), -18.0, (130.0, 0.0, 65.0))):
        m = s.new_mesh()
        s.add_cuboid((0.0, 0.0, 0.0), size, white, mesh=m)
        s.add_instance(s.finish_mesh(m), angle_deg=angle, offset=off)
    s.add_xzrect_light(213.0, 343.0, 227.0, 332.0, 554.0)
    if faithful_camera:
        c2w = glam_look_at_rh((278.0, -278.0, -800.0), (278.0, -278.0, 0.0), (0.0, 1.0, 0.0))
    else:
        c2w = look_at_c2w((278.0, 278.0, -800.0), (278.0, 278.0, 0.0))
    s.camera = rrt.Camera(aspect_ratio, 40.0, c2w)
    s.background = (0.0, 0.0, 0.0)
    return s
