"""The glTF front end (C++, reached through rrt_gltf_load) against an independent Python reading of the same files,
following the contract of the reference's loader (src/gltf.rs:108-351)."""
import base64
import json
import os

import numpy as np
import pytest

from helpers import ASSETS


def python_compat_triangles(path):
    """gltf.rs:118-244 restated with json + base64 + numpy: base64 buffers, views, accessors decoded from the start of
    their view, u16 indices, primitives[0] only."""
    d = json.load(open(path))
    bufs = [base64.b64decode(b["uri"].split(",")[1]) for b in d["buffers"]]
    views = [bufs[v["buffer"]][v["byteOffset"]:v["byteOffset"] + v["byteLength"]] for v in d["bufferViews"]]
    tris, mats = [], []
    for mesh in d["meshes"]:
        p = mesh["primitives"][0]
        ia, pa = d["accessors"][p["indices"]], d["accessors"][p["attributes"]["POSITION"]]
        idx = np.frombuffer(views[ia["bufferView"]], np.uint16, ia["count"]).astype(np.int64)
        pos = np.frombuffer(views[pa["bufferView"]], np.float32, pa["count"] * 3).reshape(-1, 3)
        n = len(idx) // 3
        tris.append(pos[idx[:3 * n]].reshape(n, 3, 3))
        mats += [p["material"]] * n
    return np.concatenate(tris), np.array(mats)


@pytest.mark.parametrize("name,count", [("default.gltf", 12), ("suzanne.gltf", 968)])
def test_compat_geometry_matches_python_reading(rrt, name, count):
    path = os.path.join(ASSETS, name)
    s = rrt.Scene.from_gltf_file(path, os.path.join(ASSETS, "suzanne.gltf"))
    t = np.concatenate(s._tri_chunks)
    want, mats = python_compat_triangles(path)
    assert len(t) == count == len(want)
    assert np.array_equal(t["v0"], want[:, 0]) and np.array_equal(t["v1"], want[:, 1]) and np.array_equal(t["v2"], want[:, 2])
    assert np.array_equal(t["material"], mats)
    # world order: the triangles, then the light sphere (gltf.rs:325-337)
    assert s.world == [(rrt.OBJ_TRIANGLES, 0, count), (rrt.OBJ_SPHERES, 0, 1)]


def test_default_cube_indices_match_the_documented_triples(rrt):
    s = rrt.Scene.from_gltf_file(os.path.join(ASSETS, "default.gltf"), os.path.join(ASSETS, "suzanne.gltf"))
    t = np.concatenate(s._tri_chunks)
    d = json.load(open(os.path.join(ASSETS, "default.gltf")))
    buf = base64.b64decode(d["buffers"][0]["uri"].split(",")[1])
    pos = np.frombuffer(buf[:96], np.float32).reshape(8, 3)
    assert np.all(np.abs(pos) == 1.0)
    triples = [(0, 4, 6), (0, 6, 2), (3, 2, 6), (3, 6, 7), (7, 6, 4), (7, 4, 5), (5, 1, 3), (5, 3, 7), (1, 0, 2), (1, 2, 3), (5, 4, 0), (5, 0, 1)]
    for k, (a, b, c) in enumerate(triples):
        assert np.array_equal(t[k]["v0"], pos[a]) and np.array_equal(t[k]["v1"], pos[b]) and np.array_equal(t[k]["v2"], pos[c])


def test_material_mapping(rrt):
    # gltf.rs:147-168: |metallic| < 1e-5 -> Lambertian(base colour), else Metal{albedo, fuzz = roughness}; doubleSided ignored
    s = rrt.Scene.from_gltf_file(os.path.join(ASSETS, "default.gltf"), os.path.join(ASSETS, "suzanne.gltf"))
    kind, rgb, param = s.materials[0]
    assert kind == rrt.MAT_LAMBERTIAN and np.allclose(rgb, 0.8)
    s = rrt.Scene.from_gltf_file(os.path.join(ASSETS, "suzanne.gltf"), os.path.join(ASSETS, "suzanne.gltf"))
    kind, rgb, param = s.materials[0]
    assert kind == rrt.MAT_METAL and np.allclose(rgb, (0.3560501, 0.34221137, 0.8000001)) and abs(param - 0.48823532) < 1e-7


def test_camera_and_lights_come_from_the_companion_file(rrt):
    # gltf.rs:317: the reference always reads camera and lights from assets/suzanne.gltf; default.gltf's own light
    # (z = -1.005454) is never seen in the reference.  With the file itself as companion it is.
    own = rrt.Scene.from_gltf_file(os.path.join(ASSETS, "default.gltf"), None)
    ref = rrt.Scene.from_gltf_file(os.path.join(ASSETS, "default.gltf"), os.path.join(ASSETS, "suzanne.gltf"))
    assert abs(own.spheres[0][0][2] - (-1.0054539)) < 1e-6
    assert abs(ref.spheres[0][0][2] - 3.0348077) < 1e-6


def test_compat_rejects_what_the_reference_cannot_load(rrt, tmp_path):
    # external .bin buffers make the reference panic at gltf.rs:123; here: a parse error
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_gltf_file(os.path.join(ASSETS, "MetalRoughSpheres", "MetalRoughSpheres.gltf"), os.path.join(ASSETS, "suzanne.gltf"))
    assert e.value.status == 6
    # serde: every listed field is required (gltf.rs:24-106)
    d = json.load(open(os.path.join(ASSETS, "default.gltf")))
    del d["materials"][0]["doubleSided"]
    p = tmp_path / "missing.gltf"
    p.write_text(json.dumps(d))
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_gltf_file(str(p), os.path.join(ASSETS, "suzanne.gltf"))
    assert e.value.status == 6 and "doubleSided" in str(e.value)
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_gltf_file(str(tmp_path / "nope.gltf"), None)
    assert e.value.status == 5
    q = tmp_path / "broken.gltf"
    q.write_text("{ not json")
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_gltf_file(str(q), None)
    assert e.value.status == 6


def test_extended_loader_reads_metal_rough_spheres(rrt):
    path = os.path.join(ASSETS, "MetalRoughSpheres", "MetalRoughSpheres.gltf")
    s = rrt.Scene.from_gltf_file(path, path, extended=True, scale=1000.0)
    assert s.triangle_count == 1038800  # 98 meshes x 10600 triangles
    t = np.concatenate(s._tri_chunks)
    v = np.concatenate([t["v0"], t["v1"], t["v2"]])
    # node translations (multiples of 0.001) are baked and scaled: 7 x 7 spheres of radius 0.35 per group
    assert np.allclose(v.min(0), (-0.35, -0.35, -3.35), atol=2e-3) and np.allclose(v.max(0), (6.35, 6.35, 0.35), atol=2e-3)
    assert len(s.materials) == 98 and len(s.lights) == 0
    kinds = {m[0] for m in s.materials}
    assert kinds == {rrt.MAT_LAMBERTIAN, rrt.MAT_METAL}


def test_writer_output_is_loadable_in_compat_mode(rrt, tmp_path):
    # the synthetic benchmark scene is emitted in the reference's dialect: u16 indices, one buffer view per accessor
    from rust_ray_tracer_amd import scenes

    p = str(tmp_path / "synthetic.gltf")
    scenes.synthetic_mesh_gltf(p, nx=32, nz=16, cell=0.4, amplitude=2.0, seed=7)
    s = rrt.Scene.from_gltf_file(p, p)
    assert s.triangle_count == 2 * 32 * 16
    want, _ = python_compat_triangles(p)
    t = np.concatenate(s._tri_chunks)
    assert np.array_equal(t["v0"], want[:, 0]) and np.array_equal(t["v2"], want[:, 2])
    # same triangles as the direct generator (order differs: blocks)
    direct = scenes.grid_triangles(scenes.height_field(32, 16, 0.4, 2.0, seed=7))
    key = lambda a: np.sort(np.round(a.reshape(len(a), -1).sum(1).astype(np.float64), 4))
    assert np.allclose(key(direct), key(np.stack([t["v0"], t["v1"], t["v2"]], 1)))
    # every triangle faces up (front faces towards +y, the side the culling test accepts)
    n = np.cross(t["v1"] - t["v0"], t["v2"] - t["v0"])
    assert np.all(n[:, 1] > 0)
    assert len(s.lights) == 1 and s.camera.origin[1] > 0
    d = json.load(open(p))
    assert all(a["type"] in ("VEC3", "SCALAR") for a in d["accessors"]) and all("byteOffset" in v for v in d["bufferViews"])


def _tiny_extended_doc():
    """One triangle, u16 indices, everything in one base64 buffer: positions at 0 (36 bytes), indices at 36 (6 bytes)."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32).tobytes()
    idx = np.array([0, 1, 2], np.uint16).tobytes()
    blob = pos + idx + b"\0\0"
    return {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
    }


def _load_extended(rrt, tmp_path, doc, name="m.gltf"):
    p = tmp_path / name
    p.write_text(json.dumps(doc))
    return rrt.Scene.from_gltf_file(str(p), os.path.join(ASSETS, "suzanne.gltf"), extended=True)


def test_extended_loader_reads_the_tiny_document(rrt, tmp_path):
    s = _load_extended(rrt, tmp_path, _tiny_extended_doc())
    assert s.triangle_count == 1
    t = np.concatenate(s._tri_chunks)[0]
    assert np.array_equal(t["v1"], (1, 0, 0)) and np.array_equal(t["v2"], (0, 1, 0))


@pytest.mark.parametrize("mutate", [
    lambda d: d["meshes"][0]["primitives"][0]["attributes"].__setitem__("POSITION", 7),      # accessor index out of range
    lambda d: d["meshes"][0]["primitives"][0]["attributes"].__setitem__("POSITION", -1),
    lambda d: d["meshes"][0]["primitives"][0].__setitem__("indices", 99),
    lambda d: d["meshes"][0]["primitives"][0].__setitem__("indices", "one"),                # not a number
    lambda d: d["meshes"][0]["primitives"][0].__setitem__("indices", 1e300),
    lambda d: d["accessors"][0].__setitem__("bufferView", 5),
    lambda d: d["accessors"][1].__setitem__("byteOffset", -2),                               # negative offsets
    lambda d: d["bufferViews"][0].__setitem__("byteOffset", -36),
    lambda d: d["bufferViews"][1].__setitem__("byteOffset", 2 ** 64 - 2048),                 # would wrap a size_t sum
    lambda d: d["bufferViews"][1].__setitem__("byteLength", 2 ** 63),
    lambda d: d["accessors"][0].__setitem__("byteOffset", 2 ** 64 - 16),
    lambda d: d["accessors"][0].__setitem__("count", 2 ** 62),
    lambda d: d["bufferViews"][0].__setitem__("byteStride", 2 ** 63),
    lambda d: d["accessors"][1].__setitem__("count", 4),                                     # one index past the view
    lambda d: d["bufferViews"][0].__setitem__("buffer", 3),
    lambda d: d["scenes"][0].__setitem__("nodes", [0.5]),
])
def test_extended_loader_rejects_malformed_references(rrt, tmp_path, mutate):
    """File-supplied indices, counts and offsets that are negative, fractional, huge or out of range are reported as
    parse errors; none of them may reach an array access or wrap a bounds check."""
    d = _tiny_extended_doc()
    mutate(d)
    try:
        s = _load_extended(rrt, tmp_path, d)
    except rrt.RrtError as e:
        assert e.status == 6, e
    else:
        # a reference that is merely skipped (e.g. a scene root that is no node index) loads an empty mesh list
        assert s.triangle_count == 0


def test_compat_loader_rejects_wrapping_offsets(rrt, tmp_path):
    d = json.load(open(os.path.join(ASSETS, "default.gltf")))
    for view_patch in ({"byteOffset": 2 ** 64 - 2048, "byteLength": 4096}, {"byteOffset": 2 ** 40}, {"byteLength": 2 ** 39}):
        e2 = json.loads(json.dumps(d))
        e2["bufferViews"][0].update(view_patch)
        p = tmp_path / "wrap.gltf"
        p.write_text(json.dumps(e2))
        with pytest.raises(rrt.RrtError) as e:
            rrt.Scene.from_gltf_file(str(p), os.path.join(ASSETS, "suzanne.gltf"))
        assert e.value.status == 6
    e3 = json.loads(json.dumps(d))
    e3["accessors"][0]["count"] = 2 ** 39
    p = tmp_path / "count.gltf"
    p.write_text(json.dumps(e3))
    with pytest.raises(rrt.RrtError) as e:
        rrt.Scene.from_gltf_file(str(p), os.path.join(ASSETS, "suzanne.gltf"))
    assert e.value.status == 6
