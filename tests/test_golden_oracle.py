"""The oracle against the committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py):
a regression guard for the CPU restatement itself.  The reference has no golden vectors; see the oracle's header."""
import os

import numpy as np
import pytest

import helpers

RAY_SCENES = ["default", "suzanne", "mixed", "instanced", "cornell"]


@pytest.mark.parametrize("name", RAY_SCENES)
def test_oracle_reproduces_golden_hits(rrt, oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"rays_{name}.npz"))
    o = oracle.OracleScene(helpers.SCENES[name](rrt))
    for kind in ("primary", "bounce", "adversarial"):
        got = o.intersect(g[f"{kind}_rays"])
        want = g[f"{kind}_hits"]
        assert np.array_equal(got["prim_id"], want["prim_id"]), (name, kind)
        hit = want["prim_id"] >= 0
        assert np.array_equal(got["t"][hit], want["t"][hit])
        assert np.allclose(got["u"][hit], want["u"][hit], atol=1e-6, equal_nan=True)


@pytest.mark.parametrize("name", ["default", "mixed"])
def test_oracle_reproduces_golden_image(rrt, oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"image_{name}.npz"))
    o = oracle.OracleScene(helpers.SCENES[name](rrt))
    h, w, _ = g["accum"].shape
    accum, rays = o.render(w, h, int(g["spp"]), seed=int(g["seed"]))
    assert rays == int(g["rays"])
    assert np.allclose(accum, g["accum"], rtol=1e-5, atol=1e-6, equal_nan=True)
    assert np.array_equal(oracle.tonemap(g["accum"], int(g["spp"])), g["rgb8"])


def test_golden_batches_cover_hits_misses_and_every_primitive_kind(golden_dir):
    g = np.load(os.path.join(golden_dir, "rays_mixed.npz"))
    ids = np.concatenate([g[f"{k}_hits"]["prim_id"] for k in ("primary", "bounce", "adversarial")])
    assert (ids < 0).any() and (ids >= 0).any()
    n_tris = 2 * 24 * 16
    assert (ids[ids >= 0] < n_tris).any()           # triangles
    assert ((ids >= n_tris) & (ids < n_tris + 3)).any()  # spheres
    assert (ids >= n_tris + 3).any()                # rectangles / cuboid sides / light
