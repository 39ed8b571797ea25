"""The `rrt` command line front end (C++ host above the C ABI) keeps the reference's CLI contract (src/main.rs:28-55,
README usage) and produces the same pixels as the library path."""
import json
import os
import subprocess

import numpy as np
import pytest

import helpers
from test_host_scene import decode_png

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RRT = os.path.join(ROOT, "rust-ray-tracer_amd", "rrt")


def run_cli(args, cwd=ROOT):
    return subprocess.run([RRT] + args, cwd=cwd, capture_output=True, text=True, timeout=120)


def test_cli_defaults_match_the_library(rrt, tmp_path):
    out = str(tmp_path / "render.png")
    # positional <HEIGHT> <SAMPLES>; --gltf defaults to assets/default.gltf, camera and lights from assets/suzanne.gltf
    p = run_cli(["-o", out, "--seed", "5", "--stats", "40", "8"])
    assert p.returncode == 0, p.stderr
    stats = json.loads(p.stdout.strip().splitlines()[-1])
    assert stats["width"] == 71 and stats["height"] == 40 and stats["spp"] == 8  # width = (40 * 1.7777778) as u32
    img = decode_png(open(out, "rb").read())
    r = rrt.Renderer(seed=5)
    r.upload(helpers.scene_default(rrt))
    accum, st = r.render(71, 40, 8)
    assert st.rays == stats["rays"]
    assert np.array_equal(img, r.tonemap(accum, 8))
    r.close()


def test_cli_options(tmp_path):
    out = str(tmp_path / "a.png")
    # --aspect_ratio changes the image width only (main.rs:76-81); long options accept "=" like clap
    p = run_cli(["--gltf=assets/suzanne.gltf", "--aspect_ratio", "1.0", "--output", out, "-t", "2", "32", "2"])
    assert p.returncode == 0, p.stderr
    assert decode_png(open(out, "rb").read()).shape == (32, 32, 3)
    # the reference silently loses the image when the output directory is missing (main.rs:116); here it is an error
    p = run_cli(["-o", str(tmp_path / "missing" / "x.png"), "16", "1"])
    assert p.returncode != 0 and "cannot open" in p.stderr
    p = run_cli(["16"])  # SAMPLES missing
    assert p.returncode == 2 and "USAGE" in p.stderr
    p = run_cli(["-g", "assets/nope.gltf", "16", "1"])
    assert p.returncode != 0 and "cannot open" in p.stderr


def test_cli_extended_loader(tmp_path):
    out = str(tmp_path / "m.png")
    p = run_cli(["-g", "assets/MetalRoughSpheres/MetalRoughSpheres.gltf", "--extended", "--scale", "1000", "-o", out, "--stats", "36", "2"])
    assert p.returncode == 0, p.stderr
    stats = json.loads(p.stdout.strip().splitlines()[-1])
    assert stats["prims"] == 1038801  # 1,038,800 triangles + the light sphere of assets/suzanne.gltf
    # without --extended the external .bin is rejected, as the reference would fail on it (gltf.rs:123)
    p = run_cli(["-g", "assets/MetalRoughSpheres/MetalRoughSpheres.gltf", "-o", out, "36", "2"])
    assert p.returncode != 0


def test_cli_gpus_flag_splits_the_samples_over_group_members(rrt, tmp_path):
    """--gpus N / --devices LIST: one process, one context per listed device, the sample range split over them and
    summed on the first.  The box has one GPU, so the list repeats device 0 (a rehearsal of the partition)."""
    a, b = str(tmp_path / "one.png"), str(tmp_path / "two.png")
    p1 = run_cli(["-o", a, "--seed", "5", "--stats", "40", "8"])
    p2 = run_cli(["-o", b, "--seed", "5", "--stats", "--devices", "0,0", "40", "8"])
    assert p1.returncode == 0 and p2.returncode == 0, p2.stderr
    s1, s2 = (json.loads(p.stdout.strip().splitlines()[-1]) for p in (p1, p2))
    assert s2["gpus"] == 2 and s2["reduce"] == "copy+add" and s1["gpus"] == 1
    assert s1["rays"] == s2["rays"] and s1["samples"] == s2["samples"]
    i1, i2 = decode_png(open(a, "rb").read()).astype(int), decode_png(open(b, "rb").read()).astype(int)
    assert np.abs(i1 - i2).max() <= 1  # same samples, summed in another order
    p = run_cli(["--gpus", "1", "-o", b, "--seed", "5", "--stats", "40", "8"])
    assert p.returncode == 0 and np.array_equal(decode_png(open(b, "rb").read()), i1)
    p = run_cli(["--gpus", "9", "-o", b, "16", "1"])  # more GPUs than the box has
    assert p.returncode != 0 and "device" in p.stderr
