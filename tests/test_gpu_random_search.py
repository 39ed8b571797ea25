"""Random searches for differences (short runs of tools/stress_*.py; the tools run tens of thousands of cases).

Each case is a random scene (triangle soups with needles and sheets, spheres, rectangles, instanced meshes) under a
random camera or ray batch -- far more hostile than the bundled scenes.  The fixed seeds make the runs reproducible."""
import os
import sys

import pytest

import helpers  # noqa: F401  (puts the repository root on the path)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_closest_hits_match_the_exhaustive_search_on_random_scenes():
    """rrt_intersect (the BVH walk, flat / unrolled / two-level) against the oracle's exhaustive search (the reference's
    primitive tests, src/geometry/triangle.rs:38-69, sphere.rs:52-95, aarect.rs, transform.rs): primitive ids equal and
    t within 1e-5 relative on 150 random scenes x 4096 rays (camera-like, random, axis-parallel, starting on a
    triangle, from 30 scene sizes away)."""
    import stress_intersect

    assert stress_intersect.run(150, 5, verbose=False) == 0


def test_every_form_of_the_render_kernel_gives_the_same_image_on_random_scenes():
    """Convert / scaled box arithmetic, one item at a time / two in flight, whole tiles / split items, unrolled /
    two-level instances: bit-identical images and ray counts on 150 random scenes and cameras."""
    import stress_forms

    assert stress_forms.run(150, 3, verbose=False) == 0


def test_pixel_lists_give_the_same_image_on_random_scenes():
    """Primary rays resolved against per-pixel candidate lists (both forms of the kernel's loop over items, both
    register budgets) against the walked frame: bit-identical on 300 random scenes and cameras without instances and
    on 150 with instanced meshes (unrolled: the lists name primitives together with their instance)."""
    import stress_lists

    assert stress_lists.run(300, 7, verbose=False) == 0
    assert stress_lists.run(150, 8, verbose=False, instanced=True) == 0


def test_same_seed_frames_match_the_oracle_on_random_scenes():
    """Small same-seed frames (all four material kinds, a sphere light, 1 .. 6 samples) against the CPU oracle on about
    130 random scenes and cameras: ray counts within 0.5 %, at least 95 % of the pixels within 1e-4 relative (measured
    over a thousand such frames: all but two have at least 97 %, the lowest 95.6 % at a 175-degree field of view), the
    tone map bit-exact."""
    import stress_render

    assert stress_render.run(200, 2, verbose=False) == 0
