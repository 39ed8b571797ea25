"""Parity gate 2: images of the HIP path (rrt_render / rrt_tonemap through the C ABI) against the CPU oracle.

Same seed: both sides draw the same counter-based random numbers and follow the same paths, so images agree far more
tightly than "statistically" (measured: relative L1 of 3e-8 to 1e-7, apart from the rare path that meets a different
triangle, about one pixel in a thousand).  The remaining differences come from libm vs the device's polynomial
sin/cos, from the 1-ulp hardware reciprocal / reciprocal square root the device uses for normalisations and the final
path weight, and from the throughput form of the estimator.  Stated tolerances:
  * same seed:        at least 99.8 % of the pixels within 1e-4 relative (1e-5 absolute), relative L1 over those pixels
                      <= 1e-5; over the whole image (the few pixels whose path met another triangle included)
                      relative L1 <= 2e-3 and mean luminance within 1e-3 relative; ray counts within 0.1 %
  * different seeds:  mean luminance within 3 sigma of the estimated Monte-Carlo noise, RMSE of the tone-mapped image
                      against the oracle no larger than 1.25 x the RMSE between two oracle renders with different seeds
"""
import os

import itertools

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def luminance(a):
    return 0.2126 * a[..., 0] + 0.7152 * a[..., 1] + 0.0722 * a[..., 2]


def rel_l1(a, b):
    return np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-12)


def assert_same_seed_images_agree(a, b, min_close=0.998, what=None):
    """The same-seed gate: at least `min_close` of the pixels within 1e-4 relative (1e-5 absolute) of each other -- all
    but the rare pixel whose path met another triangle after an ulp-sized change of direction -- and over those pixels
    a relative L1 difference of at most 1e-5 (measured: 3e-8 to 1e-7)."""
    close = np.isclose(a, b, rtol=1e-4, atol=1e-5).all(axis=-1)
    assert close.mean() >= min_close, (close.mean(), what)
    both = np.where(close[..., None], a, 0), np.where(close[..., None], b, 0)
    assert rel_l1(*both) <= 1e-5, (rel_l1(*both), what)


@pytest.mark.parametrize("name,w,h,spp", [("default", 142, 80, 8), ("suzanne", 160, 90, 4), ("mixed", 96, 54, 8),
                                          ("instanced", 96, 54, 4), ("instanced_tris", 96, 54, 4), ("cornell", 64, 64, 8),
                                          ("synthetic", 96, 54, 4)])
def test_same_seed_image_matches_the_oracle(rrt, oracle, name, w, h, spp):
    scene = helpers.SCENES[name](rrt)
    r = rrt.Renderer(device=0, seed=11)
    r.upload(scene)
    accum, st = r.render(w, h, spp)
    ref, ref_rays = oracle.OracleScene(scene).render(w, h, spp, seed=11)
    assert st.samples == w * h * spp
    assert abs(int(st.rays) - ref_rays) <= max(2, ref_rays // 1000)
    finite = np.isfinite(ref) & np.isfinite(accum)
    assert finite.mean() > 0.999
    a, b = np.where(finite, accum, 0), np.where(finite, ref, 0)
    # per-pixel: all but the rare pixel whose path met another triangle after an ulp-sized change of direction agree
    # to float precision (measured: at most 0.1 % of the pixels differ), and on those pixels the images agree to 1e-5
    # of the image's L1 norm
    assert_same_seed_images_agree(a, b, what=name)
    # the whole image, the few diverged paths included
    assert rel_l1(a, b) <= 2e-3, rel_l1(a, b)
    assert abs(luminance(a).mean() - luminance(b).mean()) <= 1e-3 * luminance(b).mean()
    rgb = r.tonemap(accum, spp)
    assert np.array_equal(rgb, oracle.tonemap(accum, spp))  # get_color is bit-exact integer output
    r.close()


@pytest.mark.parametrize("name", ["default", "mixed"])
def test_golden_images(rrt, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"image_{name}.npz"))
    h, w, _ = g["accum"].shape
    r = rrt.Renderer(seed=int(g["seed"]))
    r.upload(helpers.SCENES[name](rrt))
    accum, st = r.render(w, h, int(g["spp"]))
    assert abs(int(st.rays) - int(g["rays"])) <= 2
    assert_same_seed_images_agree(accum, g["accum"], what=name)
    # the 8-bit image: a radiance within 1e-4 can still sit on either side of a quantisation step (never two steps)
    diff = np.abs(r.tonemap(accum, int(g["spp"])).astype(int) - g["rgb8"].astype(int))
    assert (diff > 1).mean() <= 0.002 and (diff > 0).mean() < 0.01
    r.close()


def test_statistical_agreement_across_seeds(rrt, oracle):
    """The reference's own RNG stream cannot be reproduced, so this is the comparison a reference render would get."""
    scene = helpers.scene_default(rrt)
    w, h, spp = 142, 80, 64
    o = oracle.OracleScene(scene)
    ref_a, _ = o.render(w, h, spp, seed=101, mode=oracle.HIT_REF_BVH)
    ref_b, _ = o.render(w, h, spp, seed=202, mode=oracle.HIT_REF_BVH)
    r = rrt.Renderer(seed=303)
    r.upload(scene)
    gpu, _ = r.render(w, h, spp)
    la, lb, lg = (luminance(x / spp) for x in (ref_a, ref_b, gpu))
    # mean luminance: within 3 sigma of the pixel-noise estimate taken from the two oracle renders
    sigma = np.std(la - lb) / np.sqrt(2.0) / np.sqrt(la.size)
    assert abs(lg.mean() - 0.5 * (la.mean() + lb.mean())) < 3.0 * np.sqrt(1.5) * sigma + 1e-4 * la.mean()
    ta, tb, tg = (oracle.tonemap(x, spp).astype(np.float64) for x in (ref_a, ref_b, gpu))
    rmse_ref = np.sqrt(np.mean((ta - tb) ** 2))
    rmse_gpu = np.sqrt(np.mean((tg - ta) ** 2))
    assert rmse_gpu <= 1.25 * rmse_ref + 0.5, (rmse_gpu, rmse_ref)
    r.close()


def test_spp_slices_add_up_and_renders_are_deterministic(rrt):
    scene = helpers.scene_mixed(rrt)
    r = rrt.Renderer(seed=5)
    r.upload(scene)
    w, h = 120, 68
    whole, st = r.render(w, h, 32)
    again, st2 = r.render(w, h, 32)
    assert np.array_equal(whole, again) and st.rays == st2.rays  # bitwise reproducible (ordered chunk reduce, no float atomics)
    parts = [r.render(w, h, n, off) for off, n in ((0, 8), (8, 8), (16, 16))]
    total = parts[0][0] + parts[1][0] + parts[2][0]
    assert sum(int(p[1].rays) for p in parts) == st.rays
    assert np.allclose(total, whole, rtol=2e-5, atol=1e-5)
    # a different sample range is a different image
    other, _ = r.render(w, h, 32, 32)
    assert not np.array_equal(other, whole)
    r.close()


def test_chunked_and_unchunked_renders_agree(rrt):
    scene = helpers.scene_default(rrt)
    r = rrt.Renderer(seed=9)
    r.upload(scene)
    r.set_tuning(chunks=1)
    one, s1 = r.render(71, 40, 48)
    r.set_tuning(chunks=12)
    many, s2 = r.render(71, 40, 48)
    r.set_tuning(chunks=None)
    auto, s3 = r.render(71, 40, 48)
    assert s1.rays == s2.rays == s3.rays and s1.chunks == 1 and s2.chunks == 12
    assert np.allclose(one, many, rtol=2e-5, atol=1e-5) and np.allclose(one, auto, rtol=2e-5, atol=1e-5)
    r.close()


@pytest.mark.parametrize("spp,chunks", [(50, 12), (25, 6), (33, 8), (7, 7), (5, 9)])
def test_sample_counts_that_do_not_divide_into_their_chunks(rrt, spp, chunks):
    """Chunk lengths differ by at most one sample: 50 samples in 12 chunks are two chunks of 5 and ten of 4.  No chunk
    may be left without samples (the pipelined form of the kernel writes a partial image only for chunks that have
    some, and the chunk sum reads them all), so more chunks than samples are cut down to one sample each.  The
    workspace is dirtied first by a render that fills every chunk."""
    r = rrt.Renderer(seed=8)
    r.upload(helpers.scene_synthetic_small(rrt))
    w, h = 256, 256
    r.set_tuning(chunks=chunks, overlap_items=1)
    r.render(w, h, 4 * chunks)  # every chunk of the workspace now holds another image's sums
    piped, sp = r.render(w, h, spp)
    assert sp.chunks == min(chunks, spp)
    r.set_tuning(overlap_items=0)
    plain, s0 = r.render(w, h, spp)
    assert sp.rays == s0.rays and np.array_equal(piped, plain)
    r.set_tuning(chunks=1)
    one, s1 = r.render(w, h, spp)
    assert s1.rays == sp.rays and np.allclose(piped, one, rtol=2e-5, atol=1e-5)
    # a frame rendered as the chunks' own sample ranges, one launch each, and added in chunk order is the same image
    # bit for bit: the kernel's partition is the one documented (first sample of chunk c = c * q + min(c, r))
    n = min(chunks, spp)
    q, rem = divmod(spp, n)
    total = None
    for c in range(n):
        part, _ = r.render(w, h, q + (1 if c < rem else 0), spp_offset=c * q + min(c, rem))
        total = part if total is None else total + part
    assert np.array_equal(total, piped)
    # the default chunking of the library (whatever it picks) agrees as well
    r.set_tuning(chunks=None, overlap_items=1)
    auto, sa = r.render(w, h, spp)
    assert sa.rays == sp.rays and np.allclose(auto, one, rtol=2e-5, atol=1e-5)
    r.close()


@pytest.mark.parametrize("name", ["synthetic", "instanced", "instanced_tris"])
def test_image_does_not_depend_on_wave_scheduling(rrt, name):
    """How a wavefront orders its passes (share of descending lanes that keeps a node pass going, how many stragglers
    a trip leaves behind, how many lanes must wait before it does) changes the timing only: every pixel follows the same
    paths and adds its samples in the same order, so the images are identical bit for bit."""
    r = rrt.Renderer(seed=3)
    r.upload(helpers.SCENES[name](rrt))
    base_knobs = dict(chunks=2, leave_walk=0, overlap_items=0)  # one work item per wavefront at a time
    r.set_tuning(**base_knobs)
    base, s0 = r.render(96, 54, 12)
    # overlap_items=1: the pipelined form of the kernel, whose lanes move from one work item to the next by themselves
    for knobs in (dict(leave_walk=6), dict(leave_walk=40, leave_ready=1), dict(leave_walk=63, leave_ready=64),
                  dict(leave_walk=6, node_share_64ths=0), dict(leave_walk=6, node_share_64ths=64),
                  dict(overlap_items=1), dict(overlap_items=1, leave_walk=12),
                  dict(overlap_items=1, leave_walk=40, leave_ready=1), dict(overlap_items=1, blocks_per_cu=2)):
        r.set_tuning(**{**base_knobs, "leave_ready": None, "node_share_64ths": None, "blocks_per_cu": None, **knobs})
        img, st = r.render(96, 54, 12)
        assert st.rays == s0.rays and np.array_equal(img, base), knobs
    r.close()


@pytest.mark.parametrize("name", ["default", "suzanne", "mixed", "cornell", "synthetic", "instanced_tris"])
def test_image_does_not_depend_on_the_form_of_the_box_test(rrt, name):
    """The render kernels' node steps form plane distances either from converted plane indices or, on scenes walked on
    one grid, from the indices' bit patterns read as (denormal) floats on a scaled t axis (rrt_tuning.scaled_slab,
    default on).  Both box tests are conservative and the primitive tests decide, so rays and images are identical bit
    for bit."""
    r = rrt.Renderer(seed=13)
    assert r.upload(helpers.SCENES[name](rrt)).scaled_slab == 1
    for overlap in (0, 1):
        r.set_tuning(overlap_items=overlap, scaled_slab=0)
        plain, s0 = r.render(120, 68, 8)
        r.set_tuning(scaled_slab=1)
        scaled, s1 = r.render(120, 68, 8)
        assert s0.rays == s1.rays and np.array_equal(plain, scaled), (name, overlap)
    r.close()


CUTS = ((97, 55, 12, 2, 0), (97, 55, 12, 3, 5), (64, 33, 11, 4, 0), (61, 47, 13, 5, 0), (50, 31, 27, 9, 0), (41, 23, 35, 17, 3),
        (33, 18, 70, 35, 0), (30, 17, 40, 40, 0), (9, 7, 64, 32, 0), (1, 1, 50, 12, 0),
        # sample counts no power of two divides, cut into the chunk counts split items take (8, 16, 32)
        (40, 24, 100, 8, 0), (40, 24, 100, 16, 0), (37, 21, 100, 32, 7), (24, 14, 250, 8, 0), (24, 14, 250, 16, 1), (24, 14, 250, 32, 0),
        (12, 9, 1000, 8, 0), (12, 9, 1000, 16, 0), (12, 9, 1000, 32, 5))


@pytest.mark.parametrize("name", ["suzanne", "synthetic", "mixed"])
def test_image_does_not_depend_on_how_items_are_cut(rrt, name):
    """Scenes without instances: a work item is either 64 pixels x one sample chunk or a few pixels x
    up to 32 consecutive chunks of each (rrt_tuning.split_items).  Every (chunk, pixel) sum is formed by one lane in
    sample order either way and the chunks are added in the same order, so the images are identical bit for bit: with
    2 .. 40 chunks (items of 32 .. 2 pixels; a last item with fewer chunks than lane groups, chunks of unequal length
    -- 100, 250 and 1000 samples in 8, 16 and 32 chunks), on frames whose edges cut tiles of every shape, and with a
    sample offset."""
    r = rrt.Renderer(seed=21)
    r.upload(helpers.SCENES[name](rrt))
    for overlap, (width, height, spp, chunks, offset) in itertools.product((0, 1), CUTS):
        r.set_tuning(overlap_items=overlap, chunks=chunks, split_items=0)  # both forms of the kernel's loop over items
        whole, s0 = r.render(width, height, spp, spp_offset=offset)
        r.set_tuning(split_items=1)
        cut, s1 = r.render(width, height, spp, spp_offset=offset)
        assert s0.chunks == s1.chunks == chunks
        assert s0.rays == s1.rays and np.array_equal(whole, cut), (name, overlap, width, height, spp, chunks)
    # the library's own choice for sample counts that are no multiple of 8: split items (a power of two of chunks)
    r.set_tuning(overlap_items=None, chunks=None, split_items=None)
    for spp in (100, 250):
        _, st = r.render(256, 144, spp)
        assert st.chunks in (8, 16, 32), (spp, st.chunks)
    r.close()


LIST_FRAMES = ((97, 55, 16, 8, 0), (97, 55, 24, 16, 5), (61, 47, 32, 32, 0), (64, 33, 100, 8, 0), (37, 21, 100, 32, 7),
               (9, 7, 64, 32, 0), (1, 1, 50, 16, 0), (130, 70, 9, 9, 2))


@pytest.mark.parametrize("name", ["default", "suzanne", "mixed", "synthetic", "instanced", "instanced_tris", "cornell"])
def test_image_does_not_depend_on_pixel_lists(rrt, oracle, name):
    """Scenes without instances and scenes whose instances are unrolled into one tree (triangles, spheres and cuboid
    faces under Translate(RotateY): a list entry then names the primitive together with its instance), items of at most
    8 pixels, either form of the loop over items: a pass over the frame's pixels first
    collects, per pixel, the leaves whose boxes the pyramid of the pixel's camera rays touches (rrt_tuning.pixel_lists;
    at most pixel_list_len of them, else the pixel keeps walking the tree), and the pixel's primary rays test those
    leaves in order of entry distance instead of walking.  The list is a superset of the leaves any of the pixel's
    primary rays can enter and the primitive tests decide, so ray counts and images are identical bit for bit with and
    without lists and with lists of any length: on frames whose edges cut items of every shape, with chunks of unequal
    length, a sample offset, both forms of the box arithmetic, and with the counting variant of the kernel -- whose node
    visits must go down, since listed primary rays visit none.  Then the same seed against the oracle."""
    scene = helpers.SCENES[name](rrt)
    r, rc = rrt.Renderer(seed=21), rrt.Renderer(seed=21, count_visits=True)
    r.upload(scene)
    rc.upload(scene)
    listed_somewhere = False
    for overlap, (width, height, spp, chunks, offset) in itertools.product((0, 1), LIST_FRAMES):  # both forms of the loop over items
        r.set_tuning(overlap_items=overlap, split_items=1, chunks=chunks, pixel_lists=0)
        plain, s0 = r.render(width, height, spp, spp_offset=offset)
        for length, scaled, waves in ((None, 1, None), (1, 1, 7), (3, 0, 6), (6, 1, 6)):  # (waves: the kernel's two register budgets)
            r.set_tuning(pixel_lists=1, pixel_list_len=length, scaled_slab=scaled, list_waves=waves)
            img, s1 = r.render(width, height, spp, spp_offset=offset)
            assert s1.chunks == chunks and s0.rays == s1.rays and np.array_equal(plain, img), (name, overlap, width, height, spp, chunks, length)
            if length is None:
                listed_somewhere |= r.info().pixel_list_permille > 0
        r.set_tuning(scaled_slab=None, list_waves=None)
    assert listed_somewhere
    w, h, spp = 96, 54, 16
    for overlap in (0, 1):
        rc.set_tuning(overlap_items=overlap, split_items=1, chunks=8, pixel_lists=0)
        walked, c0 = rc.render(w, h, spp)
        rc.set_tuning(pixel_lists=1)
        listed, c1 = rc.render(w, h, spp)
        assert c0.rays == c1.rays and np.array_equal(walked, listed)
        assert rc.info().pixel_list_permille > 0 and c1.node_visits < c0.node_visits
    r.set_tuning(overlap_items=1, chunks=8, pixel_lists=1, pixel_list_len=None)
    img, st = r.render(w, h, spp)
    assert np.array_equal(img, listed)
    ref, ref_rays = oracle.OracleScene(scene).render(w, h, spp, seed=21)
    assert abs(int(st.rays) - ref_rays) <= max(2, ref_rays // 1000)
    assert_same_seed_images_agree(img, ref, min_close=0.995, what=name)
    r.close()
    rc.close()


def test_pixel_lists_with_hostile_cameras(rrt):
    """Per-pixel leaf lists where the pyramid of a pixel's rays is awkward: a camera inside the geometry (boxes that
    contain the apex), a field of view of 170 degrees (pyramid faces almost in one plane) and of 0.2 degrees (almost
    parallel faces), geometry far from the origin and tiny, flat geometry seen edge-on, a camera matrix scaled by 1e-3
    and 1e3, a camera whose position is 1e5 times the length of its axes, and a frame one pixel wide.  Same rays and
    images as without lists, bit for bit.  (tools/stress_lists.py runs thousands of random cases of this kind.)"""
    g = np.random.default_rng(5)
    cases = [(1.0, 0.0, 40.0, 3.0, 1.0, None), (1.0, 0.0, 60.0, 0.0, 1.0, None), (1.0, 0.0, 170.0, 1.5, 1.0, None),
             (1.0, 0.0, 0.2, 40.0, 1.0, None), (1e-3, 2e3, 40.0, 3.0, 1.0, None), (1e3, -5e4, 40.0, 3.0, 1.0, None),
             (1.0, 0.0, 40.0, 3.0, 1e-3, None), (1.0, 0.0, 40.0, 3.0, 1e3, None), (1.0, 0.0, 40.0, 3.0, 1.0, 1),
             (1.0, 0.0, 40.0, 3.0, 1.0, 2),
             # camera axes far shorter than the camera's distance from the origin: camera.rs:62 rounds the ray's end point at
             # the magnitude of the position, so a pixel's rays scatter over several pixels' pyramids
             (600.0, 0.0, 5.0, 4.0, 0.011, None), (1.0, 1e4, 5.0, 3.0, 0.3, None), (150.0, 0.0, 90.0, 0.5, 0.0107, None),
             (30.0, 0.0, 0.5, 40.0, 1.77, None)]
    for scale, offset, fov, back, cam_len, flat_axis in cases:
        n = 3000
        centres = g.uniform(-1, 1, (n, 1, 3)) * scale
        tris = centres + g.normal(0, 0.04 * scale, (n, 3, 3))
        if flat_axis is not None:
            tris[..., flat_axis] = 0.0
        tris = (tris + offset).astype(np.float32)
        s = rrt.Scene()
        m = s.add_material(rrt.Lambertian(0.7, 0.6, 0.5))
        glass = s.add_material(rrt.Dielectric(1.5))
        s.add_triangles(tris[: n // 2], m)
        s.add_triangles(tris[n // 2:], glass)
        s.add_sphere((offset + 0.3 * scale, offset, offset), 0.2 * scale, m)
        s.background = (0.6, 0.7, 0.9)
        c2w = np.float32([[cam_len, 0, 0, offset], [0, cam_len, 0, offset], [0, 0, cam_len, offset + back * scale]])
        s.camera = rrt.Camera(1.5, fov, c2w)
        r = rrt.Renderer(seed=3)
        r.upload(s)
        for w, h, spp, chunks in ((72, 48, 16, 8), (1, 40, 32, 32), (50, 1, 16, 16)):
            r.set_tuning(overlap_items=1, split_items=1, chunks=chunks, pixel_lists=0)
            plain, s0 = r.render(w, h, spp)
            for overlap in (0, 1):
                r.set_tuning(overlap_items=overlap, pixel_lists=1)
                img, s1 = r.render(w, h, spp)
                assert s0.rays == s1.rays and np.array_equal(plain, img), (scale, offset, fov, back, cam_len, flat_axis, w, h, overlap)
        r.close()


def test_scaled_box_test_with_hostile_coordinates(rrt, oracle):
    """The scaled form of the box test on triangle soups far from the origin, tiny or huge, flat in one axis, seen from
    outside and from inside along an axis; and cameras whose ray directions are too long or too short for it, for which
    the upload must choose the convert form.  Same image as the convert form bit for bit, and as the oracle's."""
    g = np.random.default_rng(77)
    cases = [(1.0, 0.0, None, 1.0), (1e-3, 0.0, None, 1.0), (1e4, 0.0, None, 1.0), (1.0, 1e4, None, 1.0), (50.0, -2e5, None, 1.0),
             (1.0, 0.0, 1, 1.0), (10.0, 500.0, 2, 1.0), (1.0, 7.0, None, 1e-6), (1.0, 7.0, None, 100.0), (1.0, 0.0, None, 1e-9),
             (1.0, 0.0, None, 1e9)]
    for scale, offset, flat_axis, cam_len in cases:
        n = 1200
        centres = g.uniform(-1, 1, (n, 1, 3)) * scale
        tris = centres + g.normal(0, 0.05 * scale, (n, 3, 3))
        if flat_axis is not None:
            tris[..., flat_axis] = 0.25 * scale
            tris[n // 2:, 1, :], tris[n // 2:, 2, :] = tris[n // 2:, 2, :].copy(), tris[n // 2:, 1, :].copy()
        tris = (tris + offset).astype(np.float32)
        s = rrt.Scene()
        m = s.add_material(rrt.Lambertian(0.7, 0.6, 0.5))
        metal = s.add_material(rrt.Metal((0.8, 0.8, 0.9), 0.0))
        s.add_triangles(tris[: n // 2], m)
        s.add_triangles(tris[n // 2:], metal)
        s.background = (0.6, 0.7, 0.9)
        # the camera looks down the z axis (the central rays are close to axis-parallel); its matrix is scaled by
        # cam_len, which scales every ray direction and leaves the image unchanged
        c2w = np.float32([[cam_len, 0, 0, offset], [0, cam_len, 0, offset], [0, 0, cam_len, offset + 3 * scale]])
        s.camera = rrt.Camera(1.0, 40.0, c2w)
        r = rrt.Renderer(seed=21)
        assert r.upload(s).scaled_slab == (1 if 1e-7 < cam_len < 1e7 else 0)
        w, h, spp = 64, 48, 4
        r.set_tuning(scaled_slab=0)
        plain, s0 = r.render(w, h, spp)
        r.set_tuning(scaled_slab=1)
        scaled, s1 = r.render(w, h, spp)
        assert s0.rays == s1.rays and np.array_equal(plain, scaled), (scale, offset, flat_axis, cam_len)
        ref, ref_rays = oracle.OracleScene(s).render(w, h, spp, seed=21)
        assert abs(int(s1.rays) - ref_rays) <= max(2, ref_rays // 500), (scale, offset, flat_axis, cam_len)
        close = np.isclose(scaled, ref, rtol=1e-4, atol=1e-5).all(axis=-1)
        assert close.mean() >= 0.99, (close.mean(), scale, offset, flat_axis, cam_len)
        r.close()


@pytest.mark.parametrize("name,n_inst", [("instanced_tris", 12), ("instanced", 9), ("cornell", 2)])
def test_unrolled_and_two_level_instances_render_the_same_image(rrt, oracle, name, n_inst):
    """The unrolled form of an instanced scene (the default: meshes of triangles, of spheres and rectangles -- the
    Cornell box's Translate(RotateY(Cuboid)), reference src/scene.rs:216-236 -- or of all three) and the two-level form
    follow the same paths: identical ray counts and images, bit for bit, with either form of the box arithmetic; and the
    two-level form, which the other tests no longer reach by default, agrees with the oracle."""
    scene = helpers.SCENES[name](rrt)
    a, b = rrt.Renderer(seed=7), rrt.Renderer(seed=7, unroll_instances=False)
    ia_info, ib_info = a.upload(scene), b.upload(scene)
    assert ia_info.unrolled_instances == n_inst and ia_info.scaled_slab == 1
    assert ib_info.unrolled_instances == 0 and ib_info.scaled_slab == 0
    w, h, spp = (160, 90, 8) if name != "cornell" else (96, 96, 8)
    ia, sa = a.render(w, h, spp)
    ib, sb = b.render(w, h, spp)
    assert sa.rays == sb.rays and np.array_equal(ia, ib)
    a.set_tuning(scaled_slab=0)
    ic, sc = a.render(w, h, spp)
    assert sc.rays == sa.rays and np.array_equal(ic, ia)
    ref, ref_rays = oracle.OracleScene(scene).render(w // 2, h // 2, 4, seed=7)
    small, ss = b.render(w // 2, h // 2, 4)
    assert abs(int(ss.rays) - ref_rays) <= max(2, ref_rays // 1000)
    assert_same_seed_images_agree(small, ref, what=name)
    a.close()
    b.close()


def test_odd_sizes_and_single_sample(rrt, oracle):
    scene = helpers.scene_default(rrt)
    r = rrt.Renderer(seed=2)
    r.upload(scene)
    o = oracle.OracleScene(scene)
    for w, h, spp in ((1, 1, 1), (7, 3, 2), (9, 17, 1), (65, 8, 3), (8, 65, 5)):
        accum, st = r.render(w, h, spp)
        ref, rays = o.render(w, h, spp, seed=2)
        assert accum.shape == (h, w, 3) and st.samples == w * h * spp
        assert int(st.rays) == rays
        assert_same_seed_images_agree(accum, ref, what=(w, h, spp))
        # the pipelined form of the kernel: same image bit for bit, partial tiles and single samples included
        r.set_tuning(overlap_items=1)
        piped, st2 = r.render(w, h, spp)
        r.set_tuning(overlap_items=None)
        assert st2.rays == st.rays and np.array_equal(piped, accum)
    r.close()


def test_furnace(rrt):
    from rust_ray_tracer_amd import scenes

    s = rrt.Scene()
    m = s.add_material(rrt.Lambertian(0.6, 0.5, 0.25))
    s.add_sphere((0, 0, 0), 1.0, m)
    s.background = (0.8, 0.7, 0.5)
    s.camera = rrt.Camera(1.0, 40.0, scenes.look_at_c2w((0, 0, 5), (0, 0, 0)))
    r = rrt.Renderer(seed=3)
    r.upload(s)
    accum, st = r.render(64, 64, 32)
    mean = accum / 32
    assert np.allclose(mean[32, 32], np.array([0.6, 0.5, 0.25]) * np.array([0.8, 0.7, 0.5]), rtol=2e-5)
    assert np.allclose(mean[0, 0], (0.8, 0.7, 0.5), rtol=1e-6)
    r.close()


def test_max_depth_and_tmin_are_honoured(rrt, oracle):
    scene = helpers.scene_mixed(rrt)
    o = oracle.OracleScene(scene)
    for depth in (1, 2, 6):
        r = rrt.Renderer(seed=4, max_depth=depth)
        r.upload(scene)
        accum, st = r.render(64, 36, 4)
        ref, rays = o.render(64, 36, 4, seed=4, max_depth=depth)
        assert abs(int(st.rays) - rays) <= max(2, rays // 500) and st.rays <= depth * st.samples
        # (deeper paths: more chances for an ulp to decide which triangle a bounce meets)
        assert_same_seed_images_agree(accum, ref, min_close=0.995, what=depth)
        r.close()


def test_tonemap_matches_the_reference_formula(rrt, oracle):
    g = np.random.default_rng(8)
    acc = (g.uniform(0, 1, (33, 47, 3)) ** 3 * 40).astype(np.float32)
    acc[0, 0] = (np.nan, -1.0, np.inf)
    acc[5, 7] = (0.0, 1e-30, 1e30)
    r = rrt.Renderer()
    s = rrt.Scene()
    s.add_material(rrt.Lambertian(0.5, 0.5, 0.5))
    r.upload(s)
    for spp in (1, 7, 64):
        assert np.array_equal(r.tonemap(acc, spp), oracle.tonemap(acc, spp))
    r.close()


def test_full_size_frame_properties(rrt):
    """Benchmark configuration 2 (suzanne) at its full 1920x1080 size (few samples): size-independent properties."""
    scene = helpers.scene_suzanne(rrt)
    r = rrt.Renderer(seed=1)
    r.upload(scene)
    w, h, spp = 1920, 1080, 4
    a, sa = r.render(w, h, spp, 0)
    b, sb = r.render(w, h, spp, spp)
    assert sa.samples == w * h * spp and sa.samples <= sa.rays <= 4 * sa.samples
    assert np.all(np.isfinite(a)) and np.all(a >= 0)
    # background pixels carry exactly spp * background
    assert np.allclose(a[0, 0], spp * 0.051, rtol=1e-6) and np.allclose(a[-1, -1], spp * 0.051, rtol=1e-6)
    # two disjoint sample ranges estimate the same image
    la, lb = luminance(a).mean(), luminance(b).mean()
    assert abs(la - lb) < 0.1 * la  # 4 spp under a radiance-1000 light: a few per cent of Monte-Carlo noise in the mean
    assert abs(int(sa.rays) - int(sb.rays)) < 0.01 * sa.rays
    rgb = r.tonemap(a + b, 2 * spp)
    assert rgb.shape == (h, w, 3) and rgb.dtype == np.uint8
    # the image is flipped vertically on output (main.rs:108): the model sits around the middle either way, the light
    # (up in the scene) must come out in the upper half
    bright = np.argwhere(rgb.sum(-1) >= 3 * 250)
    assert len(bright) and bright[:, 0].mean() < h / 2
    r.close()


def test_headline_workload_at_full_size(rrt):
    """The benchmark's default workload (1,048,576 triangles, 3840x2160) at a few samples per pixel: properties that do
    not depend on the size -- bitwise reproducibility, sample slices that add up (what the multi-GPU partition relies
    on), the bounds on rays per sample, and agreement of two independent estimates of the frame."""
    from rust_ray_tracer_amd import scenes

    scene = scenes.synthetic_mesh(rrt, helpers.ASSETS, subdiv=512)
    r = rrt.Renderer(seed=1)
    info = r.upload(scene)
    assert info.tri_count == 1048576 and info.bvh_depth <= 37
    w, h = 3840, 2160
    whole, sw = r.render(w, h, 4, 0)
    again, sa = r.render(w, h, 4, 0)
    assert sw.rays == sa.rays and np.array_equal(whole, again)
    first, s0 = r.render(w, h, 2, 0)
    second, s1 = r.render(w, h, 2, 2)
    assert s0.rays + s1.rays == sw.rays
    assert np.allclose(first + second, whole, rtol=2e-5, atol=1e-5)
    assert sw.samples == w * h * 4 and sw.samples <= sw.rays <= 4 * sw.samples
    assert np.all(np.isfinite(whole)) and np.all(whole >= 0)
    other, so = r.render(w, h, 4, 4)
    assert abs(int(so.rays) - int(sw.rays)) < 0.005 * sw.rays
    la, lb = luminance(whole).mean(), luminance(other).mean()
    assert abs(la - lb) < 0.1 * la
    r.close()


def test_headline_kernel_variant_at_full_size(rrt, oracle):
    """The instantiation of the render kernel the benchmark times -- scene without instances, two items in flight,
    scaled box arithmetic, items of a few pixels x many sample chunks, per-pixel leaf lists for the primary rays -- on
    the benchmark's own scene and frame
    (1,048,576 triangles, 3840x2160: thirty bands of one-pixel-wide tiles, millions of work items through the eight
    queues, up to 32 partial images), against the whole-tile form of the same kernel: identical images and ray
    counts.  It replaces the reference's per-pixel loop (src/main.rs:87-103), and every pixel's sum must be the
    same whichever way the frame is cut.  Cases: 32 samples in 16 chunks (items of 4 pixels), 256 samples in 32 chunks
    (items of 2 pixels), 100 samples (chunks of unequal length, the library's own choice), and the sample slices
    eight GPUs render of the 1024-sample frame (the first two and the last).  Then a crop-sized frame with 16 chunks
    against the oracle at the same seed."""
    from rust_ray_tracer_amd import scenes

    scene = scenes.synthetic_mesh(rrt, helpers.ASSETS, subdiv=512)
    r = rrt.Renderer(seed=1)
    info = r.upload(scene)
    assert info.tri_count == 1048576 and info.scaled_slab == 1 and info.unrolled_instances == 0
    w, h = 3840, 2160
    a, b = helpers.DeviceArray((h, w, 3)), helpers.DeviceArray((h, w, 3))
    for spp, off, chunks, want_chunks in ((32, 0, 16, 16), (256, 0, 32, 32), (100, 0, None, 8), (128, 0, None, 8), (128, 128, None, 8), (128, 896, None, 8)):
        r.set_tuning(chunks=chunks, split_items=None)  # the library's rule: split items on this scene
        r.render_device(w, h, spp, off, a.ptr, 0)
        sa = r.render_stats()
        assert sa.chunks == want_chunks, (spp, sa.chunks)
        if spp >= 100:  # (the library's rule: from a few dozen samples of listed pixels on)
            assert r.info().pixel_list_permille > 500  # primary rays resolved against per-pixel leaf lists
        r.set_tuning(chunks=sa.chunks, split_items=0)  # the same chunks as items of 64 pixels x one chunk
        r.render_device(w, h, spp, off, b.ptr, 0)
        sb = r.render_stats()
        assert sa.samples == sb.samples == w * h * spp and sa.rays == sb.rays, (spp, off)
        ia, ib = a.numpy(), b.numpy()
        assert np.all(np.isfinite(ia)) and np.array_equal(ia, ib), (spp, off, chunks)
        del ia, ib
    a.free()
    b.free()
    # same seed against the oracle, 16 chunks side by side in items of 4 pixels.  The oracle walks its reference-style
    # tree with a box test that carries the interval across the axes (an exhaustive search of a million triangles
    # takes it 4 ms per ray); on a corner of the frame it is checked against the exhaustive search first.
    o = oracle.OracleScene(scene)
    o.build_ref_bvh()
    tiny_a, rays_a = o.render(16, 9, 2, seed=1)
    tiny_b, rays_b = o.render(16, 9, 2, seed=1, mode=oracle.HIT_REF_BVH_RUNNING_SLAB)
    assert rays_a == rays_b and np.array_equal(tiny_a, tiny_b)
    cw, ch, spp = 64, 36, 64
    r.set_tuning(chunks=16, split_items=1)
    accum, st = r.render(cw, ch, spp)
    assert st.chunks == 16
    ref, ref_rays = o.render(cw, ch, spp, seed=1, mode=oracle.HIT_REF_BVH_RUNNING_SLAB)
    assert abs(int(st.rays) - ref_rays) <= max(2, ref_rays // 1000)
    # (a pixel holds 64 samples here, eight to sixteen times the other same-seed frames: as many more chances that one
    # of its paths meets another triangle after an ulp-sized change of direction; measured 0.35 % of the pixels)
    assert_same_seed_images_agree(accum, ref, min_close=0.99)
    r.close()


def test_metal_rough_spheres_asset_matches_the_oracle(rrt, oracle):
    """Benchmark configuration 3: assets/MetalRoughSpheres through the extended loader (external .bin, node
    translations, 1,038,800 triangles, Lambertian and Metal by the metallic factor).  Closest hits against the
    exhaustive search on a ray subset and a small same-seed image."""
    from rust_ray_tracer_amd import scenes

    scene = scenes.metal_rough_spheres(rrt, helpers.ASSETS)
    assert scene.triangle_count == 1038800
    r = rrt.Renderer(seed=4)
    r.upload(scene)
    o = oracle.OracleScene(scene)
    prim = helpers.primary_rays(o, 2560, 1440, 768, seed=21)
    bounce = helpers.bounce_rays(o, prim[:256], seed=22)
    rays = np.concatenate([prim, bounce])
    want = o.intersect(rays)
    assert (want["prim_id"] >= 0).mean() > 0.2
    helpers.assert_hits_match(r.intersect(rays), want)
    w, h, spp = 40, 22, 2
    accum, st = r.render(w, h, spp)
    ref, ref_rays = o.render(w, h, spp, seed=4)
    assert abs(int(st.rays) - ref_rays) <= 2
    assert rel_l1(accum, ref) <= 2e-3
    r.close()


def test_metal_rough_spheres_with_the_glass_sidecar(rrt, oracle):
    """Benchmark configuration 3 as the bench runs it: the same asset with every seventh material turned into
    Dielectric(1.5) (reference src/material/mod.rs:110-148), so that Lambertian, Metal, Dielectric and the emissive light
    all run on the million-triangle scene.  Closest hits against the exhaustive search (the geometry is the same, the
    check is that the sidecar changes materials only), glass really being hit, and a small same-seed image whose paths
    refract and reflect inside the glass spheres."""
    from rust_ray_tracer_amd import scenes

    scene = scenes.metal_rough_spheres(rrt, helpers.ASSETS, dielectric=True)
    kinds = [m[0] for m in scene.materials]
    assert kinds.count(rrt.MAT_DIELECTRIC) == 14 and rrt.MAT_METAL in kinds and rrt.MAT_LAMBERTIAN in kinds and rrt.MAT_DIFFUSE_LIGHT in kinds
    r = rrt.Renderer(seed=4)
    r.upload(scene)
    o = oracle.OracleScene(scene)
    prim = helpers.primary_rays(o, 2560, 1440, 768, seed=31)
    want = o.intersect(prim)
    helpers.assert_hits_match(r.intersect(prim), want)
    tri_mat = np.concatenate(scene._tri_chunks)["material"]
    hit_ids = want["prim_id"][(want["prim_id"] >= 0) & (want["prim_id"] < len(tri_mat))]
    glass_hits = np.isin(tri_mat[hit_ids] % scenes.GLASS_EVERY, [3]).sum()
    assert glass_hits > 10  # primary rays do meet glass spheres
    w, h, spp = 48, 27, 2
    accum, st = r.render(w, h, spp)
    ref, ref_rays = o.render(w, h, spp, seed=4)
    assert abs(int(st.rays) - ref_rays) <= 2
    close = np.isclose(accum, ref, rtol=1e-4, atol=1e-5).all(axis=-1)
    assert close.mean() >= 0.99, close.mean()  # refraction chains amplify an ulp more often than diffuse paths do
    assert rel_l1(accum, ref) <= 2e-3
    # without the sidecar the image differs: the glass is really there
    plain = rrt.Renderer(seed=4)
    plain.upload(scenes.metal_rough_spheres(rrt, helpers.ASSETS))
    other, _ = plain.render(w, h, spp)
    assert not np.allclose(other, accum, rtol=1e-3)
    plain.close()
    r.close()


def test_instanced_workload_at_full_size(rrt, oracle):
    """Benchmark configuration 5 (a 100k-triangle mesh x 100 Translate(RotateY(..)) instances, reference
    src/geometry/transform.rs:22-31,99-130) through the instanced form of the render kernel at its full 7680x4320 frame
    and a few samples: bitwise reproducibility, sample slices that add up, ray bounds, two independent estimates that
    agree -- a crop-sized same-seed frame against the oracle -- and a 128-sample frame, which takes per-pixel candidate
    lists, against the walked one."""
    from rust_ray_tracer_amd import scenes

    scene = scenes.synthetic_instanced(rrt, helpers.ASSETS)
    r = rrt.Renderer(seed=1)
    info = r.upload(scene)
    assert info.prim_count > 10_000_000 and info.tri_count < 200_000  # instanced: ten million ids, one mesh in memory
    w, h = 7680, 4320
    whole, sw = r.render(w, h, 2, 0)
    again, sa = r.render(w, h, 2, 0)
    assert sw.rays == sa.rays and np.array_equal(whole, again)
    first, s0 = r.render(w, h, 1, 0)
    second, s1 = r.render(w, h, 1, 1)
    assert s0.rays + s1.rays == sw.rays
    assert np.allclose(first + second, whole, rtol=2e-5, atol=1e-5)
    del first, second, again
    assert sw.samples == w * h * 2 and sw.samples <= sw.rays <= 4 * sw.samples
    assert np.all(np.isfinite(whole)) and np.all(whole >= 0)
    other, so = r.render(w, h, 2, 2)
    assert abs(int(so.rays) - int(sw.rays)) < 0.005 * sw.rays
    la, lb = luminance(whole).mean(), luminance(other).mean()
    assert abs(la - lb) < 0.1 * la
    del whole, other
    # same seed against the oracle (exhaustive search over ten million transformed primitives: a small frame)
    cw, ch, spp = 32, 18, 1
    accum, st = r.render(cw, ch, spp)
    ref, ref_rays = oracle.OracleScene(scene).render(cw, ch, spp, seed=1)
    assert abs(int(st.rays) - ref_rays) <= 2
    close = np.isclose(accum, ref, rtol=1e-4, atol=1e-5).all(axis=-1)
    assert close.mean() >= 0.99 and rel_l1(accum, ref) <= 2e-3
    # from 128 samples on the library cuts the frame into items of 8 pixels x 8 chunks whose primary rays take per-pixel
    # candidate lists (entries name a triangle together with its instance): against whole tiles that walk, bit for bit
    w2, h2 = 1920, 1080
    listed, sl = r.render(w2, h2, 128)
    assert sl.chunks == 8 and r.info().pixel_list_permille > 300, (sl.chunks, r.info().pixel_list_permille)
    r.set_tuning(split_items=0, pixel_lists=0, chunks=8)
    walked, swk = r.render(w2, h2, 128)
    assert swk.rays == sl.rays and np.array_equal(listed, walked)
    r.close()


def test_flattened_workload_takes_pixel_lists(rrt, oracle):
    """Benchmark configuration 6 (the instanced scene written out as ten million unique triangles, one flat BVH of
    five million nodes): from 128 samples on the library cuts such a frame into items of 8 pixels x 8 sample chunks and
    resolves the primary rays from per-pixel candidate lists.  That launch against whole-tile items whose primary rays
    walk the tree, bit for bit; shorter frames keep whole tiles; and a same-seed frame with the 8K frame's pixel
    footprint (a camera zoomed onto one of the blobs) with lists against the oracle's exhaustive search."""
    from rust_ray_tracer_amd import scenes

    scene = scenes.synthetic_flattened(rrt, helpers.ASSETS)
    r = rrt.Renderer(seed=1)
    info = r.upload(scene)
    assert info.tri_count > 10_000_000 and info.node_count >= (1 << 21)
    w, h = 1920, 1080
    listed, sl = r.render(w, h, 128)
    assert sl.chunks == 8
    share = r.info().pixel_list_permille
    assert share > 300, share  # (pixels four times the 8K frame's: fewer of them see <= 11 candidates)
    r.set_tuning(split_items=0, pixel_lists=0, chunks=8)
    walked, sw = r.render(w, h, 128)
    assert sw.chunks == 8 and sw.rays == sl.rays and np.array_equal(listed, walked)
    del listed, walked
    r.set_tuning(split_items=None, pixel_lists=None, chunks=None)
    _, ss = r.render(w, h, 16)
    assert ss.chunks < 8  # a short frame: whole tiles, no lists
    # the 8K frame's pixel footprint on a frame the exhaustive search can follow
    blob = scenes.synthetic_instanced(rrt, helpers.ASSETS).instances[45][3]
    eye = scene.camera.origin
    fov = float(np.degrees(2.0 * np.arctan(np.tan(np.radians(20.0)) * 9.0 / 4320.0)))
    scene.camera = rrt.Camera(16.0 / 9.0, fov, scenes.look_at_c2w(tuple(float(v) for v in eye), (float(blob[0]), 0.3, float(blob[2]))))
    r.upload(scene)
    cw, ch, spp = 16, 9, 8
    r.set_tuning(split_items=1, pixel_lists=1, chunks=8)
    accum, st = r.render(cw, ch, spp)
    assert r.info().pixel_list_permille > 500
    r.set_tuning(split_items=0, pixel_lists=0, chunks=8)
    plain, sp = r.render(cw, ch, spp)
    assert sp.rays == st.rays and np.array_equal(accum, plain)
    ref, ref_rays = oracle.OracleScene(scene).render(cw, ch, spp, seed=1)
    assert abs(int(st.rays) - ref_rays) <= 2
    close = np.isclose(accum, ref, rtol=1e-4, atol=1e-5).all(axis=-1)
    assert close.mean() >= 0.98 and rel_l1(accum, ref) <= 2e-3
    r.close()


def test_pixel_lists_with_more_instances_than_the_lds_table_holds(rrt):
    """An unrolled scene of 200 instances: the render kernel's LDS copy of the instance transforms holds 127, so the
    listed primitives' transforms come from device memory.  Same rays and image as the walk, bit for bit, in both forms
    of the loop over items."""
    from rust_ray_tracer_amd import scenes

    s = rrt.Scene()
    grey = s.add_material(rrt.Lambertian(0.7, 0.7, 0.7))
    metal = s.add_material(rrt.Metal((0.8, 0.8, 0.9), 0.1))
    light = s.add_material(rrt.DiffuseLight(300.0, 300.0, 300.0))
    m = s.new_mesh()
    s.add_triangles(scenes.blob_mesh(6, 7, 0.45, 0.2, seed=9), grey, mesh=m)
    s.add_sphere((0.0, 0.6, 0.0), 0.15, metal, mesh=m)
    blob = s.finish_mesh(m)
    for k in range(200):
        s.add_instance(blob, angle_deg=17.0 * k, offset=((k % 20 - 9.5) * 1.1, 0.02 * (k % 7), (k // 20 - 4.5) * 1.1))
    s.add_sphere((0.0, 8.0, 0.0), 0.5, light)
    s.add_sphere_light((0.0, 8.0, 0.0), 0.5)
    s.camera = rrt.Camera(16.0 / 9.0, 50.0, scenes.look_at_c2w((1.0, 9.0, 14.0), (0.0, 0.0, 0.0)))
    s.background = (0.05, 0.06, 0.08)
    r = rrt.Renderer(seed=5)
    info = r.upload(s)
    assert info.unrolled_instances == 200
    w, h, spp = 160, 90, 32
    r.set_tuning(overlap_items=1, split_items=1, chunks=8, pixel_lists=0)
    plain, s0 = r.render(w, h, spp)
    for overlap in (0, 1):
        r.set_tuning(overlap_items=overlap, pixel_lists=1)
        img, s1 = r.render(w, h, spp)
        assert r.info().pixel_list_permille > 500
        assert s0.rays == s1.rays and np.array_equal(plain, img), overlap
    r.close()
