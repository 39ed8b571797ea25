"""The multi-GPU partition through the C ABI (rrt_group_*): one process, one context per listed device, sample slices
side by side, sum on member 0.  The GPU box has one GPU, so groups of several members share it (the copy-and-add form
of the sum); the RCCL form of the sum needs distinct GPUs and runs here as a group of ONE member that is told to go
through RCCL all the same (RRT_CFG_GROUP_FORCE_RCCL): communicator set-up, the grouped in-place ncclReduce to member 0
and its ordering behind the render on the member's stream all execute; what a one-GPU box cannot run is the transport
between GPUs.  Checked: the partition arithmetic, the sums, the stream / event ordering between members."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_two_members_on_one_gpu_add_their_slices(rrt):
    scene = helpers.scene_mixed(rrt)
    w, h = 120, 68
    single = rrt.Renderer(device=0, seed=5)
    single.upload(scene)
    lo, s_lo = single.render(w, h, 16, 0)
    hi, s_hi = single.render(w, h, 16, 16)
    whole, s_whole = single.render(w, h, 32, 0)
    g = rrt.RendererGroup([0, 0], seed=5)
    assert g.size == 2 and g.reduce_path == "copy+add"
    info = g.upload(scene)
    assert info.prim_count > 0
    for _ in range(2):  # twice: buffers and events are reused
        img, st = g.render(w, h, 32, 0)
        assert st.rays == s_lo.rays + s_hi.rays == s_whole.rays and st.samples == w * h * 32
        assert np.array_equal(img, lo + hi)  # the sum adds the two slices and nothing else
        assert np.allclose(img, whole, rtol=2e-5, atol=1e-5)  # same samples as one GPU, another summation order
    assert np.array_equal(g.tonemap(img, 32), single.tonemap(img, 32))
    g.close()
    single.close()


def test_uneven_split_idle_members_and_a_group_of_one(rrt):
    scene = helpers.scene_default(rrt)
    w, h = 71, 40
    single = rrt.Renderer(device=0, seed=2)
    single.upload(scene)
    g3 = rrt.RendererGroup([0, 0, 0], seed=2)
    g3.upload(scene)
    # 7 samples over 3 members: 3 + 2 + 2, offsets 5, 8, 10
    parts = [single.render(w, h, n, off)[0] for off, n in ((5, 3), (8, 2), (10, 2))]
    img, st = g3.render(w, h, 7, 5)
    assert np.array_equal(img, (parts[0] + parts[1]) + parts[2]) and st.samples == w * h * 7
    # 2 samples over 3 members: the third member has nothing to render and contributes zeros
    img2, st2 = g3.render(w, h, 2, 0)
    a, b = single.render(w, h, 1, 0)[0], single.render(w, h, 1, 1)[0]
    assert np.array_equal(img2, a + b) and st2.samples == w * h * 2
    g3.close()
    g1 = rrt.RendererGroup([0], seed=2)
    assert g1.reduce_path == "none"
    g1.upload(scene)
    one, s1 = g1.render(w, h, 7, 5)
    ref, sr = single.render(w, h, 7, 5)
    assert np.array_equal(one, ref) and s1.rays == sr.rays
    g1.close()
    single.close()


def test_a_group_of_one_can_be_made_to_sum_through_rccl(rrt):
    scene = helpers.scene_mixed(rrt)
    single = rrt.Renderer(device=0, seed=5)
    single.upload(scene)
    g = rrt.RendererGroup([0], seed=5, force_rccl=True)
    assert g.size == 1 and g.reduce_path == "rccl"
    g.upload(scene)
    # twice per frame size (communicator, buffers and streams are reused), the second frame larger (buffers regrow)
    for w, h, spp, off in ((120, 68, 16, 4), (120, 68, 16, 4), (333, 190, 24, 0), (333, 190, 24, 0)):
        ref, sr = single.render(w, h, spp, off)
        img, st = g.render(w, h, spp, off)
        assert st.rays == sr.rays and st.samples == w * h * spp
        assert np.array_equal(img, ref)  # the in-place reduce of one rank leaves the member's own image
    assert np.array_equal(g.tonemap(img, 24), single.tonemap(ref, 24))
    g.close()
    single.close()
    # the forced form needs distinct devices: RCCL refuses two ranks on one GPU
    with pytest.raises(rrt.RrtError) as e:
        rrt.RendererGroup([0, 0], force_rccl=True)
    assert "distinct" in str(e.value)


def test_group_errors(rrt):
    with pytest.raises(rrt.RrtError):
        rrt.RendererGroup([0, 99])
    g = rrt.RendererGroup([0, 0])
    with pytest.raises(rrt.RrtError) as e:
        g.render(8, 8, 4)
    assert "no scene uploaded" in str(e.value)
    g.close()
