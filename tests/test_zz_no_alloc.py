"""The render entry point that allocates nothing (rrt_render_device_noalloc) against rrt_render_device, through the
C ABI: same kernels, counters / work-queue heads / partial images in the library's static device workspace instead of
buffers the context grows.  Bit-identical images and equal ray counts, on two calls in a row, the second with a
smaller input (the workspace keeps the first call's contents); frames beyond the workspace are refused."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _both(r, w, h, spp, off):
    a, b = helpers.DeviceArray((h, w, 3)), helpers.DeviceArray((h, w, 3))  # both start as NaN everywhere
    r.render_device(w, h, spp, off, a.ptr, 0)
    sa = r.render_stats()
    r.render_device_noalloc(w, h, spp, off, b.ptr, 0)
    sb = r.render_stats()
    ia, ib = a.numpy(), b.numpy()
    a.free()
    b.free()
    return ia, sa, ib, sb


@pytest.mark.parametrize("name", ["mixed", "synthetic", "instanced"])
def test_noalloc_render_equals_the_allocating_one(rrt, name):
    r = rrt.Renderer(device=0, seed=6)
    r.upload(helpers.SCENES[name](rrt))
    # the second call is smaller in every respect: fewer pixels, fewer samples, hence fewer chunks
    for w, h, spp, off in ((200, 112, 50, 3), (71, 40, 9, 0)):
        a, sa, b, sb = _both(r, w, h, spp, off)
        assert np.all(np.isfinite(a)) and sa.rays > 0
        assert sa.rays == sb.rays and sa.samples == sb.samples == w * h * spp
        assert np.array_equal(a, b), (name, w, h, spp)
    r.close()


def test_noalloc_render_refuses_frames_beyond_the_static_workspace(rrt):
    r = rrt.Renderer(device=0, seed=6)
    r.upload(helpers.scene_default(rrt))
    w, h, spp = 1920, 1080, 64  # 4 chunks x 24.9 MB of partial images > 64 MiB
    buf = helpers.DeviceArray((h, w, 3))
    with pytest.raises(rrt.RrtError) as e:
        r.render_device_noalloc(w, h, spp, 0, buf.ptr, 0)
    assert e.value.status == 1 and "static workspace" in str(e.value)
    # the same frame at one chunk (few samples) needs no partial images and is accepted
    r.render_device_noalloc(w, h, 2, 0, buf.ptr, 0)
    st = r.render_stats()
    assert st.samples == w * h * 2 and np.all(np.isfinite(buf.numpy()))
    buf.free()
    r.close()
