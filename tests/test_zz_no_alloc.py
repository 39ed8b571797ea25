"""The render entry point that allocates nothing (rrt_render_device_noalloc) against rrt_render_device, through the
C ABI: same kernels, counters / work-queue heads / partial images in the library's static device workspace instead of
buffers the context grows.  Bit-identical images and equal ray counts, on two calls in a row, the second with a
smaller input (the workspace keeps the first call's contents); frames whose partial images exceed the workspace are
cut into fewer sample chunks, and refused when not even two partial images fit."""
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


def test_noalloc_render_cuts_fewer_chunks_to_fit_and_refuses_what_cannot_fit(rrt):
    r = rrt.Renderer(device=0, seed=6)
    r.upload(helpers.scene_default(rrt))
    w, h, spp = 1920, 1080, 64  # the library's 4 chunks x 24.9 MB of partial images > 64 MiB: two chunks fit
    a, b = helpers.DeviceArray((h, w, 3)), helpers.DeviceArray((h, w, 3))
    r.render_device_noalloc(w, h, spp, 0, a.ptr, 0)
    sa = r.render_stats()
    assert sa.chunks == 2 and sa.samples == w * h * spp
    r.render_device(w, h, spp, 0, b.ptr, 0)
    assert r.render_stats().chunks == 4
    r.set_tuning(chunks=2)  # the allocating entry point with the same chunks: the same image bit for bit
    r.render_device(w, h, spp, 0, b.ptr, 0)
    sb = r.render_stats()
    assert sb.chunks == 2 and sb.rays == sa.rays and np.array_equal(a.numpy(), b.numpy())
    r.set_tuning(chunks=None)
    a.free()
    b.free()
    # a frame of which not even two partial images fit is refused ...
    w, h = 3840, 2160
    buf = helpers.DeviceArray((h, w, 3))
    with pytest.raises(rrt.RrtError) as e:
        r.render_device_noalloc(w, h, spp, 0, buf.ptr, 0)
    assert e.value.status == 1 and "static workspace" in str(e.value)
    # ... unless it is rendered as one chunk (few samples), which needs no partial image
    r.render_device_noalloc(w, h, 2, 0, buf.ptr, 0)
    st = r.render_stats()
    assert st.chunks == 1 and st.samples == w * h * 2 and np.all(np.isfinite(buf.numpy()))
    buf.free()
    r.close()
