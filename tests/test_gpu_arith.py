"""The kernels form 1/x and sqrt(x) in the primitive tests and the light pdf with a short instruction sequence
(hardware estimate + one Newton step in fused arithmetic; IEEE sequence for operands near the ends of the exponent
range).  It must round exactly like the IEEE division and square root of the reference's f32 arithmetic
(src/geometry/triangle.rs:46 `1.0 / det`, src/geometry/sphere.rs:61 `discriminant.sqrt()`): the library runs both
over every one of the 2^32 float bit patterns on the device and counts the patterns whose results differ."""
import pytest

pytestmark = pytest.mark.gpu


def test_short_reciprocal_and_sqrt_equal_the_ieee_operations_on_every_float(rrt):
    r = rrt.Renderer(seed=1)
    recip_bad, sqrt_bad = r.selftest_arith()
    r.close()
    assert recip_bad == 0
    assert sqrt_bad == 0
