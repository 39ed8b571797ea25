"""Known-answer tests for the parts of the CPU restatement that the reference's own files cannot pin (the reference has
no tests) but that have a closed form: the `gen_range` mapping of rand 0.8.3, the orthonormal basis glam 0.20 builds,
Schlick's reflectance and `refract` at their special angles, the cosine lobe, and the light pdfs against the analytic
solid angle.  Everything here runs on the CPU."""
import math

import numpy as np
import pytest


# ------------------------------------------------------------------------------------ rand 0.8.3 gen_range(f32)
def test_gen_range_follows_sample_single(oracle):
    """rand 0.8.3, src/distributions/uniform.rs, UniformFloat::sample_single (the routine `gen_range(low..high)` calls):
    value0_1 = (bits >> 9 as [1,2)) - 1;  res = value0_1 * (high - low) + low;  accept if res < high.  The other
    formula in that file, value1_2 * scale + (low - scale), belongs to Uniform::new(..).sample(), which the reference
    (src/vec3.rs:9-16, src/geometry/aarect.rs:135-138) never calls."""
    f = np.float32
    # endpoints of the 23-bit grid on (-1, 1): draw 0 -> low exactly; the largest draw -> high - 2^-22, never high
    v, ok = oracle.gen_range_map(0, -1.0, 1.0)
    assert v == f(-1.0) and ok
    v, ok = oracle.gen_range_map(0xFFFFFFFF, -1.0, 1.0)
    assert v == f(1.0) - f(2.0 ** -22) and ok
    v, ok = oracle.gen_range_map(0x80000000, -1.0, 1.0)
    assert v == f(0.0) and ok
    # only the top 23 bits matter
    assert oracle.gen_range_map(0x000001FF, -1.0, 1.0)[0] == f(-1.0)
    assert oracle.gen_range_map(0x00000200, -1.0, 1.0)[0] == f(-1.0) + f(2.0 ** -22)
    # a range whose top value rounds up to `high` (343 - 130 * 2^-23 is nearer to 343 than to the float below it):
    # the first try is refused, as rand refuses it
    v, ok = oracle.gen_range_map(0xFFFFFFFF, 213.0, 343.0)
    assert v == f(343.0) and not ok


def test_gen_range_distribution(oracle):
    x = np.concatenate([oracle.gen_range(3, p, 0, -1.0, 1.0, 4096) for p in range(64)])
    assert x.min() >= -1.0 and x.max() < 1.0
    n = len(x)
    assert abs(x.mean()) < 4.0 * (2.0 / math.sqrt(12.0)) / math.sqrt(n)
    assert abs(x.var() - 1.0 / 3.0) < 0.01
    # every value lies on the grid low + k * 2^-22
    k = (x.astype(np.float64) + 1.0) * 2.0 ** 22
    assert np.all(k == np.round(k))
    y = oracle.gen_range(3, 1, 0, 213.0, 343.0, 100000)
    assert y.min() >= 213.0 and y.max() < 343.0 and abs(y.mean() - 278.0) < 0.5


# ------------------------------------------------------------------------------------ glam any_orthonormal_pair
@pytest.mark.parametrize("n", [(0, 0, 1), (0, 0, -1), (0, 0, 5), (1, 0, 0), (0, -1, 0), (0.6, 0.0, 0.8), (1e-4, -1e-4, -1.0),
                               (0.3, -0.7, 0.2), (-2.0, 3.0, -6.0), (1e-20, 1e-20, 1e-10)])
def test_onb_is_orthonormal_and_right_handed(oracle, n):
    """vec3.rs:59-68 over glam 0.20 `any_orthonormal_pair` (Duff et al. 2017): u, v, w orthonormal, w = normalize(n),
    u x v = w -- also at w.z = +1 (sign = +1, a = -1/2) and w.z = -1 (sign = -1, a = +1/2), where the naive
    Frisvad form divides by zero."""
    m = oracle.onb(n).astype(np.float64)
    u, v, w = m
    nn = np.asarray(n, np.float64)
    assert np.allclose(w, nn / np.linalg.norm(nn), atol=2e-7)
    assert np.allclose(m @ m.T, np.eye(3), atol=5e-7)
    assert np.allclose(np.cross(u, v), w, atol=5e-7)


def test_onb_at_the_poles_has_the_published_values(oracle):
    # sign = copysign(1, w.z); a = -1/(sign + w.z); b = w.x w.y a; u = (1 + sign w.x^2 a, sign b, -sign w.x); v = (b, sign + w.y^2 a, -w.y)
    assert np.array_equal(oracle.onb((0, 0, 1)), np.array([[1, 0, -0.0], [0, 1, -0.0], [0, 0, 1]], np.float32))
    assert np.array_equal(oracle.onb((0, 0, -1)), np.array([[1, -0.0, 0], [0, -1, -0.0], [0, 0, -1]], np.float32))


# ------------------------------------------------------------------------------------ dielectric pieces
def test_schlick_reflectance_at_its_special_angles(oracle):
    """material/mod.rs:165-169: r0 = ((1 - ir)/(1 + ir))^2; reflectance(1) = r0 (normal incidence),
    reflectance(0) = 1 (grazing), and the same for ir and 1/ir up to rounding."""
    for ir in (1.5, 1.33, 2.4):
        r0 = ((1 - ir) / (1 + ir)) ** 2
        assert abs(oracle.reflectance(1.0, ir) - r0) < 1e-7
        assert oracle.reflectance(0.0, ir) == 1.0
        assert abs(oracle.reflectance(1.0, 1.0 / ir) - r0) < 1e-6
        mid = r0 + (1 - r0) * 0.5 ** 5
        assert abs(oracle.reflectance(0.5, ir) - mid) < 1e-7
    assert abs(oracle.reflectance(1.0, 1.5) - 0.04) < 1e-7


def test_refract_obeys_snell_and_the_total_reflection_boundary(oracle):
    """material/mod.rs:154-163: None iff ratio * sin(theta) > 1; otherwise the refracted unit vector."""
    n = np.array([0, 1, 0], np.float32)
    # normal incidence: straight through, whatever the ratio
    assert np.allclose(oracle.refract((0, -1, 0), n, 1 / 1.5), (0, -1, 0), atol=1e-7)
    # 45 degrees into glass: sin(theta_t) = sin(45) / 1.5
    uv = np.array([math.sin(math.pi / 4), -math.cos(math.pi / 4), 0], np.float32)
    out = oracle.refract(uv, n, 1 / 1.5)
    assert abs(np.linalg.norm(out) - 1.0) < 1e-6 and abs(out[0] - math.sin(math.pi / 4) / 1.5) < 1e-6 and out[1] < 0
    # leaving glass (ratio 1.5): the critical angle is asin(1/1.5) = 41.81 degrees
    crit = math.asin(1 / 1.5)
    for deg, passes in ((crit - 1e-3, True), (crit + 1e-3, False), (math.radians(80), False), (math.radians(10), True)):
        uv = np.array([math.sin(deg), -math.cos(deg), 0], np.float32)
        out = oracle.refract(uv, n, 1.5)
        assert (out is not None) == passes
        if passes:
            assert abs(out[0] - 1.5 * math.sin(deg)) < 2e-6  # Snell
    # reflect, material/mod.rs:150-152
    assert np.allclose(oracle.reflect((1, -1, 0), n), (1, 1, 0))


# ------------------------------------------------------------------------------------ pdfs
def test_cosine_lobe_samples_and_pdf(oracle):
    """pdf.rs:14-26,41-48: directions are unit vectors in the hemisphere of the normal, distributed as cos/pi
    (E[cos] = 2/3), and the pdf value is cos/pi."""
    nrm = np.array([0.3, -0.5, 0.8], np.float32)
    dirs, pdfs = oracle.cosine_samples(nrm, 5, 200000)
    w = nrm / np.linalg.norm(nrm)
    cos = dirs.astype(np.float64) @ w
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1.0, atol=1e-5)
    assert cos.min() >= -1e-6
    assert np.allclose(pdfs, np.maximum(cos, 0) / math.pi, atol=1e-6)
    assert abs(cos.mean() - 2.0 / 3.0) < 3e-3
    # Monte-Carlo check of the normalisation: E[1/pdf] over the lobe = area of the hemisphere
    assert abs(np.mean(1.0 / np.maximum(pdfs[cos > 0.05], 1e-9) * (cos[cos > 0.05] > 0)) * (cos > 0.05).mean() - 2 * math.pi * (1 - 0.05)) < 0.05


def _light_scene(rrt, kind):
    s = rrt.Scene()
    m = s.add_material(rrt.DiffuseLight(1, 1, 1))
    if kind == "sphere":
        s.add_sphere((0, 5, 0), 1.0, m)
        s.add_sphere_light((0, 5, 0), 1.0)
    else:
        s.add_rect(1, -1.0, 1.0, -2.0, 2.0, 4.0, m)
        s.add_xzrect_light(-1.0, 1.0, -2.0, 2.0, 4.0)
    return s


def test_sphere_light_pdf_is_one_over_the_cone_solid_angle(rrt, oracle):
    """sphere.rs:101-112: 1 / (2 pi (1 - sqrt(1 - r^2/d^2))) for directions that meet the sphere, 0 otherwise; and
    sphere.rs:114-119,217-228: sampled directions all meet it and integrate the pdf to one."""
    o = oracle.OracleScene(_light_scene(rrt, "sphere"))
    origin = (0, 0, 0)
    omega = 2 * math.pi * (1 - math.sqrt(1 - 1.0 / 25.0))
    assert abs(oracle.lights_pdf(o, origin, (0, 1, 0)) - 1 / omega) < 1e-4 / omega
    assert abs(oracle.lights_pdf(o, origin, (0, 3, 0)) - 1 / omega) < 1e-4 / omega  # length of v does not matter
    assert oracle.lights_pdf(o, origin, (1, 0, 0)) == 0.0
    assert oracle.lights_pdf(o, origin, (0, -1, 0)) == 0.0
    edge = math.asin(1 / 5)
    assert oracle.lights_pdf(o, origin, (math.sin(edge * 0.99), math.cos(edge * 0.99), 0)) > 0
    assert oracle.lights_pdf(o, origin, (math.sin(edge * 1.01), math.cos(edge * 1.01), 0)) == 0.0
    d = oracle.lights_random(o, origin, 7, 20000).astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    assert np.all(d[:, 1] >= math.cos(edge) - 1e-6)  # inside the cone
    # uniform in the cone: mean cos = (1 + cos_max) / 2
    assert abs(d[:, 1].mean() - (1 + math.cos(edge)) / 2) < 1e-4


def test_xz_rect_light_pdf_matches_the_analytic_solid_angle(rrt, oracle):
    """aarect.rs:123-132: pdf = distance^2 / (cos * area) is the density, per solid angle, of a point chosen uniformly on
    the rectangle (aarect.rs:134-142).  Its reciprocal integrated over the rectangle's directions is the rectangle's
    solid angle; for a point on the axis of an a x b rectangle at height h that is
    4 * atan(a b / (2 h sqrt(4 h^2 + a^2 + b^2)))."""
    o = oracle.OracleScene(_light_scene(rrt, "rect"))
    origin = (0.0, 0.0, 0.0)
    a, b, h = 2.0, 4.0, 4.0
    omega = 4 * math.atan(a * b / (2 * h * math.sqrt(4 * h * h + a * a + b * b)))
    # straight up: distance h, cos 1 -> h^2 / area
    assert abs(oracle.lights_pdf(o, origin, (0, 1, 0)) - h * h / (a * b)) < 1e-6
    assert abs(oracle.lights_pdf(o, origin, (0, 2.5, 0)) - h * h / (a * b)) < 1e-6
    assert oracle.lights_pdf(o, origin, (0, -1, 0)) == 0.0 and oracle.lights_pdf(o, origin, (1, 0.1, 0)) == 0.0
    # importance-sampled estimate of the solid angle: E[1/pdf] over directions drawn by XZRect::random
    d = oracle.lights_random(o, origin, 11, 20000)
    inv = np.array([1.0 / oracle.lights_pdf(o, origin, v) for v in d[:4000]])
    assert abs(inv.mean() - omega) < 0.01 * omega
    # the sampled points cover the rectangle uniformly
    pts = d.astype(np.float64)  # origin is 0: direction = point on the light
    assert np.allclose(pts[:, 1], h) and abs(pts[:, 0].mean()) < 0.02 and abs(pts[:, 2].mean()) < 0.04
    assert pts[:, 0].min() >= -1 and pts[:, 0].max() < 1 and pts[:, 2].min() >= -2 and pts[:, 2].max() < 2
