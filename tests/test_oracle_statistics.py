"""The reference's live tests are chi-squared tests of its samplers against their pdfs
(statistics/spherical_sampling.rs:64-226, statistics/bxdfs/*.rs tests, distributions.rs:186-300).
They are restated here on the oracle: sample directions, bin them on the sphere, integrate the pdf
over the bins, chi-squared; plus "the pdf integrates to 1" (spherical_sampling.rs:96-99)."""
import numpy as np
import pytest
from scipy import stats

import scenes

N_COS, N_PHI = 24, 48  # equal-area bins in (cos theta, phi)


def bin_directions(d):
    d = d.astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ci = np.clip(((d[:, 2] + 1) / 2 * N_COS).astype(int), 0, N_COS - 1)
    phi = np.arctan2(d[:, 1], d[:, 0]) % (2 * np.pi)
    pi_ = np.clip((phi / (2 * np.pi) * N_PHI).astype(int), 0, N_PHI - 1)
    return np.bincount(ci * N_PHI + pi_, minlength=N_COS * N_PHI)


def integrate_pdf(pdf_fn, sub=12):
    """expected probability per bin by midpoint quadrature with sub x sub points per bin"""
    cz = (np.arange(N_COS * sub) + 0.5) / (N_COS * sub) * 2 - 1
    ph = (np.arange(N_PHI * sub) + 0.5) / (N_PHI * sub) * 2 * np.pi
    CZ, PH = np.meshgrid(cz, ph, indexing="ij")
    s = np.sqrt(np.maximum(0, 1 - CZ ** 2))
    dirs = np.stack([s * np.cos(PH), s * np.sin(PH), CZ], axis=-1).reshape(-1, 3).astype(np.float32)
    p = pdf_fn(dirs).astype(np.float64).reshape(N_COS, sub, N_PHI, sub)
    p = np.nan_to_num(p, nan=0.0, posinf=0.0)
    cell = 4 * np.pi / (N_COS * N_PHI * sub * sub)
    return (p.sum(axis=(1, 3)) * cell).reshape(-1)


def chi2_ok(counts, probs, n):
    exp = probs * n
    order = np.argsort(exp)
    e, c = exp[order], counts[order].astype(np.float64)
    # pool the small-expectation bins (chi_squared.rs pools cells under 5)
    cut = np.searchsorted(np.cumsum(e), 5.0) + 1
    e = np.concatenate([[e[:cut].sum()], e[cut:]])
    c = np.concatenate([[c[:cut].sum()], c[cut:]])
    e *= c.sum() / e.sum()
    stat = ((c - e) ** 2 / np.maximum(e, 1e-12)).sum()
    return stats.chi2.sf(stat, len(e) - 1)


def run_case(sample_fn, pdf_fn, n=300000, integrates_to=1.0, tol=2e-2, sub=12):
    probs = integrate_pdf(pdf_fn, sub)
    assert abs(probs.sum() - integrates_to) < tol, probs.sum()
    counts = bin_directions(sample_fn(n))
    p = chi2_ok(counts, probs / probs.sum(), n)
    assert p > 1e-3, p


def test_lambertian_local(O):  # bxdfs/lambertian.rs:30-37
    run_case(lambda n: O.sample_directions_noscene(0, n, seed=11),
             lambda d: O.eval_pdfs_noscene(0, d))


def test_lambertian_non_local(O):  # bxdfs/lambertian.rs:39-48
    nrm = np.array([0.3, -0.5, 0.81]); nrm /= np.linalg.norm(nrm)
    run_case(lambda n: O.sample_directions_noscene(0, n, seed=12, normal=nrm),
             lambda d: O.eval_pdfs_noscene(0, d, normal=nrm))


@pytest.mark.parametrize("alpha", [0.1, 0.3, 0.7])
def test_trowbridge_reitz_vndf(O, alpha):  # trowbridge_reitz_vndf.rs:150-219
    inc = np.array([0.4, 0.2, 0.89]); inc /= np.linalg.norm(inc)
    run_case(lambda n: O.sample_directions_noscene(1, n, seed=13, incoming=inc, alpha=alpha),
             lambda d: O.eval_pdfs_noscene(1, d, incoming=inc, alpha=alpha))


@pytest.mark.parametrize("alpha", [0.1, 0.3, 0.7])
def test_trowbridge_reitz_vndf_half_vectors(O, alpha):  # `isotropic_h`, trowbridge_reitz_vndf.rs:157-165
    """isotropic::vndf(alpha, h, incoming) as the density of the half vectors isotropic::sample_vndf draws
    (local frame; incoming = -generate_wi(): a direction of the upper hemisphere, spherical_sampling.rs:237-242)"""
    inc = np.array([-0.35, 0.45, 0.82]); inc /= np.linalg.norm(inc)
    # (the density of half vectors is D-shaped: at alpha 0.1 its peak is narrower than a 12 x 12 quadrature cell)
    run_case(lambda n: O.sample_directions_noscene(5, n, seed=19, incoming=inc, alpha=alpha),
             lambda d: O.eval_pdfs_noscene(5, d, incoming=inc, alpha=alpha), sub=96 if alpha < 0.2 else 24)


def test_trowbridge_reitz_vndf_rotated_normal(O):
    nrm = np.array([-0.6, 0.2, 0.5]); nrm /= np.linalg.norm(nrm)
    inc = nrm * 0.8 + np.array([0.3, 0.3, 0.0]); inc /= np.linalg.norm(inc)
    run_case(lambda n: O.sample_directions_noscene(1, n, seed=14, incoming=inc, normal=nrm, alpha=0.25),
             lambda d: O.eval_pdfs_noscene(1, d, incoming=inc, normal=nrm, alpha=0.25))


@pytest.mark.parametrize("res", [(60, 30), (100, 100), (7, 13)])
def test_sky_sampling_matches_sky_pdf(O, res):
    """the test the reference left as todo!() (sky.rs:104-115): Lerp sky, sample vs pdf"""
    sc = scenes.SceneDescription()
    sc.sphere((0, 0, -50), 1.0, sc.lambertian(sc.solid(0.5), 0.5))
    sc.set_sky(sc.lerp((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), res)
    s = O.Scene(sc)
    run_case(lambda n: s.sample_directions(2, n, seed=15), lambda d: s.eval_pdfs(2, d))


def test_random_unit_vector_is_uniform(O):
    run_case(lambda n: O.sample_directions_noscene(3, n, seed=16),
             lambda d: np.full(len(d), 1 / (4 * np.pi), dtype=np.float32))


def test_sphere_light_sampling_is_uniform_in_its_cone(O):
    """Sphere::sample_visible_from_point (sphere.rs:124-154) against Sphere::scattering_pdf (:155-166)"""
    sc = scenes.SceneDescription()
    light = sc.emissive(sc.solid(1.0), 1.0)
    sc.sphere((0.0, 0.0, 4.0), 1.5, light)
    sc.set_sky(sc.solid(0.0), (0, 0))
    s = O.Scene(sc)
    cos_max = np.sqrt(1 - (1.5 / 4.0) ** 2)
    pdf = 1 / (2 * np.pi * (1 - cos_max))
    run_case(lambda n: s.sample_directions(4, n, seed=17, incoming=(0, 0, 0), prim_index=0),
             lambda d: np.where(d[:, 2] / np.linalg.norm(d, axis=1) >= cos_max, pdf, 0).astype(np.float32),
             tol=6e-2)  # the pdf is a step: the midpoint quadrature is coarse at the cone's rim


def test_mesh_triangle_sampling_hits_the_triangle(O):
    sc = scenes.SceneDescription()
    light = sc.emissive(sc.solid(1.0), 1.0)
    n = (0, 0, -1)
    sc.triangle([(-1, -1, 3), (1, -1, 3), (0, 1, 3)], [n, n, n], light)
    sc.set_sky(sc.solid(0.0), (0, 0))
    s = O.Scene(sc)
    d = s.sample_directions(4, 20000, seed=18, incoming=(0, 0, 0), prim_index=0).astype(np.float64)
    p = d * (3.0 / d[:, 2:3])  # intersect with the plane z = 3
    # barycentric test
    a, b, c = np.array([-1, -1]), np.array([1, -1]), np.array([0, 1])
    def side(p, q, r): return (q[0] - p[0]) * (r[:, 1] - p[1]) - (q[1] - p[1]) * (r[:, 0] - p[0])
    inside = (side(a, b, p) >= -1e-5) & (side(b, c, p) >= -1e-5) & (side(c, a, p) >= -1e-5)
    assert inside.all()
    # Triangle (not MeshTriangle) sampling is area-uniform: centroid of the samples = centroid of the triangle
    assert np.abs(p[:, :2].mean(axis=0) - np.array([0, -1 / 3])).max() < 0.02


@pytest.mark.parametrize("shape", [(1, 64), (800, 1), (40, 60)])
def test_distribution_1d_sampling(O, shape):  # distributions.rs:186-300 (1-D cases)
    rng = np.random.default_rng(shape[0])
    vals = rng.uniform(0, 1, shape[0] * shape[1]).astype(np.float32) ** 3
    idx, pdf, cdf = O.dist1d(vals, 400000, seed=19)
    counts = np.bincount(idx, minlength=vals.size)
    assert chi2_ok(counts, pdf.astype(np.float64) / pdf.sum(), 400000) > 1e-3


# ---- Distribution2D: statistics/distributions.rs:206-300 (random_2d_small / _medium / _large) ----
# The reference fills an x-by-y table with uniform values in [0, 100), builds Distribution2D::new(values, x), draws
# 100 000 samples per batch (64 batches averaged, 10 repeats, Sidak-corrected p >= 0.01) and chi-squares the histogram
# of (x, y) cells against y_distribution.pdf[y] * x_distributions[y].pdf[x].  Restated on the oracle's Distribution2D
# with one large draw per case (the 64-batch average is a variance reduction of the same histogram).
@pytest.mark.parametrize("x_res,y_res,n", [(3, 3, 400000),        # random_2d_small   :287-290
                                            (30, 60, 2000000),     # random_2d_medium  :292-295
                                            (800, 1200, 12000000)])  # random_2d_large   :297-300
def test_distribution_2d_sampling(O, x_res, y_res, n):
    rng = np.random.default_rng(x_res * 7919 + y_res)
    values = rng.uniform(0.0, 100.0, x_res * y_res).astype(np.float32)
    x, y, pdf = O.dist2d(values, x_res, n, seed=23)
    assert x.max() < x_res and y.max() < y_res  # the reference's `unreachable!()` arm (:226-231)
    assert abs(float(pdf.astype(np.float64).sum()) - 1.0) < 1e-3
    # the discrete pdf is the normalised table (Distribution1D::new of rows and of row sums, :12-44, :83-99)
    table = values.astype(np.float64).reshape(y_res, x_res)
    assert np.abs(pdf.astype(np.float64) - table / table.sum()).max() < 2e-6 / (x_res * y_res) ** 0.5 + 1e-9
    counts = np.bincount(y.astype(np.int64) * x_res + x, minlength=x_res * y_res)
    p = chi2_ok(counts, pdf.astype(np.float64).reshape(-1) / pdf.astype(np.float64).sum(), n)
    assert p > 1e-3, p


# ---- GGX identities: statistics/bxdfs/trowbridge_reitz.rs:128-230 ----
# The reference integrates with nested adaptive Simpson over an 80 x 160 (theta, phi) grid (integrate_over_sphere,
# spherical_sampling.rs:39-62) and draws alpha and the directions from thread_rng; here the integrands are built from
# the ORACLE's d / g1 / g2 (f32, the functions the shading code calls) at fixed alphas and directions and integrated in
# float64 with Gauss-Legendre in cos(theta) x uniform phi.  Tolerance 1e-4, the reference's own (measured residuals:
# 1e-8 .. 8e-5, the largest on the weak furnace).
TR_TOL = 1e-4
_GL_X, _GL_W = np.polynomial.legendre.leggauss(600)
_PHI = (np.arange(1200) + 0.5) / 1200 * 2 * np.pi


def sphere_quadrature():
    cz, ph = np.meshgrid(_GL_X, _PHI, indexing="ij")
    s = np.sqrt(np.maximum(0.0, 1.0 - cz ** 2))
    dirs = np.stack([s * np.cos(ph), s * np.sin(ph), cz], axis=-1).reshape(-1, 3)
    w = np.repeat(_GL_W, len(_PHI)) * (2 * np.pi / len(_PHI))
    return dirs, w


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def wi_from(cos_theta, phi):  # -generate_wi (spherical_sampling.rs:237-242): a direction in the upper hemisphere
    s = np.sqrt(1 - cos_theta ** 2)
    return np.array([s * np.cos(phi), s * np.sin(phi), cos_theta])


Z = np.array([0.0, 0.0, 1.0])
TR_CASES = [(0.15, 0.35, 1.0), (0.4, 0.8, 4.0), (0.8, 0.6, 2.5)]  # (alpha, cos theta of the fixed direction, its phi)


@pytest.mark.parametrize("alpha,cos_theta,phi", TR_CASES)
def test_tr_g1_cos(O, alpha, cos_theta, phi):  # g1_cos_test :128-143: int g1(h, wi) max(wi.h, 0) d(h.z) dh = cos theta_i
    incoming = wi_from(cos_theta, phi)
    h, w = sphere_quadrature()
    f = O.tr_g1(alpha, Z, h, incoming).astype(np.float64) * np.maximum(h @ incoming, 0.0) * O.tr_d(alpha, h[:, 2]).astype(np.float64)
    assert abs((f * w).sum() - cos_theta) < TR_TOL


@pytest.mark.parametrize("alpha", [0.15, 0.4, 0.8])
def test_tr_projected_area_local(O, alpha):  # projected_area_test_local :145-152: int d(h.z) h.z dh = 1
    h, w = sphere_quadrature()
    f = O.tr_d(alpha, h[:, 2]).astype(np.float64) * h[:, 2]
    assert abs((f * w).sum() - 1.0) < TR_TOL


@pytest.mark.parametrize("alpha", [0.15, 0.4, 0.8])
def test_tr_projected_area_non_local(O, alpha):  # projected_area_test_non_local :154-162
    normal = unit([0.3, -0.5, 0.81])
    h, w = sphere_quadrature()
    c = h @ normal
    f = O.tr_d(alpha, c).astype(np.float64) * c
    assert abs((f * w).sum() - 1.0) < TR_TOL


def _h_upper(a, b, normal):  # (a + b).normalised(), flipped to the normal's side (:170-173, :193-196, :216-219)
    h = a + b
    h /= np.maximum(np.linalg.norm(h, axis=1, keepdims=True), 1e-300)
    flip = (h @ normal) < 0.0
    h[flip] = -h[flip]
    return h


@pytest.mark.parametrize("alpha,cos_theta,phi", TR_CASES)
def test_tr_weak_furnace(O, alpha, cos_theta, phi):  # weak_furnace_test :164-185: int g1(h, wo) d(h.z) / (4 |wo.z|) dwi = 1
    wo = wi_from(cos_theta, phi)
    wi, w = sphere_quadrature()
    h = _h_upper(wi, wo, Z)
    f = O.tr_g1(alpha, Z, h, wo).astype(np.float64) * O.tr_d(alpha, h[:, 2]).astype(np.float64) / (4.0 * abs(wo[2]))
    assert abs((f * w).sum() - 1.0) < TR_TOL


@pytest.mark.parametrize("alpha,cos_theta,phi", TR_CASES)
def test_tr_g2_bounded(O, alpha, cos_theta, phi):  # g2_test :187-207: int g2(h, a, b) d(h.z) / (4 |a.z|) db <= 1
    a = wi_from(cos_theta, phi)
    b, w = sphere_quadrature()
    h = _h_upper(a, b, Z)
    f = O.tr_g2(alpha, Z, h, a, b).astype(np.float64) * O.tr_d(alpha, h[:, 2]).astype(np.float64) / (4.0 * abs(a[2]))
    integral = (f * w).sum()
    assert 0.0 < integral <= 1.0 + TR_TOL


@pytest.mark.parametrize("normal", [(-0.6, 0.2, 0.5), (0.6, -0.2, -0.7)])  # a.(-normal) < 0 (d = 0) and > 0
@pytest.mark.parametrize("alpha,cos_theta,phi", TR_CASES)
def test_tr_g2_bounded_non_local(O, alpha, cos_theta, phi, normal):  # g2_test_non_local :209-230 (one-sided, as there)
    a = wi_from(cos_theta, phi)
    normal = unit(normal)
    b, w = sphere_quadrature()
    h = _h_upper(a, b, normal)
    denom = 4.0 * abs(a @ (-normal))
    f = O.tr_g2(alpha, normal, h, a, b).astype(np.float64) * float(O.tr_d(alpha, [a @ (-normal)])[0]) / denom
    integral = (f * w).sum() if denom >= 1e-9 else 0.0
    assert integral <= 1.0 + TR_TOL
