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


def run_case(sample_fn, pdf_fn, n=300000, integrates_to=1.0, tol=2e-2):
    probs = integrate_pdf(pdf_fn)
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
