"""Device shot sampler (vqa_sample_counts[_dev]) against the exact binomial / multinomial
laws: agreement with the reference's finite-shot mode is in distribution only (qulacs'
private generator cannot be reproduced), so these are statistical known-answer tests with
fixed seeds: conservation, determinism, moments, and chi-square against scipy's pmf over
both binomial algorithms (inversion below n*p = 10, transformed rejection above)."""
import numpy as np
import pytest
import scipy.stats as st

from vqa_amd import sample_counts, sample_z_means

pytestmark = pytest.mark.gpu


def test_counts_conserve_shots_and_are_deterministic():
    rng = np.random.default_rng(0)
    E, M = 5000, 4
    p = rng.dirichlet(np.ones(M), E)
    n = rng.integers(0, 5000, E)
    a = sample_counts(p, n, seed=7)
    b = sample_counts(p, n, seed=7)
    c = sample_counts(p, n, seed=8)
    d = sample_counts(p, n, seed=7, stream_id=1)
    assert a.dtype == np.int64 and a.shape == (E, M)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, c) and not np.array_equal(a, d)
    assert np.array_equal(a.sum(axis=1), n) and a.min() >= 0
    # a row's draw depends on the row number, not on the batch around it
    assert np.array_equal(sample_counts(p[:100], n[:100], seed=7), a[:100])


def test_degenerate_rows():
    p = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.5, 0.5, 0.0], [0.2, 0.3, 0.5], [np.nan, 0.5, 0.5]])
    n = np.array([10, 10, 11, 0, 6])
    k = sample_counts(p, n, seed=1)
    assert list(k[0]) == [10, 0, 0] and list(k[1]) == [0, 0, 10]
    assert k[2, 2] == 0 and k[2].sum() == 11
    assert list(k[3]) == [0, 0, 0]
    assert k[4, 0] == 0 and k[4].sum() == 6
    assert sample_counts(np.zeros((0, 2)), np.zeros(0, np.int64), seed=1).shape == (0, 2)


@pytest.mark.parametrize("n,p", [(1, 0.3), (7, 0.5), (40, 0.1), (40, 0.93), (1000, 0.004), (30, 0.4), (1000, 0.3),
                                 (100000, 0.5), (5000, 0.998), (2_000_000, 0.01)])
def test_binomial_law(n, p):
    """Chi-square of 200000 draws against the exact pmf (bins merged to expected counts >= 20)."""
    R = 200_000
    k = sample_counts(np.tile([p, 1 - p], (R, 1)), np.full(R, n), seed=1234 + n)[:, 0]
    assert abs(k.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / R)
    assert abs(k.var() - n * p * (1 - p)) < 0.05 * n * p * (1 - p) + 1e-9
    lo, hi = int(max(0, np.floor(n * p - 8 * np.sqrt(n * p * (1 - p)) - 2))), int(min(n, np.ceil(n * p + 8 * np.sqrt(n * p * (1 - p)) + 2)))
    ks = np.arange(lo, hi + 1)
    expect = R * st.binom.pmf(ks, n, p)
    obs = np.bincount(np.clip(k, lo, hi) - lo, minlength=len(ks)).astype(float)
    # merge neighbouring bins until every expected count is at least 20
    eb, ob, ce, co = [], [], 0.0, 0.0
    for e_, o_ in zip(expect, obs):
        ce += e_; co += o_
        if ce >= 20:
            eb.append(ce); ob.append(co); ce = co = 0.0
    if eb:
        eb[-1] += ce; ob[-1] += co
    eb, ob = np.array(eb), np.array(ob)
    if len(eb) > 1:
        eb *= ob.sum() / eb.sum()
        chi2 = ((ob - eb) ** 2 / eb).sum()
        assert st.chi2.sf(chi2, len(eb) - 1) > 1e-4, (chi2, len(eb))


def test_multinomial_covariance_and_device_path():
    import torch

    p = np.array([0.1, 0.2, 0.3, 0.4])
    n, R = 500, 100_000
    k = sample_counts(torch.from_numpy(np.tile(p, (R, 1))).cuda(), torch.full((R,), n).cuda(), seed=5)
    assert k.is_cuda and k.dtype == torch.int64
    k = k.cpu().numpy().astype(float)
    assert np.all(k.sum(axis=1) == n)
    assert np.abs(k.mean(axis=0) - n * p).max() < 0.2
    cov = np.cov(k.T)
    want = n * (np.diag(p) - np.outer(p, p))
    assert np.abs(cov - want).max() < 0.03 * n * 0.25
    # the host-pointer path draws the same stream
    assert np.array_equal(sample_counts(np.tile(p, (50, 1)), np.full(50, n), seed=5), k[:50].astype(np.int64))


def test_sampled_z_means():
    e = np.linspace(-1, 1, 41)
    m = sample_z_means(np.tile(e, (20000, 1)), 64, seed=3)
    assert m.shape == (20000, 41) and np.abs(m).max() <= 1
    assert np.abs(m.mean(axis=0) - e).max() < 5 / np.sqrt(64 * 20000)
    assert np.abs(m.var(axis=0) - (1 - e ** 2) / 64).max() < 1e-3
    assert np.all(m[:, 0] == -1) and np.all(m[:, -1] == 1)
