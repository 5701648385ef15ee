"""Host-side evaluation metrics over the embeddings `MotionEncoder` produces.
Same function names, arguments and results as the reference's codes/utils/metrics.py; they run on
a few thousand 512-vectors once per evaluation, so they stay numpy/scipy on the host (the device
work of the evaluation is the sampling loop and the classifier forward).

The two sampling metrics draw their index pairs from numpy's GLOBAL generator with the same two
`np.random.choice(n, times, replace=False)` calls, in the same order, as the reference
(metrics.py:78-79, 91-92), so a seeded evaluation reproduces the reference's numbers.
"""
import numpy as np
from scipy import linalg


def euclidean_distance_matrix(matrix1, matrix2):
    """(N1, D), (N2, D) -> (N1, N2) pairwise distances via |a|^2 - 2ab + |b|^2 (metrics.py:6-20: the
    expansion, not a difference, so tiny negative arguments of sqrt give NaN exactly as there)."""
    assert matrix1.shape[1] == matrix2.shape[1]
    cross = -2 * np.dot(matrix1, matrix2.T)
    sq1 = np.sum(np.square(matrix1), axis=1, keepdims=True)
    sq2 = np.sum(np.square(matrix2), axis=1)
    return np.sqrt(cross + sq1 + sq2)


def calculate_top_k(mat, top_k):
    """mat (N, N) = per-row ranking of column ids; -> (N, top_k) bool, [i, k] = i is among row i's first k+1."""
    n = mat.shape[0]
    hit = mat[:, :top_k] == np.arange(n)[:, None]
    return np.logical_or.accumulate(hit, axis=1)


def calculate_R_precision(embedding1, embedding2, top_k, sum_all=False):
    order = np.argsort(euclidean_distance_matrix(embedding1, embedding2), axis=1)
    top = calculate_top_k(order, top_k)
    return top.sum(axis=0) if sum_all else top


def calculate_matching_score(embedding1, embedding2, sum_all=False):
    assert embedding1.ndim == 2 and embedding1.shape == embedding2.shape
    dist = linalg.norm(embedding1 - embedding2, axis=1)
    return dist.sum(axis=0) if sum_all else dist


def calculate_activation_statistics(activations):
    """(num_samples, dim) -> mean (dim,), covariance (dim, dim) (unbiased, like np.cov)."""
    return np.mean(activations, axis=0), np.cov(activations, rowvar=False)


def _two_draws(population, times):
    # the reference draws twice from numpy's global generator, first set then second set
    return [np.random.choice(population, times, replace=False) for _ in range(2)]


def calculate_diversity(activation, diversity_times):
    """Mean distance between `diversity_times` random pairs of embeddings (metrics.py:73-81)."""
    assert activation.ndim == 2 and activation.shape[0] > diversity_times
    i, j = _two_draws(activation.shape[0], diversity_times)
    return linalg.norm(activation[i] - activation[j], axis=1).mean()


def calculate_multimodality(activation, multimodality_times):
    """activation (captions, repeats, dim): mean distance between random pairs of repeats (metrics.py:84-93)."""
    assert activation.ndim == 3 and activation.shape[1] > multimodality_times
    i, j = _two_draws(activation.shape[1], multimodality_times)
    return linalg.norm(activation[:, i] - activation[:, j], axis=2).mean()


def _sqrt_of_product(c1, c2, eps):
    root = linalg.sqrtm(c1 @ c2)
    if not np.all(np.isfinite(root)):
        print("fid calculation produces singular product; adding %s to diagonal of cov estimates" % eps)
        ridge = eps * np.eye(len(c1))
        root = linalg.sqrtm((c1 + ridge) @ (c2 + ridge))
    if np.iscomplexobj(root):
        worst = np.abs(np.diagonal(root).imag)
        if not np.allclose(worst, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.abs(root.imag).max()))
        root = root.real
    return root


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """Frechet distance between N(mu1, sigma1) and N(mu2, sigma2):
    |mu1 - mu2|^2 + Tr(sigma1 + sigma2 - 2 (sigma1 sigma2)^(1/2))   (metrics.py:96-148, incl. its
    eps-regularised retry on a singular product and its tolerance for an imaginary residue)."""
    m1, m2 = (np.atleast_1d(m) for m in (mu1, mu2))
    c1, c2 = (np.atleast_2d(c) for c in (sigma1, sigma2))
    if m1.shape != m2.shape or c1.shape != c2.shape:
        raise AssertionError("mean vectors / covariances of the two sets differ in size")
    gap = m1 - m2
    return gap @ gap + np.trace(c1) + np.trace(c2) - 2 * np.trace(_sqrt_of_product(c1, c2, eps))
