"""CPU restatement of the reference's greedy conditional-variance inducing-point selector.  TEST INFRASTRUCTURE ONLY.

Follows src/inducing_point_selectors/conditional_variance.py:27-120 line by line (numpy), with the kernel as a callable
k(x1, x2) -> ndarray (the reference calls a gpytorch kernel, which is not installed here).  One deliberate change: the
kernel diagonal is evaluated directly instead of through the full N x N Gram (:64-69) -- same numbers, no O(N^2) memory.
Parity of the *indices* is exact on inputs without ties in the residual variances; the reference's own tie-breaking is
the order numpy's argsort happens to produce, which no independent implementation can be held to."""
import numpy as np


def rbf_ard(lengthscale, outputscale):
    ls = np.asarray(lengthscale, dtype=np.float64).reshape(-1)

    def k(x1, x2):
        a, b = x1 / ls, x2 / ls
        d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        return outputscale * np.exp(-0.5 * d2)

    return k


def conditional_variance_select(x, m, kernel, jitter=1e-12, threshold=0.0, perm=None):
    assert m > 1, "Must have at least 2 inducing points"
    n = x.shape[0]
    if perm is None:
        perm = np.random.permutation(n)  # :58-60
    x = x[perm, ...]
    indices = np.zeros(m, dtype=int) + n  # :63
    di = np.array([kernel(x[i : i + 1], x[i : i + 1])[0, 0] for i in range(n)]) + jitter  # :64-69 (diagonal only)
    indices[0] = np.argmax(di)  # :70
    ci = np.zeros((m - 1, n))
    count = 1
    for i in range(m - 1):
        j = int(indices[i])
        dj = np.sqrt(di[j])
        cj = ci[:i, j]
        gram_matrix = np.round(np.squeeze(kernel(x, x[j : j + 1])), 20)  # :80-93
        gram_matrix[j] += jitter
        ei = (gram_matrix - np.dot(cj, ci[:i])) / dj  # :95
        ci[i, :] = ei
        di -= ei**2
        di = np.clip(di, 0, None)  # :101
        for next_idx in reversed(np.argsort(di)):  # :103-106
            if int(next_idx) not in indices[: i + 1]:
                indices[i + 1] = int(next_idx)
                count = i + 2
                break
        if np.sum(np.clip(di, 0, None)) < threshold:  # :108-113
            break
    sel = indices[:count]
    return x[sel], perm[sel], di, ci


def residual_variances(x, picks, kernel, jitter=1e-12):
    """d after conditioning on x[picks] in that order: the recurrence of conditional_variance.py:74-101 along a GIVEN
    pivot sequence (used to show that two selections part ways only at a numerical tie)."""
    n = x.shape[0]
    di = np.array([kernel(x[i : i + 1], x[i : i + 1])[0, 0] for i in range(n)]) + jitter
    ci = np.zeros((len(picks), n))
    for i, j in enumerate(picks):
        dj = np.sqrt(di[j])
        g = np.round(np.squeeze(kernel(x, x[j : j + 1])), 20)
        g[j] += jitter
        ei = (g - np.dot(ci[:i, j], ci[:i])) / dj
        ci[i, :] = ei
        di = np.clip(di - ei**2, 0, None)
    return di
