"""The error bound behind the certified argmax of the large-batch network kernels (mlp_tile_kernel, screen pass): two
floating-point evaluations of the same three-layer network on the same features -- any summation order, fused or not --
differ by at most 2 * e3, with e3 built from the largest absolute row sum A_k and bias B_k of each layer and the largest
absolute feature X exactly as the kernel builds it.  Checked here on the CPU for the reference's left-to-right sums
(MLP.scala:15-22) against numpy's blocked / pairwise sums and against sums taken in reverse order."""
import numpy as np

U = 2.0 ** -53


def bound(W1, b1, W2, b2, W3, b3, X):
    H = W2.shape[0]
    A1, B1 = np.abs(W1).sum(axis=1).max(), np.abs(b1).max()
    A2, B2 = np.abs(W2).sum(axis=1).max(), np.abs(b2).max()
    A3, B3 = np.abs(W3).sum(axis=1).max(), np.abs(b3).max()
    g8, gh = 8.0 * 1.01 * U, (H + 1) * 1.01 * U
    H1 = A1 * X + B1; e1 = g8 * H1
    H2 = A2 * H1 + B2; e2 = A2 * e1 + gh * (A2 * (H1 + e1) + B2)
    return A3 * e2 + gh * (A3 * (H2 + e2) + B3)


def forward_sequential(W1, b1, W2, b2, W3, b3, x, reverse=False):
    def layer(W, b, v, relu):
        out = np.empty(W.shape[0])
        order = range(W.shape[1] - 1, -1, -1) if reverse else range(W.shape[1])
        for i in range(W.shape[0]):
            acc = None
            for j in order:
                p = W[i, j] * v[j]
                acc = p if acc is None else acc + p
            z = acc + b[i]
            out[i] = max(z, 0.0) if relu else z
        return out
    return layer(W3, b3, layer(W2, b2, layer(W1, b1, x, True), True), False)


def forward_numpy(W1, b1, W2, b2, W3, b3, x):
    h1 = np.maximum(W1 @ x + b1, 0.0)
    h2 = np.maximum(W2 @ h1 + b2, 0.0)
    return W3 @ h2 + b3


def test_two_evaluation_orders_stay_within_twice_the_bound():
    rng = np.random.default_rng(11)
    worst = 0.0
    for H in (64, 128):
        for trial in range(6):
            scale = 10.0 ** rng.uniform(-2, 1)
            W1 = rng.normal(size=(H, 7)) * scale; b1 = rng.normal(size=H) * 0.1
            W2 = rng.normal(size=(H, H)) * scale / np.sqrt(H); b2 = rng.normal(size=H) * 0.1
            W3 = rng.normal(size=(8, H)) * scale; b3 = rng.normal(size=8) * 0.1
            for _ in range(3):
                x = np.abs(rng.normal(size=7)) * rng.choice([0.01, 1.0, 30.0])
                e3 = bound(W1, b1, W2, b2, W3, b3, np.abs(x).max())
                q_ref = forward_sequential(W1, b1, W2, b2, W3, b3, x)
                for q in (forward_numpy(W1, b1, W2, b2, W3, b3, x), forward_sequential(W1, b1, W2, b2, W3, b3, x, reverse=True)):
                    diff = np.abs(q - q_ref).max()
                    assert diff <= 2.0 * e3, (H, trial, diff, e3)
                    worst = max(worst, diff / e3)
    assert worst < 0.5   # the bound is a worst case: observed differences stay far inside it


def test_bound_is_small_against_the_margins_of_an_untrained_network():
    """MLP.scala:5-9 initialises with Gaussians * sqrt(2 / fan_in): the bound is ~1e-12, Q-values differ by ~1e-2."""
    rng = np.random.default_rng(9)
    H = 128
    W1 = rng.normal(size=(H, 7)) * np.sqrt(2.0 / 7); W2 = rng.normal(size=(H, H)) * np.sqrt(2.0 / H)
    W3 = rng.normal(size=(8, H)) * np.sqrt(2.0 / H)
    z = np.zeros(H)
    e3 = bound(W1, z, W2, z, W3, np.zeros(8), 1.0)
    assert e3 < 1e-9
    q = forward_numpy(W1, z, W2, z, W3, np.zeros(8), np.array([0.9, 0.05, 0.03, 0.01, 0.0, 0.4, 0.5]))
    top = np.sort(q)[::-1]
    assert top[0] - top[1] > 100 * 4 * e3
