"""ORACLE — TEST INFRASTRUCTURE ONLY.

Seeded inputs and the digest format of the learner-math fixtures (tests/golden/*_math.*).  Used by
oracle/gen_golden_math.py (build container: feeds them to the reference's own learner classes
running on oracle/tf_shim.py) and by tests/ (feeds the same inputs to the oracles and to the HIP
path), so the fixtures hold OUTPUTS only.  Every value is float32-representable, so the float64
fixture run, the oracles and the float32 kernels start from bit-identical numbers.
"""
import math
import zlib

import numpy as np

DIGEST_SAMPLES = 257


def _rs(tag, seed):
    return np.random.RandomState((zlib.crc32(tag.encode()) + 1000003 * seed) & 0x7FFFFFFF)


def make_params(specs, seed, scope="main"):
    """specs: [(name, shape)] in variable order.  Kernels glorot-uniform, biases uniform(-0.05, 0.05)
    (non-zero so that every bias path is exercised), one stream per (scope, name, seed)."""
    out = []
    for name, shape in specs:
        rs = _rs(scope + ":" + name, seed)
        shape = tuple(int(s) for s in shape)
        if len(shape) == 2:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            out.append(rs.uniform(-lim, lim, size=shape).astype(np.float32))
        else:
            out.append(rs.uniform(-0.05, 0.05, size=shape).astype(np.float32))
    return out


def sac_batch(obs_dim, act_dim, n, seed, act_scale=1.0):
    """SURVEY §8(d) synthetic transitions: obs ~ N(0,1), acts ~ U(-scale, scale), rews ~ N(0,1),
    done ~ Bernoulli(0.05), + the four explicit normal draws the SAC1 graph asks for (creation order:
    main pi @ x, main pi @ x2, target pi @ x2, target pi @ x2 again — the last one is never fetched)."""
    rs = _rs("sac_batch", seed)
    batch = dict(obs1=rs.randn(n, obs_dim).astype(np.float32), obs2=rs.randn(n, obs_dim).astype(np.float32),
                 acts=rs.uniform(-act_scale, act_scale, (n, act_dim)).astype(np.float32),
                 rews=rs.randn(n).astype(np.float32), done=(rs.rand(n) < 0.05).astype(np.float32))
    noise = [rs.randn(n, act_dim).astype(np.float32) for _ in range(4)]
    return batch, noise


def dqn_batch(obs_dim, n_actions, n, seed, pixels=False):
    """Discrete-action transitions; pixels=True: observations U{0..255} as float32 (config 5)."""
    rs = _rs("dqn_batch", seed)
    if pixels:
        o1 = rs.randint(0, 256, (n, obs_dim)).astype(np.float32)
        o2 = rs.randint(0, 256, (n, obs_dim)).astype(np.float32)
    else:
        o1, o2 = rs.randn(n, obs_dim).astype(np.float32), rs.randn(n, obs_dim).astype(np.float32)
    return dict(obs1=o1, obs2=o2, acts=rs.randint(0, n_actions, n).astype(np.float32),
                rews=rs.randn(n).astype(np.float32), done=(rs.rand(n) < 0.05).astype(np.float32))


def sample_idx(n):
    return np.unique(np.linspace(0, n - 1, min(n, DIGEST_SAMPLES)).astype(np.int64))


def digest(arr):
    """What a fixture keeps of a (possibly 11 M-element) tensor: evenly spaced samples, sum, L2 norm."""
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    return np.concatenate([a[sample_idx(a.size)], [a.sum(), math.sqrt(float((a * a).sum()))]])


def digest_close(got, want_digest, rtol, atol_of_norm):
    """got: full tensor; want_digest: digest() of the expected one.  |Δ| <= rtol*|want| + atol_of_norm*norm/sqrt(n)
    element-wise on the samples; sum and norm to the same relative bar (sum: against norm)."""
    g = digest(got)
    w = np.asarray(want_digest, dtype=np.float64)
    n = np.asarray(got).size
    norm = w[-1]
    atol = atol_of_norm * norm / math.sqrt(max(n, 1))
    err = np.abs(g[:-2] - w[:-2]) - (rtol * np.abs(w[:-2]) + atol)
    ok = bool((err <= 0).all()) and abs(g[-1] - w[-1]) <= rtol * norm + atol \
        and abs(g[-2] - w[-2]) <= (rtol + atol_of_norm) * norm * math.sqrt(max(n, 1))
    return ok, float(np.abs(g[:-2] - w[:-2]).max()), float(norm)
