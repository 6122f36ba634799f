"""ORACLE — TEST INFRASTRUCTURE ONLY.

NumPy restatement of the build's counter-based noise generator (csrc/ddrl_common.h: mix32 / hash3 /
u01; csrc/common.hip: k_normal_fill / k_uniform_fill).  It stands where the reference calls
tf.random_normal (algos/sac1/core.py:77) and env.action_space.sample() (example/dsac.py:99); those
third-party generators are not reproducible outside TF/gym, so the noise is an explicit, seeded,
counter-indexed input on both sides.  Integer parts are bit-exact; Box-Muller goes through
log/cos/sqrt (float32) and is compared with a small tolerance.
"""
import numpy as np


def mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def hash3(seed, a, b):
    seed = np.uint32(int(seed) & 0xFFFFFFFF)
    a = np.asarray(a, dtype=np.uint32)
    b = np.asarray(b, dtype=np.uint32)
    h = mix32(np.uint32(seed ^ np.uint32(0x9E3779B9)))
    h = mix32(h + a * np.uint32(0x85EBCA6B) + np.uint32(0x27D4EB2F))
    h = mix32(h ^ (b * np.uint32(0xC2B2AE35) + np.uint32(0x165667B1)))
    return h


def u01(h):
    return (np.asarray(h, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def normal_fill(n, seed, counter=0):
    c = np.uint64(counter) + np.arange(n, dtype=np.uint64)
    lo = (c & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (c >> np.uint64(32)).astype(np.uint32)
    h1 = hash3(seed, lo, np.uint32(2) * hi)
    h2 = hash3(seed, lo, np.uint32(2) * hi + np.uint32(1))
    u1 = ((h1 >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    u2 = u01(h2)
    r = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
    return (r * np.cos(np.float32(6.28318530717958647692) * u2, dtype=np.float32)).astype(np.float32)


def uniform_fill(n, lo_v, hi_v, seed, counter=0):
    c = np.uint64(counter) + np.arange(n, dtype=np.uint64)
    lo = (c & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (c >> np.uint64(32)).astype(np.uint32)
    h = hash3(seed, lo, np.uint32(2) * hi)
    return (np.float32(lo_v) + (np.float32(hi_v) - np.float32(lo_v)) * u01(h)).astype(np.float32)
