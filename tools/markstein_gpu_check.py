import sys, numpy as np
sys.path.insert(0, '/root/repo')
from cpuraytracer_amd import HipRenderer
r = HipRenderer(0)
# (a) refined reciprocal for ALL 2^23 significands at several exponents
sig = np.arange(1 << 23, dtype=np.uint32)
bad = 0
for e in (27, 60, 100, 126, 127, 128, 150, 200, 227):
    x = (sig | np.uint32(e << 23)).view(np.float32)
    got = r.unit_math(4, x)
    want = (np.float32(1.0) / x).astype(np.float32)
    n = int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
    bad += n
    print("recip exponent", e - 127, "mismatches", n, flush=True)
# (b) guarded quotient vs the device's own IEEE division and numpy's
rng = np.random.default_rng(3)
tot = 0
for rep in range(8):
    n = 4_000_000
    eb = rng.integers(-30, 110, n); ex = eb + rng.integers(-100, 30, n)
    b = (rng.uniform(1, 2, n) * np.exp2(eb.astype(np.float64))).astype(np.float32)
    x = (rng.uniform(1, 2, n) * np.exp2(np.clip(ex, -148, 126).astype(np.float64)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    x[:1000] = 0.0; x[1000:2000] = -0.0
    got = r.unit_math(5, x, b); dev = r.unit_math(6, x, b)
    want = (x / b).astype(np.float32)
    m1 = int(np.count_nonzero(got.view(np.uint32) != dev.view(np.uint32)))
    m2 = int(np.count_nonzero(dev.view(np.uint32) != want.view(np.uint32)))
    tot += m1
    print("quotients rep", rep, "markstein vs device ieee:", m1, " device ieee vs numpy:", m2, flush=True)
# (c) the all-ones significand and friends
sp = np.array([0x3fffffff, 0x3f800000, 0x3f800001, 0x3ffffffe, 0x3fc00000], dtype=np.uint32).view(np.float32)
xs = rng.uniform(-4, 4, 1_000_000).astype(np.float32)
for b0 in sp:
    bb = np.full_like(xs, b0)
    tot += int(np.count_nonzero(r.unit_math(5, xs, bb).view(np.uint32) != (xs / bb).astype(np.float32).view(np.uint32)))
print("TOTAL reciprocal mismatches", bad, "quotient mismatches", tot)
