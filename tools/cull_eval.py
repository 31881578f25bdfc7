"""Offline estimate (CPU, float64): candidate groups per ray of the filter, with and without a 'beyond the nearest
big-sphere hit' cull.  Rays: primary rays, then bounce rays generated from their hit points (diffuse / mirror)."""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle_py as O
from cpuraytracer_amd import _capi
L = _capi.load()
sc = O.build_scene("cover", 1, 1.5)
sph = sc.spheres
n = C.c_uint32(0)
L.rt_unit_layout(sph.ctypes.data, sph.shape[0], 0, C.byref(n), None, None)
orig = np.zeros(n.value * 4, dtype=np.uint32); bounds = np.zeros((n.value, 4), dtype=np.float32)
L.rt_unit_layout(sph.ctypes.data, sph.shape[0], n.value, C.byref(n), orig.ctypes.data, bounds.ctypes.data)
orig = orig.reshape(-1, 4); G = orig.shape[0]
c = np.stack([sph["cx"], sph["cy"], sph["cz"]], 1).astype(np.float64); r = sph["r"].astype(np.float64)
Cg = bounds[:, :3].astype(np.float64); W = bounds[:, 3].astype(np.float64)
valid = W < 1e29
Rf = np.sqrt(np.maximum((Cg ** 2).sum(1) - W, 0)); Rf[~valid] = 0
nmem = (orig != 0xFFFFFFFF).sum(1)
single = valid & (nmem == 1)
print("groups", G, "valid", valid.sum(), "singletons", single.sum(), "Rf ordinary median %.2f max %.2f" % (np.median(Rf[valid & ~single]), Rf[valid & ~single].max()))
orc = O.Oracle(); orc.upload(sc)
rng = np.random.default_rng(0); m = 3000
ijs = np.stack([rng.integers(0, 1200, m), rng.integers(0, 800, m), rng.integers(1, 129, m)], 1).astype(np.uint32)
rays = orc.primary_rays(1200, 800, ijs).astype(np.float64)

def closest(rs):
    h = orc.closest_hit(rs.astype(np.float32)); idx = h[:, 1].view(np.int32); return idx, h[:, 0].astype(np.float64), h[:, 2:5].astype(np.float64), h[:, 5:8].astype(np.float64)

def analyse(name, rs):
    o = rs[:, :3]; d = rs[:, 3:]; a = (d * d).sum(1)
    oc = o[:, None, :] - Cg[None]; b = (oc * d[:, None, :]).sum(2); cc = (oc ** 2).sum(2) - Rf[None] ** 2
    F = b * b - a[:, None] * cc
    behind = (cc > 0) & (b > 0)
    cand = (F >= 0) & ~behind & valid[None]
    # exact nearest hit among the singleton (big) spheres
    tub = np.full(len(rs), np.inf)
    for g in np.where(single)[0]:
        i = orig[g][orig[g] != 0xFFFFFFFF][0]
        ocs = o - c[i]; bs = (ocs * d).sum(1); cs = (ocs ** 2).sum(1) - r[i] ** 2; ds = bs * bs - a * cs
        with np.errstate(invalid="ignore"):
            sq = np.sqrt(np.maximum(ds, 0)); t1 = (-bs - sq) / a; t2 = (-bs + sq) / a
        t = np.where(t1 > 1e-3, t1, np.where(t2 > 1e-3, t2, np.inf)); t = np.where(ds > 0, t, np.inf); tub = np.minimum(tub, t)
    # a group is beyond if even its nearest point is farther than tub: t_centre - R/sqrt(a) > tub
    tcen = -b / a[:, None]
    beyond = (tcen - Rf[None] / np.sqrt(a)[:, None]) > tub[:, None]
    # sharper: the entry point of the bound, (-b - sqrt(F))/a > tub
    with np.errstate(invalid="ignore"):
        tin = (-b - np.sqrt(np.maximum(F, 0))) / a[:, None]
    beyond2 = tin > tub[:, None]
    ord_ = cand & ~single[None]
    c0 = cand.sum(1); c1 = (ord_ & ~beyond).sum(1); c2 = (ord_ & ~beyond2).sum(1)
    def wave_max(x):  # mean over random 64-ray waves of the max
        k = len(x) // 64 * 64; p = rng.permutation(len(x))[:k]; return x[p].reshape(-1, 64).max(1).mean()
    print("%-8s rays %5d  cand now: mean %.2f wavemax %.1f | ordinary only, centre cull: mean %.2f wavemax %.1f | entry cull: mean %.2f wavemax %.1f | tub finite %.2f"
          % (name, len(rs), c0.mean(), wave_max(c0), c1.mean(), wave_max(c1), c2.mean(), wave_max(c2), np.isfinite(tub).mean()))

analyse("primary", rays)
idx, t, pos, nrm = closest(rays)
hit = idx >= 0
pos = pos[hit]; nrm = nrm[hit]; din = rays[hit, 3:]
# diffuse bounce: uniform hemisphere about the normal; mirror bounce
v = rng.normal(size=pos.shape); v /= np.linalg.norm(v, axis=1, keepdims=True); v *= np.sign((v * nrm).sum(1))[:, None]
analyse("diffuse", np.concatenate([pos, v], 1))
dn = (din * nrm).sum(1)[:, None]; refl = din - 2 * dn * nrm; refl /= np.linalg.norm(refl, axis=1, keepdims=True)
analyse("mirror", np.concatenate([pos, refl], 1))
# second bounce from diffuse
idx2, t2, pos2, nrm2 = closest(np.concatenate([pos, v], 1)); h2 = idx2 >= 0
v2 = rng.normal(size=pos2[h2].shape); v2 /= np.linalg.norm(v2, axis=1, keepdims=True); v2 *= np.sign((v2 * nrm2[h2]).sum(1))[:, None]
analyse("diffuse2", np.concatenate([pos2[h2], v2], 1))
# distribution of candidates per ray (all ray kinds pooled): what a per-iteration step cap would defer
allr = np.concatenate([rays, np.concatenate([pos, v], 1), np.concatenate([pos, refl], 1)], 0)
o = allr[:, :3]; d = allr[:, 3:]; a = (d * d).sum(1)
oc = o[:, None, :] - Cg[None]; b = (oc * d[:, None, :]).sum(2); cc = (oc ** 2).sum(2) - Rf[None] ** 2
cand = ((b * b - a[:, None] * cc) >= 0) & ~((cc > 0) & (b > 0)) & valid[None]
k = cand.sum(1)
print("cand/ray percentiles:", {q: int(np.percentile(k, q)) for q in (50, 75, 90, 95, 98, 99, 100)}, "mean %.2f" % k.mean())
for cap in (4, 5, 6, 8):
    print("cap %d: rays over %.3f, mean extra iterations %.3f" % (cap, (k > cap).mean(), (np.ceil(np.maximum(k, 1) / cap) - 1).mean()))
