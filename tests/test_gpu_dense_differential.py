"""Dense GPU-vs-oracle differentials (-m gpu), through the C ABI.

Every speed-up of the scan is a conservative cull with a hand-derived margin, and one early version of
one of them was wrong on 1 path in 1e5.  Spot checks of ~1e4 paths per config cannot see that, so this file compares

  * EVERY path of BASELINE.json's C2, C3, C4 and C5 at full size with the CPU oracle's render of the same job, through
    digests the oracle produced in the build container (tests/golden/full_size_oracle_digests.json, made by
    tests/golden/make_full_size_golden.py: traversal and segment totals, sha256 of the HDR strip and the LDR bytes, one
    CRC32 per HDR row so that a failure names its rows): 1.2e8 + 9.8e8 + 1.06e9 + 1.07e9 paths;
  * whole frames with the LIVE oracle on the GPU box's cores (C2 at full spp; C4 and C5 at reduced spp), so that the
    comparison does not rest on the committed digests alone;
  * 2 M random (i, j, s) per-sample radiances + traversal counts on C4 and C5 geometry against the live oracle.

Reference semantics being protected: ray-tracing.cpp:42-84,174-214 (closest hit), light.cpp:13-18 (occlusion)."""
import hashlib
import json
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

DIGESTS = json.load(open(os.path.join(GOLDEN, "full_size_oracle_digests.json")))
THREADS = max(1, min(16, os.cpu_count() or 1))


def _scene_for(scenes_mod, rec):
    sc = scenes_mod.build_scene(rec["scene"], rec["scene_seed"], rec["W"], rec["H"], aperture=rec["aperture"])
    assert sc.n == rec["n_spheres"]
    return sc


@pytest.fixture(scope="module")
def scenes_mod(built):
    from cpuraytracer_amd import scenes
    return scenes


def _assert_matches_digest(name, st, hdr, ldr):
    rec = DIGESTS[name]
    assert st.samples == rec["samples"] == rec["W"] * rec["H"] * rec["spp"]
    assert (st.traversals, st.segments) == (rec["traversals"], rec["segments"]), "%s: traversal/segment totals differ from the oracle's" % name
    hdr = np.ascontiguousarray(hdr, dtype="<f4")
    if hashlib.sha256(hdr.tobytes()).hexdigest() != rec["hdr_sha256"]:
        bad = [j for j in range(hdr.shape[0]) if zlib.crc32(hdr[j].tobytes()) != rec["hdr_row_crc32"][j]]
        raise AssertionError("%s: HDR strip differs from the oracle's render in %d of %d rows, first rows %s" % (
            name, len(bad), hdr.shape[0], bad[:12]))
    assert hashlib.sha256(np.ascontiguousarray(ldr).tobytes()).hexdigest() == rec["ldr_sha256"], "%s: LDR bytes differ" % name


@pytest.mark.parametrize("name", ["c2", "c4", "c5", "c3", "c2_scene2", "c2_scene3_seed7", "c5_small_scene2"])
def test_full_size_config_equals_the_oracles_render_of_every_path(hip, scenes_mod, name):
    """BASELINE.json configs[1..4] at full size on one device (and the headline job on two other random scenes / streams, and a
    smaller job on another 10,004-sphere scene): totals, HDR bits and LDR bytes of the whole job equal the oracle's (list-scan
    semantics, PaddedListTree) — every one of the job's paths is compared, through the committed digests."""
    rec = DIGESTS[name]
    hip.upload(_scene_for(scenes_mod, rec))
    st = hip.render(rec["W"], rec["H"], 1, 1 + rec["spp"], rec["depth"], rec["render_seed"])
    hip.resolve()
    hdr, ldr = hip.download()
    _assert_matches_digest(name, st, hdr, ldr)


def test_c3_job_passes_through_the_c2_image_after_128_samples(hip, scenes_mod):
    """BASELINE config 3 is config 2's accumulation carried on to spp 1024: rendered progressively, the strip after samples
    1..128 is C2's image (its digest) and after 129..1024 it is C3's."""
    rec2, rec3 = DIGESTS["c2"], DIGESTS["c3"]
    hip.upload(_scene_for(scenes_mod, rec3))
    st = hip.render(rec3["W"], rec3["H"], 1, 129, rec3["depth"], rec3["render_seed"])
    hip.resolve()
    hdr, ldr = hip.download()
    _assert_matches_digest("c2", st, hdr, ldr)
    st2 = hip.render(rec3["W"], rec3["H"], 129, 1025, rec3["depth"], rec3["render_seed"])
    hip.resolve()
    hdr, ldr = hip.download()
    assert st.traversals + st2.traversals == rec3["traversals"] and st.segments + st2.segments == rec3["segments"]
    assert hashlib.sha256(np.ascontiguousarray(hdr, dtype="<f4").tobytes()).hexdigest() == rec3["hdr_sha256"]
    assert hashlib.sha256(np.ascontiguousarray(ldr).tobytes()).hexdigest() == rec3["ldr_sha256"]
    assert rec2["samples"] * 8 == rec3["samples"]


def _live_oracle_frame(oracle, sc, W, H, spp, depth, seed):
    orc = oracle.Oracle()
    orc.upload(sc)
    st = orc.render(W, H, 1, 1 + spp, depth, seed, accel=oracle.ACCEL_PADDED_LIST, threads=THREADS)
    orc.resolve()
    hdr, ldr = orc.download()
    orc.close()
    return st, hdr, ldr


@pytest.mark.parametrize("name,spp", [("c2", 128), ("c4", 16), ("c5", 2)])
def test_whole_frames_equal_the_live_oracle(hip, oracle, scenes_mod, name, spp):
    """The same comparison against the oracle RUN HERE on the box's host cores: C2's whole job (about 10 s of CPU), C4 and C5
    whole frames at spp 16 and 2 (3 s and 6 s): HDR bits, LDR bytes, traversal and segment totals."""
    rec = DIGESTS[name]
    W, H = rec["W"], rec["H"]
    sc = _scene_for(scenes_mod, rec)
    hip.upload(sc)
    sg = hip.render(W, H, 1, 1 + spp, rec["depth"], rec["render_seed"])
    hip.resolve()
    hg, lg = hip.download()
    so, ho, lo = _live_oracle_frame(oracle, sc, W, H, spp, rec["depth"], rec["render_seed"])
    assert (sg.samples, sg.traversals, sg.segments) == (so.samples, so.traversals, so.segments)
    diff = np.flatnonzero((hg.view(np.uint32) != ho.view(np.uint32)).any(axis=2).ravel())
    assert diff.size == 0, "%s spp %d: %d pixels differ from the live oracle, first (i, j): %s" % (
        name, spp, diff.size, [(int(p % W), int(p // W)) for p in diff[:8]])
    assert np.array_equal(lg, lo)
    if spp == rec["spp"]:  # the live oracle on this box also reproduces the digest made in the build container
        _assert_matches_digest(name, so, ho, lo)


@pytest.mark.parametrize("name,row0,nrows", [("c4", 620, 48), ("c5", 2016, 64), ("c3", 520, 16)])
def test_digest_rows_are_reproduced_by_the_live_oracle_at_full_spp(hip, oracle, scenes_mod, name, row0, nrows):
    """the C3 / C4 / C5 digests were made in the build container; here the oracle renders a BAND of rows of the same
    job at FULL spp on this box (rows through the middle of the sphere layer: the busiest part of the picture) and must reproduce
    the committed per-row CRC32s of exactly those rows -- and so must the GPU's render of the band alone (a row shard of the job)."""
    import cpuraytracer_amd as pkg
    rec = DIGESTS[name]
    W, H, spp = rec["W"], rec["H"], rec["spp"]
    sc = _scene_for(scenes_mod, rec)
    rs_o = oracle.RtRowset(row0, nrows, nrows, 0, 1)
    orc = oracle.Oracle()
    orc.upload(sc)
    so = orc.render(W, H, 1, 1 + spp, rec["depth"], rec["render_seed"], rowset=rs_o, accel=oracle.ACCEL_PADDED_LIST, threads=THREADS)
    ho, _ = orc.download()
    orc.close()
    ho = np.ascontiguousarray(ho, dtype="<f4")
    want = rec["hdr_row_crc32"][row0:row0 + nrows]
    assert [zlib.crc32(ho[k].tobytes()) for k in range(nrows)] == want, "%s: the live oracle's rows %d..%d differ from the committed digest" % (name, row0, row0 + nrows - 1)
    hip.upload(sc)
    sg = hip.render(W, H, 1, 1 + spp, rec["depth"], rec["render_seed"], rowset=pkg._capi.RtRowset(row0, nrows, nrows, 0, 1))
    hg = np.ascontiguousarray(hip.download(ldr=False)[0], dtype="<f4")
    assert [zlib.crc32(hg[k].tobytes()) for k in range(nrows)] == want
    assert (sg.samples, sg.traversals, sg.segments) == (so.samples, so.traversals, so.segments)


@pytest.mark.parametrize("name,n", [("c4", 2_000_000), ("c5", 2_000_000), ("c2", 1_000_000)])
def test_millions_of_random_samples_equal_the_live_oracle(hip, oracle, scenes_mod, name, n):
    """rt_unit_trace (the production launch with a caller-given list of paths) against orc_unit_trace on n random
    (i, j, s), s over the config's whole sample range: per-sample radiance bits and traversal counts."""
    rec = DIGESTS[name]
    W, H = rec["W"], rec["H"]
    sc = _scene_for(scenes_mod, rec)
    hip.upload(sc)
    rng = np.random.default_rng({"c2": 201, "c4": 204, "c5": 205}[name])
    ijs = np.stack([rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(1, rec["spp"] + 1, n)], 1).astype(np.uint32)
    rg, tg = hip.unit_trace(W, H, ijs, rec["depth"], rec["render_seed"])
    orc = oracle.Oracle()
    orc.upload(sc)
    chunks = np.array_split(np.arange(n), THREADS * 4)

    def work(idx):  # ctypes releases the GIL; TraceSample only reads the scene
        return orc.trace(W, H, ijs[idx], rec["depth"], rec["render_seed"], accel=oracle.ACCEL_PADDED_LIST)
    with ThreadPoolExecutor(THREADS) as ex:
        parts = list(ex.map(work, chunks))
    ro = np.concatenate([p[0] for p in parts])
    to = np.concatenate([p[1] for p in parts])
    bad = np.flatnonzero((rg.view(np.uint32) != ro.view(np.uint32)).any(axis=1) | (tg != to))
    assert bad.size == 0, "%s: %d of %d samples differ, first (i, j, s): %s" % (name, bad.size, n, ijs[bad[:6]].tolist())


def test_reference_media_colors_on_the_device(hip, scenes_mod):
    """The reference's own pixels (tests/golden/reference_media_colors.json, see tests/test_reference_pins.py) through the
    HIP path: a pixel whose rays miss everything resolves to the sky bytes of both reference captures, and a
    direct-light-only frame (depth 0) has the six most frequent colours of media/direct-lighting.png."""
    from test_reference_pins import MEDIA, SKY, _assert_direct_lighting_colours
    sc = scenes_mod.build_scene("cover", 1, 1200, 800)
    hip.upload(sc)
    for spp in (1, 128):
        hip.render(1200, 800, 1, 1 + spp, 50, 1)
        hip.resolve()
        _, ldr = hip.download(hdr=False)
        assert list(ldr[0, 0]) == SKY and list(ldr[2, 2]) == MEDIA["direct-lighting.png"]["corner_pixel_2_2"]
        sky = (ldr.reshape(-1, 3) == np.array(SKY, dtype=np.uint8)).all(axis=1).sum()
        assert sky > 330_000  # the whole sky region, as in the captures
    hip.render(1200, 800, 1, 5, 0, 1)
    hip.resolve()
    _, ldr = hip.download(hdr=False)
    _assert_direct_lighting_colours(ldr)
