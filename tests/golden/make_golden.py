"""Regenerates the oracle-made fixtures in tests/golden/ (run from the repo root).

The reference has no tests, fixtures or golden images of its own; its only pinned
outputs are the Halton known answers in halton_known_answers.json.  The files written here are
outputs of the repo's CPU oracle (oracle/), committed so that (a) a regression in the oracle itself
is caught on CPU and (b) the GPU parity tests have data that does not depend on rebuilding the
oracle identically on the GPU box.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    orc = O.Oracle()
    # C1: three spheres 200x100 spp 1 depth 8 (BASELINE.json configs[0])
    sc = O.build_scene("three", 1, 2.0)
    orc.upload(sc)
    st = orc.render(200, 100, 1, 2, 8, 1)
    orc.resolve()
    hdr, ldr = orc.download()
    np.savez_compressed(os.path.join(HERE, "c1_three_200x100_spp1_d8.npz"), hdr=hdr, ldr=ldr,
                        traversals=np.uint64(st.traversals), segments=np.uint64(st.segments))
    # cover scene dump, seed 1 (pins InitScene)
    sc = O.build_scene("cover", 1, 1.5)
    np.savez_compressed(os.path.join(HERE, "cover_seed1_scene.npz"), spheres=sc.spheres, materials=sc.materials,
                        camera=np.frombuffer(bytes(sc.camera), dtype=np.float32), sun=np.frombuffer(bytes(sc.sun), dtype=np.float32),
                        sky=np.frombuffer(bytes(sc.sky), dtype=np.uint8), exposure=np.float32(sc.exposure_scale))
    # C2 crop: rows 400..463 of the 1200x800 cover image, all columns would be big; keep a 96x64 full-image render instead
    orc.upload(sc)
    st = orc.render(96, 64, 1, 5, 50, 1, threads=8)
    orc.resolve()
    hdr, ldr = orc.download()
    np.savez_compressed(os.path.join(HERE, "cover_96x64_spp4_d50.npz"), hdr=hdr, ldr=ldr, traversals=np.uint64(st.traversals),
                        segments=np.uint64(st.segments))
    # per-sample vectors at the headline config's geometry (1200x800, s up to 128, depth 50)
    rng = np.random.default_rng(2024)
    n = 1500
    ijs = np.stack([rng.integers(0, 1200, n), rng.integers(0, 800, n), rng.integers(1, 129, n)], 1).astype(np.uint32)
    rgb, trav = orc.trace(1200, 800, ijs, 50, 1)
    rays = orc.primary_rays(1200, 800, ijs)
    hits = orc.closest_hit(rays)
    np.savez_compressed(os.path.join(HERE, "c2_cover_1200x800_samples.npz"), ijs=ijs, rgb=rgb, traversals=trav, rays=rays, hits=hits)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
