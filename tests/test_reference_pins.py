"""The oracle against the reference's OWN output: the exactly repeated colours of media/direct-lighting.png and
media/indirect-lighting.png (tests/golden/reference_media_colors.json, extracted in the build container by
tests/golden/extract_reference_media_colors.py — colour counts only, never the images).

These are the only values of the reference binary that exist anywhere besides the Halton known answers; they pin, against
the original's pixels: XMCOLOR quantisation + XMLoadColor (texture.cpp:5), Emissive::Emit (material.cpp:172-175,
spheres-app.cpp:120-121,255), the exposure 2^-15 (spheres-app.cpp:174), ACES + gamma + XMStoreColor
(spheres-app.cpp:196-214), the CheckerTexture colours and DirectionalLight::Shade's diffuse term with its occlusion test
(light.cpp:11-42, spheres-app.cpp:60,129), and Metal::Scatter's attenuation (material.cpp:87).  No GPU here; the GPU
side of the same pins is tests/test_gpu_parity.py::test_reference_media_colors_on_the_device."""
import ctypes as C
import json
import os

import numpy as np

from conftest import GOLDEN

MEDIA = json.load(open(os.path.join(GOLDEN, "reference_media_colors.json")))
SKY = MEDIA["meaning"]["sky"]


def test_fixture_is_what_the_extraction_script_reads():
    d = MEDIA["direct-lighting.png"]
    i = MEDIA["indirect-lighting.png"]
    assert d["corner_pixel_2_2"] == SKY and i["corner_pixel_2_2"] == SKY
    assert d["top_colors"][0]["rgb"] == SKY and d["top_colors"][0]["pixels"] > 350_000
    assert i["top_colors"][0]["rgb"] == SKY and i["top_colors"][0]["pixels"] > 350_000
    if os.path.isdir("/root/reference/media"):  # build container only: the fixture is current
        import subprocess
        import sys
        here = json.dumps(MEDIA, indent=1)
        subprocess.check_call([sys.executable, os.path.join(GOLDEN, "extract_reference_media_colors.py")], stdout=subprocess.DEVNULL)
        assert json.dumps(json.load(open(os.path.join(GOLDEN, "reference_media_colors.json"))), indent=1) == here


def _sky_emit(oracle, sc):
    """Emissive::Emit of the sky material (the Shade half of the unit entry is switched off with a dark sun)."""
    dark = oracle.RtLight.from_buffer_copy(bytes(sc.sun))
    dark.luminance = 0.0
    z = (C.c_float * 3)(0, 0, 0)
    emit = (C.c_float * 3)()
    oracle.lib().orc_unit_emit_shade(C.byref(sc.sky), C.byref(dark), (C.c_float * 3)(0, 0, 5), z, (C.c_float * 3)(0, 1, 0),
                                     (C.c_float * 2)(0, 0), emit)
    return np.array(list(emit), dtype=np.float32)


def _tonemap(oracle, rgb_sum, n):
    out = (C.c_uint8 * 3)()
    oracle.lib().orc_tonemap((C.c_float * 3)(*[float(np.float32(v)) for v in rgb_sum]), n, out)
    return list(out)


def test_sky_pixels_equal_the_reference_images(oracle):
    """A path that misses everything returns sky.Emit (spheres-app.cpp:255); n such samples x 2^-15, summed in order,
    resolve to the sky colour of both reference captures for any n."""
    sc = oracle.build_scene("cover", 1, 1.5)
    sample = _sky_emit(oracle, sc) * np.float32(sc.exposure_scale)
    assert sc.exposure_scale == 2.0 ** -15
    for n in (1, 2, 128, 1024):
        acc = np.zeros(3, dtype=np.float32)
        for _ in range(n):
            acc = acc + sample
        assert _tonemap(oracle, acc, n) == SKY, n
    # and through the whole render loop: the top-left pixel of the headline image never hits anything
    orc = oracle.Oracle()
    orc.upload(sc)
    orc.render(1200, 800, 1, 9, 50, 1, rowset=oracle.RtRowset(0, 2, 2, 0, 1))
    orc.resolve()
    _, ldr = orc.download()
    assert list(ldr[0, 0]) == SKY and list(ldr[1, 1199]) == SKY


def test_direct_lighting_frame_has_the_reference_images_colours(oracle):
    """depth 0 = Emit + Shade of the first hit only: the frame media/direct-lighting.png shows.  Its six most frequent
    colours — sky, occluded, sun-lit light and dark floor squares (two roundings each) — are the six most frequent colours
    of the reference's capture, and nothing else comes close."""
    sc = oracle.build_scene("cover", 1, 1.5)
    orc = oracle.Oracle()
    orc.upload(sc)
    orc.render(1200, 800, 1, 5, 0, 1, threads=8)
    orc.resolve()
    _, ldr = orc.download()
    _assert_direct_lighting_colours(ldr)


def _assert_direct_lighting_colours(ldr):
    u, c = np.unique(ldr.reshape(-1, 3), axis=0, return_counts=True)
    o = np.argsort(-c, kind="stable")
    top6 = {tuple(int(v) for v in u[i]) for i in o[:6]}
    want = {tuple(t["rgb"]) for t in MEDIA["direct-lighting.png"]["top_colors"]}
    assert top6 == want, (top6, want)
    assert tuple(int(v) for v in u[o[0]]) == tuple(SKY) and tuple(int(v) for v in u[o[1]]) == (0, 0, 0)
    assert c[o[5]] > 1.2 * c[o[6]]  # the six stand clear of the anti-aliased rest


def test_big_metal_sphere_mirroring_the_sky_equals_the_reference_image(oracle):
    """indirect-lighting.png's second colour (59,827 px) is the top of the big Metal sphere (spheres-app.cpp:108-109)
    mirroring the sky: attenuation = reflectance texture XMCOLOR(0.7,0.6,0.5) (material.cpp:87) times sky.Emit."""
    sc = oracle.build_scene("cover", 1, 1.5)
    big = int(np.flatnonzero((sc.spheres["cx"] == 4) & (sc.spheres["r"] == 1))[0])
    m = oracle.RtMaterial.from_buffer_copy(sc.materials[big].tobytes())
    assert m.type == 1
    att = (C.c_float * 3)()
    d = (C.c_float * 3)()
    nd = C.c_uint32(0)
    rc = oracle.lib().orc_unit_scatter(C.byref(m), (C.c_float * 3)(0.6, -0.8, 0.0), (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0, 1, 0),
                                       (C.c_float * 2)(0.5, 0.5), (C.c_float * 3)(0.5, 0, 0), att, d, C.byref(nd))
    assert rc == 1
    sample = (np.array(list(att), dtype=np.float32) * _sky_emit(oracle, sc)) * np.float32(sc.exposure_scale)
    assert _tonemap(oracle, sample, 1) == MEDIA["meaning"]["big_metal_sphere_mirroring_sky"]
    assert MEDIA["indirect-lighting.png"]["top_colors"][1]["rgb"] == MEDIA["meaning"]["big_metal_sphere_mirroring_sky"]
