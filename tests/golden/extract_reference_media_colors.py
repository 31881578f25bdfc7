"""Extracts the colour statistics the reference's own output images hold (run in the BUILD container only:
/root/reference does not exist on the GPU box) and writes them to reference_media_colors.json.

media/direct-lighting.png and media/indirect-lighting.png are lossless captures of the reference's window
(spheres-app.cpp:163-214 output, D2D blit of the BGRA8 back buffer).  They are the only outputs of the reference
binary that exist anywhere, so their exactly repeated colours are the only golden VALUES the reference holds for
the path beyond the Halton known answers:

* the sky, ~356,000 px in each image: Emissive::Emit (material.cpp:172-175) of XMCOLOR(0.85,0.91,0.98) x 8000
  (spheres-app.cpp:120-121,255), exposure 2^-15 (:174), ACES + gamma + XMStoreColor (:196-214);
* direct-lighting.png (a direct-light-only frame): sun-lit floor squares of the CheckerTexture (texture.cpp:13-33,
  spheres-app.cpp:60) through DirectionalLight::Shade (light.cpp:11-42), and black for occluded points (:15-18);
* indirect-lighting.png: the big Metal sphere (spheres-app.cpp:108-109) mirroring the sky: Metal::Scatter's
  attenuation (material.cpp:87) x the sky.

Only counts of exact colours are stored (data), never the images.  media/Screenshot.PNG comes from a build with another
exposure (sky 221,226,231) and is not used.
"""
import json
import os

import numpy as np
from PIL import Image

MEDIA = "/root/reference/media"
HERE = os.path.dirname(os.path.abspath(__file__))


def top_colors(name, k):
    im = np.array(Image.open(os.path.join(MEDIA, name)).convert("RGB"))
    u, c = np.unique(im.reshape(-1, 3), axis=0, return_counts=True)
    o = np.argsort(-c, kind="stable")[:k]
    return {"size": [int(im.shape[1]), int(im.shape[0])], "corner_pixel_2_2": [int(v) for v in im[2, 2]],
            "top_colors": [{"rgb": [int(v) for v in u[i]], "pixels": int(c[i])} for i in o]}


def main():
    out = {
        "source": "SakibSaikia/CPURayTracer media/*.png (window captures of the reference's own renders)",
        "direct-lighting.png": top_colors("direct-lighting.png", 6),
        "indirect-lighting.png": top_colors("indirect-lighting.png", 2),
        "meaning": {
            "sky": [150, 155, 160],
            "occluded": [0, 0, 0],
            "lit_floor_light_square": [[216, 214, 210], [215, 214, 210]],
            "lit_floor_dark_square": [[123, 149, 73], [122, 149, 73]],
            "big_metal_sphere_mirroring_sky": [124, 119, 111],
        },
    }
    with open(os.path.join(HERE, "reference_media_colors.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
