"""Render a pbrt-v3 scene file on the GPU.

    python scripts/render_pbrt.py scene.pbrt [out.png|out.pfm] [--spp N]

What `rustracer scene.pbrt` does, with the C++ host's parser (rtxh_pbrt_load) in front of the HIP path. Without an output
name the image goes where the reference writes it: "rt-" + the Film's filename, or image.png (rc/film.rs:118-123), as an
8-bit sRGB PNG with write_image_png's quantisation (rc/imageio.rs:52-74). A .pfm name gets the linear film values."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rustracer_amd import host
    from rustracer_amd.ingest import write_pfm, write_png
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("out", nargs="?")
    ap.add_argument("--spp", type=int, default=0, help="override Sampler pixelsamples")
    a = ap.parse_args()
    host.build()
    s = host.PbrtScene(a.scene)
    if a.spp:
        s.params.spp = a.spp
    film, stats = s.render()
    out = a.out or s.film_filename
    rgb = host.film_to_rgb(film, s.params.film_scale)
    if out.endswith(".pfm"):
        write_pfm(out, rgb)
    elif out.endswith(".png"):
        write_png(out, host.rgb_to_png8(rgb), 2, 8, filters=(1,))
    else:
        raise SystemExit("Unsupported file format")   # rc/imageio.rs:47-49 (EXR output is not written here)
    print(f"{out}: {film.shape[1]}x{film.shape[0]}, {s.params.spp} spp, {stats['ms_total']:.1f} ms, {s.n_warnings} parser warnings")


if __name__ == "__main__":
    main()
