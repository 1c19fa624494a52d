"""Render a pbrt-v3 scene file on the GPU and write the linear-RGB film as a PFM image.

    python scripts/render_pbrt.py scene.pbrt [out.pfm] [--spp N]

What `rustracer scene.pbrt` does, with the C++ host's parser (rtxh_pbrt_load) in front of the HIP path. The reference
writes PNG / EXR from the same film values (rc/film.rs:196-247); image encoding is outside this backend."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rustracer_amd import host
    from rustracer_amd.ingest import write_pfm
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
    out = a.out or os.path.splitext(s.film_filename)[0] + ".pfm"
    write_pfm(out, host.film_to_rgb(film, s.params.film_scale))
    print(f"{out}: {film.shape[1]}x{film.shape[0]}, {s.params.spp} spp, {stats['ms_total']:.1f} ms, {s.n_warnings} parser warnings")


if __name__ == "__main__":
    main()
