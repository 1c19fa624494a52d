"""Write the four benchmark workloads as pbrt-v3 scene files that rustracer itself can render:

    python scripts/export_scenes.py out_dir [--small]

cornell.pbrt, blob.pbrt (+ binary PLY), mis.pbrt, room.pbrt (+ PFM textures and environment map). They are the scenes
bench.py times; a maintainer with a Rust toolchain can run `rustracer-cli out_dir/cornell.pbrt` for the reference's own
numbers on the same inputs. --small: reduced resolutions / samples for a quick look."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rustracer_amd.pbrt_export import write_pbrt
    from rustracer_amd.scenes import blob_scene, cornell_box, mis_plates, room_env
    out = sys.argv[1] if len(sys.argv) > 1 else "scenes_out"
    small = "--small" in sys.argv
    os.makedirs(out, exist_ok=True)
    scenes = [cornell_box(256, 256, 64) if small else cornell_box(1024, 1024, 1024),
              blob_scene(nu=128, nv=64, xres=320, yres=180, spp=16) if small else blob_scene(),
              mis_plates(320, 180, 16, sphere_level=1) if small else mis_plates(1280, 720, 512),
              room_env(480, 270, 16, detail=2, tex_size=128, env_size=256) if small else room_env()]
    for d, name in zip(scenes, ("cornell", "blob", "mis", "room")):
        d.name = name
        write_pbrt(d, os.path.join(out, name + ".pbrt"), ply_over=20000)
        print(f"{name}.pbrt: {d.n_tris} triangles, {len(d.lights)} lights, {d.film.xres}x{d.film.yres}, {d.sampler.spp} spp")


if __name__ == "__main__":
    main()
