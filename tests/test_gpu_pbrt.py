"""A scene that arrives as a pbrt-v3 file renders to the same film as the same scene handed over call by call, and
to the oracle's film within the usual gate (parser -> host -> HIP path)."""
import os

import numpy as np
import pytest

from util import rel_l2

pytestmark = pytest.mark.gpu


def _films(gpu_host, desc, tmp_path):
    from rustracer_amd.pbrt_export import write_pbrt
    path = os.path.join(str(tmp_path), f"{desc.name}.pbrt")
    write_pbrt(desc, path)
    fp, sp = gpu_host.PbrtScene(path).render(count_traversal=True)
    fh, sh = gpu_host.HostScene(desc).render(count_traversal=True)
    return fp, sp, fh, sh


def test_cornell_from_file_matches_direct_and_oracle(gpu_host, orc, tmp_path):
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(96, 96, 16)
    fp, sp, fh, sh = _films(gpu_host, d, tmp_path)
    assert np.array_equal(fp, fh)
    for k in ("camera_rays", "rays_closest", "rays_shadow", "rays_mis", "nodes_closest", "tris_closest"):
        assert sp[k] == sh[k], k
    fo, _ = orc.OracleScene(d).render(mode=1)
    assert np.array_equal(fo[..., 3], fp[..., 3])
    assert rel_l2(gpu_host.film_to_rgb(fp), orc.film_to_rgb(fo)) < 1e-3


@pytest.mark.parametrize("material, light", [("matte_image_ewa", "infinite"), ("mix_nested", "point"), ("disney_sheen_textured", "distant"), ("glass_rough", "area_two_sided")])
def test_zoo_from_file_matches_direct(gpu_host, tmp_path, material, light):
    from test_gpu_materials import _zoo
    d = _zoo(material, light)
    d.name = f"{material}_{light}"
    fp, sp, fh, sh = _films(gpu_host, d, tmp_path)
    assert np.array_equal(fp[..., 3], fh[..., 3])
    assert np.isfinite(fp).all()
    # material / texture ids differ between the two builds (the parser shares equal constants), values do not
    assert rel_l2(fp[..., :3], fh[..., :3]) < 1e-6
    assert sp["rays_closest"] == sh["rays_closest"]


def test_object_instances_render_like_the_written_out_scene(gpu_host, orc):
    """ObjectBegin / ObjectInstance (api.rs:1018-1090): three placements of one box, one of them mirrored."""
    from rustracer_amd.scene_desc import SceneDesc
    from rustracer_amd.scenes.procedural import box_mesh
    from test_pbrt_cpu import _mm, _pt, _scale, _translate
    bp, bi, _ = box_mesh((-0.5, 0.0, -0.5), (0.5, 1.0, 0.5))
    bp, bi = np.asarray(bp, np.float32), np.asarray(bi, np.int32)
    box = (f'Shape "trianglemesh" "integer indices" [{" ".join(str(int(x)) for x in bi.reshape(-1))}] '
           f'"point P" [{" ".join("%.9g" % float(x) for x in bp.reshape(-1))}]\n')
    room = ['Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 -4 -4 0 4 4 0 4 4 0 -4]\n',
            'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-4 0 4 -4 5 4 4 5 4 4 0 4]\n']
    text = ('LookAt 0 3 -9  0 1 0  0 1 0\nCamera "perspective" "float fov" [45]\nSampler "02sequence" "integer pixelsamples" [16]\n'
            'Film "image" "integer xresolution" [96] "integer yresolution" [64]\nWorldBegin\n'
            'Material "matte" "rgb Kd" [0.7 0.7 0.7]\n' + room[0] + room[1] +
            'AttributeBegin\nAreaLightSource "diffuse" "rgb L" [20 18 15]\n'
            'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 4.9 -1 1 4.9 -1 1 4.9 1 -1 4.9 1]\nAttributeEnd\n'
            'ObjectBegin "box"\nMaterial "plastic" "rgb Kd" [0.2 0.4 0.7] "float roughness" [0.2]\n' + box + 'ObjectEnd\n'
            'AttributeBegin\nTranslate -2 0 0\nObjectInstance "box"\nAttributeEnd\n'
            'AttributeBegin\nTranslate 2 0 1\nScale 1 2 1\nObjectInstance "box"\nAttributeEnd\n'
            'AttributeBegin\nTranslate 0 0 -1\nScale -1.5 0.5 1.5\nObjectInstance "box"\nAttributeEnd\n'
            'WorldEnd\n')
    p = gpu_host.PbrtScene(text=text, flatten_instances=True)
    d = SceneDesc()
    grey, blue = d.matte((0.7, 0.7, 0.7)), d.plastic((0.2, 0.4, 0.7), (0.25, 0.25, 0.25), 0.2)
    d.add_quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4), grey)
    d.add_quad((-4, 0, 4), (-4, 5, 4), (4, 5, 4), (4, 0, 4), grey)
    d.add_quad((-1, 4.9, -1), (1, 4.9, -1), (1, 4.9, 1), (-1, 4.9, 1), grey, emission=(20.0, 18.0, 15.0))
    for m, mirrored in ((_translate(-2, 0, 0), False), (_mm(_translate(2, 0, 1), _scale(1, 2, 1)), False), (_mm(_translate(0, 0, -1), _scale(-1.5, 0.5, 1.5)), True)):
        d.add_mesh(np.stack([_pt(m, v) for v in bp]), bi, blue, reverse_orientation=mirrored)
    d.camera.pos, d.camera.look, d.camera.fov = (0.0, 3.0, -9.0), (0.0, 1.0, 0.0), 45.0
    d.film.xres, d.film.yres = 96, 64
    d.sampler.spp = 16
    h = gpu_host.HostScene(d)
    for name in ("P", "indices", "tri_flags", "tri_light"):
        assert np.array_equal(p.table(name), h.table(name)), name
    fp, sp = p.render(count_traversal=True)
    fh, sh = h.render(count_traversal=True)
    assert np.array_equal(fp, fh)
    fo, _ = orc.OracleScene(d).render(mode=1)
    assert np.array_equal(fo[..., 3], fp[..., 3]) and rel_l2(gpu_host.film_to_rgb(fp), orc.film_to_rgb(fo)) < 1e-3
    # the default: the reference's own form - one tree for the box, three TransformedPrimitives - against the same scene built by calls, and the oracle
    p2 = gpu_host.PbrtScene(text=text)
    d2 = SceneDesc()
    grey, blue = d2.matte((0.7, 0.7, 0.7)), d2.plastic((0.2, 0.4, 0.7), (0.25, 0.25, 0.25), 0.2)
    d2.add_quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4), grey)
    d2.add_quad((-4, 0, 4), (-4, 5, 4), (4, 5, 4), (4, 0, 4), grey)
    d2.add_quad((-1, 4.9, -1), (1, 4.9, -1), (1, 4.9, 1), (-1, 4.9, 1), grey, emission=(20.0, 18.0, 15.0))
    o = d2.add_object([dict(P=bp, idx=bi, material=blue)])
    for m in (_translate(-2, 0, 0), _mm(_translate(2, 0, 1), _scale(1, 2, 1)), _mm(_translate(0, 0, -1), _scale(-1.5, 0.5, 1.5))):
        d2.add_instance(o, m)
    d2.camera.pos, d2.camera.look, d2.camera.fov = (0.0, 3.0, -9.0), (0.0, 1.0, 0.0), 45.0
    d2.film.xres, d2.film.yres = 96, 64
    d2.sampler.spp = 16
    assert len(p2.table("instances")) == 3 and np.array_equal(p2.table((0, "P")), bp) and len(p2.table("indices")) == 6
    assert np.array_equal(p2.table("instances")["o2w"], np.stack([i.o2w for i in d2.instances]))
    f2, _ = p2.render()
    fo2, _ = orc.OracleScene(d2).render(mode=1)
    assert np.array_equal(fo2[..., 3], f2[..., 3]) and rel_l2(gpu_host.film_to_rgb(f2), orc.film_to_rgb(fo2)) < 1e-3
    assert rel_l2(gpu_host.film_to_rgb(f2), gpu_host.film_to_rgb(fp)) < 5e-3   # two roundings of one scene


def test_quadrics_inside_objects_are_placed_by_their_instances(gpu_host, orc):
    """ObjectBegin .. Shape "sphere" .. ObjectEnd + ObjectInstance: the reference's TransformedPrimitive over a sphere (rc/api.rs:1053-1090, primitive.rs:79-118).
    Round 6: the object HOLDS the spheres (object space) and is placed by its instances - against the oracle's TransformedPrimitive over the same object, built by
    calls; written out (flatten_instances) it is the sphere under instance_to_world * object_to_world - against the oracle on the scene with the spheres placed by
    hand. And an emitting sphere of an object glows at L where the camera sees it and lights nothing (api.rs:954-964)."""
    from rustracer_amd.scene_desc import SceneDesc
    from test_pbrt_cpu import _mm, _scale, _translate
    head = ('LookAt 0 3 -9  0 1 0  0 1 0\nCamera "perspective" "float fov" [45]\nSampler "02sequence" "integer pixelsamples" [16]\n'
            'Film "image" "integer xresolution" [96] "integer yresolution" [64]\nWorldBegin\nMaterial "matte" "rgb Kd" [0.7 0.7 0.7]\n'
            'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-6 0 -6 -6 0 6 6 0 6 6 0 -6]\n')
    lamp = ('AttributeBegin\nAreaLightSource "diffuse" "rgb L" [20 18 15]\n'
            'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 5.9 -1 1 5.9 -1 1 5.9 1 -1 5.9 1]\nAttributeEnd\n')
    obj = ('ObjectBegin "ball"\nMaterial "plastic" "rgb Kd" [0.2 0.4 0.7] "float roughness" [0.2]\nTranslate 0 1 0\nShape "sphere" "float radius" 0.8\n'
           'Translate 0 1.2 0\nShape "sphere" "float radius" 0.4 "float zmax" 0.2\nObjectEnd\n')
    places = ('AttributeBegin\nTranslate -2.5 0 0\nObjectInstance "ball"\nAttributeEnd\nAttributeBegin\nTranslate 2 0 1\nScale 1 1.5 1\nObjectInstance "ball"\nAttributeEnd\n')
    text = head + lamp + obj + places + 'WorldEnd\n'

    def base():
        d = SceneDesc()
        grey, blue = d.matte((0.7, 0.7, 0.7)), d.plastic((0.2, 0.4, 0.7), (0.25, 0.25, 0.25), 0.2)
        d.add_quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), grey)
        d.add_quad((-1, 5.9, -1), (1, 5.9, -1), (1, 5.9, 1), (-1, 5.9, 1), grey, emission=(20.0, 18.0, 15.0))
        d.camera.pos, d.camera.look, d.camera.fov = (0.0, 3.0, -9.0), (0.0, 1.0, 0.0), 45.0
        d.film.xres, d.film.yres = 96, 64
        d.sampler.spp = 16
        return d, blue
    # two-level (the default): the object holds its two spheres
    p = gpu_host.PbrtScene(text=text)
    assert len(p.table("quadrics")) == 0 and len(p.table("instances")) == 2 and len(p.table((0, "quadrics"))) == 2
    d, blue = base()
    o = d.add_object([], quadrics=[dict(kind=0, o2w=_translate(0, 1, 0).astype(np.float32), radius=0.8, material=blue),
                                   dict(kind=0, o2w=_translate(0, 2.2, 0).astype(np.float32), radius=0.4, z_max=0.2, material=blue)])
    for inst in (_translate(-2.5, 0, 0), _mm(_translate(2, 0, 1), _scale(1, 1.5, 1))):
        d.add_instance(o, inst.astype(np.float32))
    h = gpu_host.HostScene(d)
    for k in ("o2w", "w2o", "radius", "z_min", "z_max", "phi_max", "material", "light", "kind"):
        assert np.array_equal(p.table((0, "quadrics"))[k], h.table((0, "quadrics"))[k]), k
    fo, _ = orc.OracleScene(d).render(mode=1)
    fp, _ = p.render()
    assert np.array_equal(fo[..., 3], fp[..., 3]) and rel_l2(gpu_host.film_to_rgb(fp), orc.film_to_rgb(fo)) < 1e-3
    # written out: the spheres at the top level under the composed transforms
    pf = gpu_host.PbrtScene(text=text, flatten_instances=True)
    assert len(pf.table("quadrics")) == 4 and len(pf.table("instances")) == 0
    d, blue = base()
    for inst in (_translate(-2.5, 0, 0), _mm(_translate(2, 0, 1), _scale(1, 1.5, 1))):
        d.add_sphere(radius=0.8, material=blue, o2w=_mm(inst, _translate(0, 1, 0)).astype(np.float32))
        d.add_sphere(radius=0.4, material=blue, o2w=_mm(inst, _translate(0, 2.2, 0)).astype(np.float32), z_max=0.2)
    fo, _ = orc.OracleScene(d).render(mode=1)
    ff, _ = pf.render()
    assert np.array_equal(fo[..., 3], ff[..., 3]) and rel_l2(gpu_host.film_to_rgb(ff), orc.film_to_rgb(fo)) < 1e-3
    assert rel_l2(gpu_host.film_to_rgb(ff), gpu_host.film_to_rgb(fp)) < 5e-3   # two roundings of one scene
    # an emitting sphere inside an object, the only emitter of the scene: L where it is seen, black everywhere else
    glow = ('ObjectBegin "glow"\nAreaLightSource "diffuse" "rgb L" [5 4 3]\nTranslate 0 2 0\nShape "sphere" "float radius" 1.5\nObjectEnd\nObjectInstance "glow"\n')
    g = gpu_host.PbrtScene(text=head + glow + 'WorldEnd\n')
    assert g.n_lights() == 0 and len(g.table("emitters")) == 1
    rgb = gpu_host.film_to_rgb(g.render()[0])
    lit = rgb.sum(-1) > 0
    assert 0.03 < lit.mean() < 0.4
    inside = rgb[(rgb[..., 0] > 4.99)]
    assert len(inside) > 50 and np.allclose(inside, np.float32([5, 4, 3]), rtol=2e-5)   # pixels whose samples all see the sphere
    assert rgb[-8:, :, :].max() == 0.0                                              # the floor in front of it receives nothing: nothing samples the emitter
