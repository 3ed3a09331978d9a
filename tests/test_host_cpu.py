"""CPU-only tests of the host side: the C-ABI library loads and exports every declared symbol, the dict
loader mirrors Mitsuba's error behaviour, and the multi-GPU decomposition (pixel bands + all-reduce of film,
weights and gradients) is correct with world_size = 2 over gloo."""
import ctypes as C
import os
import re
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(mi):
    lib = mi.abi.lib()
    hdr = open(os.path.join(ROOT, "include", "mi355.h")).read()
    declared = set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in mi.abi.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.mi355_abi_version() == mi.abi.ABI_VERSION


def test_struct_layouts_match_header(mi):
    """ctypes mirrors vs the C compiler's view of include/mi355.h (sizes + a few offsets)."""
    import subprocess, tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "mi355.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mi355_texture_desc), sizeof(mi355_bsdf_desc), sizeof(mi355_emitter_desc),
         sizeof(mi355_shape_desc), sizeof(mi355_sensor_desc), sizeof(mi355_scene_desc), sizeof(mi355_render_params),
         sizeof(mi355_counters), sizeof(mi355_param_info));
  printf("%zu %zu %zu\n", offsetof(mi355_shape_desc, faces), offsetof(mi355_scene_desc, sensor), offsetof(mi355_sensor_desc, lens_data));
  return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(td, "t.c"), "-o", os.path.join(td, "t")])
        out = subprocess.check_output([os.path.join(td, "t")]).decode().split()
    a = mi.abi
    sizes = [C.sizeof(x) for x in (a.TextureDesc, a.BsdfDesc, a.EmitterDesc, a.ShapeDesc, a.SensorDesc, a.SceneDesc, a.RenderParams, a.Counters, a.ParamInfo)]
    assert [int(x) for x in out[:9]] == sizes
    assert [int(x) for x in out[9:]] == [a.ShapeDesc.faces.offset, a.SceneDesc.sensor.offset, a.SensorDesc.lens_data.offset]


def test_no_gpu_means_loud_failure(mi):
    """Without a device the product path must refuse to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    scene = mi.load_dict(mi.cornell_box())
    with pytest.raises(mi.Mi355Error) as e:
        scene.handle(0)
    assert e.value.code == mi.abi.ERR_NO_DEVICE


def test_load_dict_errors_and_scope(mi):
    d = mi.cornell_box()
    sc = mi.load_dict(d)
    assert len(sc.shapes) == 8 and len(sc.bsdfs) == 3 and len(sc.emitters) == 1
    assert sc.sensor.film.size() == (256, 256) and sc.sensor.sample_count == 64
    assert abs(sc.sensor.x_fov - 39.3077) < 1e-4
    assert sc.integrator().max_depth == 8 and sc.integrator().rr_depth == 5
    assert mi.PRBIntegrator().max_depth == 6                        # common.py:31
    for bad in ({"type": "cylinder"}, {"type": "frobnicator"}):
        with pytest.raises(mi.PluginError):
            mi.load_dict({"type": "scene", "x": bad})
    with pytest.raises(mi.PluginError):                             # reflectance outside [0,1] (srgb.cpp:52-66)
        mi.load_dict({"type": "scene", "b": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [1.5, 0, 0]}}})
    with pytest.raises(mi.PluginError):                             # dangling reference
        mi.load_dict({"type": "scene", "r": {"type": "rectangle", "bsdf": {"type": "ref", "id": "nope"}}})
    with pytest.raises(mi.PluginError):                             # camera scale (perspective.cpp:144-145)
        mi.load_dict({"type": "scene", "sensor": {"type": "perspective", "to_world": mi.ScalarTransform4f.scale(2.0)}})
    with pytest.raises(mi.PluginError):
        mi.PathIntegrator({"rr_depth": 0})
    with pytest.raises(mi.PluginError):
        mi.load_dict({"type": "scene", "t": {"type": "diffuse", "reflectance": {"type": "bitmap", "data": np.zeros((1, 4, 3))}}})
    # Appendix C of SURVEY.md: resolved world-space geometry of the fixture
    light = [s for s in sc.shapes if s.id == "light"][0]
    assert np.allclose(light.to_world.transform_affine([0, 0, 0]), [0, 0.99, 0.01], atol=1e-6)
    assert np.allclose(light.to_world.transform_normal([0, 0, 1]) / np.linalg.norm(light.to_world.transform_normal([0, 0, 1])), [0, -1, 0], atol=1e-6)
    assert abs(light.surface_area() - 0.1748) < 1e-4
    lb = [s for s in sc.shapes if s.id == "large-box"][0]
    lo, hi = lb.bbox()
    assert abs(lo[1] + 1.01) < 1e-5 and abs(hi[1] - 0.21) < 1e-5


def test_obj_loader(mi, tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\n")
    sc = mi.load_dict({"type": "scene", "q": {"type": "obj", "filename": str(p), "to_world": mi.ScalarTransform4f.translate([0, 0, 2])}})
    s = sc.shapes[0]
    assert s.primitive_count() == 2 and s.vertex_positions.shape == (4, 3) and np.allclose(s.vertex_positions[:, 2], 2)
    assert np.isclose(s.surface_area(), 1.0) and s.vertex_normals is not None and s.vertex_texcoords is not None


def test_ply_loader(mi, tmp_path):
    """ASCII and binary PLY (src/shapes/ply.cpp's buffer format): same quad as the OBJ test, a skipped extra property, a fan."""
    import struct
    v = [(0, 0, 0, 0, 0), (1, 0, 0, 1, 0), (1, 1, 0, 1, 1), (0, 1, 0, 0, 1)]
    hdr = "ply\nformat %s 1.0\ncomment test\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n" \
          "property float u\nproperty float v\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"
    a = tmp_path / "a.ply"
    a.write_text(hdr % "ascii" + "".join("%g %g %g 7 %g %g\n" % (x, y, z, s, t) for x, y, z, s, t in v) + "4 0 1 2 3\n")
    b = tmp_path / "b.ply"
    b.write_bytes((hdr % "binary_little_endian").encode() + b"".join(struct.pack("<fffBff", x, y, z, 7, s, t) for x, y, z, s, t in v)
                  + struct.pack("<Biiii", 4, 0, 1, 2, 3))
    for p in (a, b):
        sc = mi.load_dict({"type": "scene", "q": {"type": "ply", "filename": str(p), "to_world": mi.ScalarTransform4f.translate([0, 0, 2])}})
        s = sc.shapes[0]
        assert s.primitive_count() == 2 and s.vertex_positions.shape == (4, 3) and np.allclose(s.vertex_positions[:, 2], 2)
        # ply.cpp:444-446: a file without normals gets angle-weighted vertex normals (recompute_vertex_normals) unless face_normals is set
        assert np.isclose(s.surface_area(), 1.0) and np.allclose(s.vertex_normals, [0, 0, 1], atol=1e-6) and np.allclose(s.vertex_texcoords[2], [1, 1])
        sf = mi.load_dict({"type": "scene", "q": {"type": "ply", "filename": str(p), "face_normals": True}}).shapes[0]
        assert sf.vertex_normals is None
    bad = tmp_path / "bad.ply"; bad.write_text("plx\n")
    with pytest.raises(mi.PluginError):
        mi.load_dict({"type": "scene", "q": {"type": "ply", "filename": str(bad)}})


XML_SCENE = """<scene version="3.0.0">
    <default name="spp" value="8"/>
    <default name="res" value="12"/>
    <integrator type="path"><integer name="max_depth" value="4"/></integrator>
    <sensor type="perspective" id="sensor">
        <float name="fov" value="45"/><float name="near_clip" value="0.01"/><float name="far_clip" value="50"/>
        <transform name="to_world"><lookat origin="0, 0, 4" target="0, 0, 0" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sample_count" value="$spp"/></sampler>
        <film type="hdrfilm"><integer name="width" value="$res"/><integer name="height" value="$res"/>
            <rfilter type="tent"/><string name="pixel_format" value="rgb"/></film>
    </sensor>
    <bsdf type="diffuse" id="grey"><rgb name="reflectance" value="0.5"/></bsdf>
    <bsdf type="conductor" id="mirror"/>
    <shape type="obj" id="lamp"><string name="filename" value="quad.obj"/>
        <transform name="to_world"><scale value="0.5"/><rotate x="1" angle="90"/><translate x="0" y="1.5" z="0"/></transform>
        <ref id="grey"/><emitter type="area"><rgb name="radiance" value="10, 8, 6"/></emitter></shape>
    <shape type="rectangle" id="floor"><transform name="to_world"><rotate x="1" angle="-90"/><scale x="3" y="3" z="3"/><translate y="-1"/></transform>
        <ref id="grey"/></shape>
    <shape type="sphere"><transform name="to_world"><scale value="0.4"/><translate x="0.2" y="-0.6"/></transform><ref id="mirror"/></shape>
</scene>"""


def test_load_file_xml(mi, orc, tmp_path):
    """mi.load_file on the XML subset of src/core/xml.cpp: $defaults (overridable), nested plugins, <ref>, <rgb>, <transform>
    whose children apply in document order (scale, then rotate, then translate), filenames relative to the XML file."""
    (tmp_path / "quad.obj").write_text("v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nf 1 2 3 4\n")
    (tmp_path / "scene.xml").write_text(XML_SCENE)
    sc = mi.load_file(str(tmp_path / "scene.xml"), res=10)
    assert sc.sensor.film.size() == (10, 10) and sc.sensor.sample_count == 8 and sc.integrator().max_depth == 4
    assert [b.type for b in sc.bsdfs] == [mi.abi.BSDF_DIFFUSE, mi.abi.BSDF_CONDUCTOR] and len(sc.emitters) == 1
    lamp = [s for s in sc.shapes if s.id == "lamp"][0]
    assert lamp.primitive_count() == 2 and np.allclose(lamp.vertex_positions[:, 1], 1.5, atol=1e-6) and np.isclose(lamp.surface_area(), 1.0)
    assert np.allclose(np.abs(lamp.vertex_positions[:, [0, 2]]), 0.5, atol=1e-6)
    ball = [s for s in sc.shapes if s.id not in ("lamp", "floor")][0]
    assert np.allclose(ball.to_world.transform_affine([0, 0, 0]), [0.2, -0.6, 0.0], atol=1e-6)
    o = orc.OracleScene(sc, mi.abi)
    img, _ = o.render(o.params(sc.integrator(), 0, 8))
    assert np.isfinite(img).all() and img.mean() > 0.01
    for bad in ("<scene version='3.0.0'><shape type='rectangle'><float value='1'/></shape></scene>",
                "<scene version='3.0.0'><bsdf type='diffuse'><rgb name='reflectance' value='$nope'/></bsdf></scene>",
                "<scene version='3.0.0'><include filename='x.xml'/></scene>"):
        (tmp_path / "bad.xml").write_text(bad)
        with pytest.raises(mi.PluginError):
            mi.load_file(str(tmp_path / "bad.xml"))


@pytest.mark.skipif(not os.path.exists("/root/reference/notebooks/scenes/cboxes/cbox_path.xml"), reason="reference tree not present")
def test_reference_fixture_scenes_load_and_render(mi, orc):
    """SURVEY 8(f)-1 names notebooks/scenes/cboxes/{cbox_path,cbox_real}.xml as the fork's fixture scenes: OBJ walls, a mesh
    luminaire, a `conductor` and a `dielectric` sphere, tent filter, the realisticlens sensor (tessar / singlet), path / ptracer.
    They load through mi.load_file as they are and render in the oracle (CPU-only check: the reference tree is not on the GPU box)."""
    base = "/root/reference/notebooks/scenes/cboxes/"
    a = mi.load_file(base + "cbox_path.xml", res=12, spp=8)
    assert a.integrator().name == "path" and a.integrator().max_depth == -1 and a.sensor.lens_table.shape == (8, 16)
    assert len(a.shapes) == 8 and len(a.emitters) == 1 and a.shapes[[s.id for s in a.shapes].index("light")].primitive_count() == 2
    oa = orc.OracleScene(a, mi.abi)
    img, _ = oa.render(oa.params(mi.PathIntegrator({"max_depth": 5}), 0, 8))
    assert np.isfinite(img).all() and img.mean() > 0.005
    b = mi.load_file(base + "cbox_real.xml", res=12, spp=8)
    assert b.integrator().name == "ptracer" and b.sensor.shutter_open_time == 30.0
    ob = orc.OracleScene(b, mi.abi)
    img2, lanes = ob.ptracer_render(ob.params(mi.ParticleTracerIntegrator({"max_depth": 4}), 0, 16))
    assert lanes == 12 * 12 * 16 and np.isfinite(img2).all() and img2.max() > 0


def test_shard_pixels_partition(mi):
    for n, w in [(1024, 1), (1024, 3), (1000, 8), (7, 8)]:
        cover = np.zeros(n, dtype=int)
        for r in range(w):
            b, c = mi.shard_pixels(n, r, w)
            cover[b:b + c] += 1
        assert (cover == 1).all()


def test_shard_tiles_partition(mi):
    """Interleaved row tiles: every pixel exactly once, through the mapping mi355.h documents for pixel_tile."""
    for (w, h, world, rows) in [(512, 512, 8, 16), (32, 32, 4, 3), (16, 16, 3, 2), (20, 7, 2, 16), (64, 40, 5, 16)]:
        cover = np.zeros(w * h, dtype=int)
        for r in range(world):
            b, n, tile, stride = mi.shard_tiles(w, h, r, world, rows=rows)
            q = np.arange(n)
            cover[b + (q // tile) * stride + q % tile] += 1
        assert (cover == 1).all(), (w, h, world, rows)
    class F: crop_width = 512; crop_height = 512
    assert len(mi.shard_of(F, 1, 8)) == 4                      # tiles when every rank gets one
    F.crop_height = 16
    assert len(mi.shard_of(F, 1, 8)) == 2                      # else contiguous bands


# ---- world_size = 2 over gloo --------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package, load_oracle
    from conftest import small_cbox
    mi = load_package(); orc = load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = small_cbox(mi, 32)
    tex = np.random.default_rng(12345).uniform(0.2, 0.8, (4, 4, 3)).astype(np.float32)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    scene = mi.load_dict(d)
    o = orc.OracleScene(scene, mi.abi)

    class OracleBackend:
        """Stands in for the HIP back-end so that the sharding + collectives of integrators.py run on CPU."""
        name = "oracle-test"
        def device(self, scene): return torch.device("cpu")
        def accumulate(self, scene, p, film):
            film += torch.from_numpy((o.ptracer_render if p.integrator == mi.abi.INTEGRATOR_PTRACER else o.render)(p, threads=2, develop=False)[0])
        def develop(self, scene, film, img): img.copy_(torch.from_numpy(o.develop(film.numpy())))
        def splat_weights(self, scene, p, weight): weight += torch.from_numpy(o.splat_weights(p))
        def backward(self, scene, p, grad_in, weight, grads):
            if p.integrator == mi.abi.INTEGRATOR_PTRACER: grads += torch.from_numpy(o.ptracer_backward(p, grad_in.numpy(), threads=2))
            else: grads += torch.from_numpy(o.render_backward(p, grad_in.numpy(), weight.numpy(), threads=2))
        def param_total(self, scene): return o.param_layout()[1]
        def synchronize(self, device): pass

    integ = mi.PRBIntegrator({"max_depth": 4}); integ.backend = OracleBackend()
    img = integ.render(scene, seed=3, spp=4, shard=(rank, world))
    gin = np.random.default_rng(5).normal(size=(32, 32, 3)).astype(np.float32)
    g = integ.render_backward(scene, None, torch.from_numpy(gin), seed=9, spp=4, shard=(rank, world))
    np.save(os.path.join(out_dir, "img%d.npy" % rank), img.numpy()); np.save(os.path.join(out_dir, "g%d.npy" % rank), g.numpy())
    pt = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 3}); pt.backend = OracleBackend()     # the light tracer shards over the FILM grid
    pimg = pt.render(scene, seed=3, spp=4, shard=(rank, world))
    pg = pt.render_backward(scene, None, torch.from_numpy(gin), seed=9, spp=4, shard=(rank, world))
    np.save(os.path.join(out_dir, "pimg%d.npy" % rank), pimg.numpy()); np.save(os.path.join(out_dir, "pg%d.npy" % rank), pg.numpy())
    if rank == 0:
        wpt = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 3}); wpt.backend = OracleBackend()
        np.save(os.path.join(out_dir, "pimg_ref.npy"), wpt.render(scene, seed=3, spp=4).numpy())
        np.save(os.path.join(out_dir, "pg_ref.npy"), wpt.render_backward(scene, None, torch.from_numpy(gin), seed=9, spp=4).numpy())
        whole = mi.PRBIntegrator({"max_depth": 4}); whole.backend = OracleBackend()
        np.save(os.path.join(out_dir, "img_ref.npy"), whole.render(scene, seed=3, spp=4).numpy())
        np.save(os.path.join(out_dir, "g_ref.npy"), whole.render_backward(scene, None, torch.from_numpy(gin), seed=9, spp=4).numpy())
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_sharding_over_gloo(mi, orc, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    img_ref = np.load(tmp_path / "img_ref.npy"); g_ref = np.load(tmp_path / "g_ref.npy")
    for r in range(2):
        assert np.allclose(np.load(tmp_path / ("img%d.npy" % r)), img_ref, rtol=1e-5, atol=1e-6)
        assert np.allclose(np.load(tmp_path / ("g%d.npy" % r)), g_ref, rtol=1e-4, atol=1e-6)
    pimg_ref = np.load(tmp_path / "pimg_ref.npy"); pg_ref = np.load(tmp_path / "pg_ref.npy")
    assert pimg_ref.max() > 0 and np.abs(pg_ref).max() > 0
    for r in range(2):          # particle tracer: film shards and gradient shards add up to the whole on every rank
        assert np.allclose(np.load(tmp_path / ("pimg%d.npy" % r)), pimg_ref, rtol=1e-4, atol=1e-6)
        assert np.allclose(np.load(tmp_path / ("pg%d.npy" % r)), pg_ref, rtol=1e-3, atol=1e-5 * np.abs(pg_ref).max())
    assert np.abs(g_ref).max() > 0


def test_huge_jobs_are_split_into_passes(mi):
    """W * H * spp above 2^32: the reference divides the job into passes on its own (integrator.cpp:247-265); so does the host mirror."""
    sc = mi.load_dict({"type": "scene", "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4096, "height": 4096}},
                       "r": {"type": "rectangle"}})
    p = mi.PathIntegrator({"max_depth": 3})._params(sc, 0, 1024)
    assert p.passes >= 5 and 1024 % p.passes == 0 and 4096 * 4096 * (1024 // p.passes) <= 0xffffffff
    assert mi.PathIntegrator({"max_depth": 3})._params(sc, 0, 64).passes in (0, 1)
