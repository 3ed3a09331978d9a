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
    assert int(re.search(r"#define MI355_MAX_SENSORS_PER_PASS (\d+)u", hdr).group(1)) == mi.abi.MAX_SENSORS_PER_PASS
    # mi355_multi_film_layout is host arithmetic (no device): the limit the header states is the one the library enforces
    pitch, x0, virt = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert lib.mi355_multi_film_layout(128, mi.abi.MAX_SENSORS_PER_PASS, C.byref(pitch), C.byref(x0), C.byref(virt)) == mi.abi.OK
    assert (pitch.value, x0.value, virt.value) == (160, 16, 160 * mi.abi.MAX_SENSORS_PER_PASS)
    assert lib.mi355_multi_film_layout(128, mi.abi.MAX_SENSORS_PER_PASS + 1, C.byref(pitch), C.byref(x0), C.byref(virt)) == mi.abi.ERR_INVALID


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
  printf("%zu %zu %zu\n", offsetof(mi355_shape_desc, faces), offsetof(mi355_scene_desc, sensors), offsetof(mi355_sensor_desc, lens_data));
  return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(td, "t.c"), "-o", os.path.join(td, "t")])
        out = subprocess.check_output([os.path.join(td, "t")]).decode().split()
    a = mi.abi
    sizes = [C.sizeof(x) for x in (a.TextureDesc, a.BsdfDesc, a.EmitterDesc, a.ShapeDesc, a.SensorDesc, a.SceneDesc, a.RenderParams, a.Counters, a.ParamInfo)]
    assert [int(x) for x in out[:9]] == sizes
    assert [int(x) for x in out[9:]] == [a.ShapeDesc.faces.offset, a.SceneDesc.sensors.offset, a.SensorDesc.lens_data.offset]


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
    with pytest.raises(mi.PluginError):                             # perspective.cpp:147-149: an offset the engine does not carry is refused, not ignored
        mi.load_dict({"type": "scene", "sensor": {"type": "perspective", "principal_point_offset_x": 0.1}})
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


def _worker(rank, world, port, out_dir, width=32, height=32, sample_border=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package, load_oracle
    from conftest import small_cbox
    mi = load_package(); orc = load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = small_cbox(mi, 32)
    d["sensor"]["film"]["width"] = width; d["sensor"]["film"]["height"] = height
    if sample_border:
        d["sensor"]["film"]["sample_border"] = True
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
        def develop(self, scene, film, img, sensor=0): img.copy_(torch.from_numpy(o.develop(film.numpy())))
        def splat_weights(self, scene, p, weight): weight += torch.from_numpy(o.splat_weights(p))
        def backward(self, scene, p, grad_in, weight, grads):
            if p.integrator == mi.abi.INTEGRATOR_PTRACER: grads += torch.from_numpy(o.ptracer_backward(p, grad_in.numpy(), threads=2))
            else: grads += torch.from_numpy(o.render_backward(p, grad_in.numpy(), weight.numpy(), threads=2))
        def param_total(self, scene): return o.param_layout()[1]
        def synchronize(self, device): pass

    integ = mi.PRBIntegrator({"max_depth": 4}); integ.backend = OracleBackend()
    img = integ.render(scene, seed=3, spp=4, shard=(rank, world))
    gin = np.random.default_rng(5).normal(size=(height, width, 3)).astype(np.float32)
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


def test_three_rank_sharding_ragged_tiles_and_sample_border(mi, orc, tmp_path):
    """world_size = 3 over gloo on a 40 x 56 film with `sample_border`: the sampled grid is (40 + 2 b) x (56 + 2 b) with b = the filter's
    border, i.e. 60 rows of 44 lanes = three 16-row tiles and a ragged fourth of 12 rows, dealt 0, 1, 2, 0 -- rank 0 owns two tiles of
    different size, ranks 1 and 2 one each.  Image, PRB gradient, light-traced image and its gradient are the whole job's on every rank."""
    import torch.multiprocessing as mp
    from conftest import small_cbox
    W, H = 40, 56
    d = small_cbox(mi, 32); d["sensor"]["film"]["width"] = W; d["sensor"]["film"]["height"] = H; d["sensor"]["film"]["sample_border"] = True
    film = mi.load_dict(d).sensor.film
    b = film.border_size if hasattr(film, "border_size") else int(np.ceil(film.rfilter.radius - 0.5))
    assert b >= 1
    gw, gh = W + 2 * b, H + 2 * b
    tiles = [mi.shard_tiles(gw, gh, r, 3) for r in range(3)]
    assert sum(t[1] for t in tiles) == gw * gh and len({t[1] for t in tiles}) >= 2 and (gh % 16) != 0      # ragged: the ranks' lane counts differ
    port = _free_port()
    mp.start_processes(_worker, args=(3, port, str(tmp_path), W, H, True), nprocs=3, join=True, start_method="spawn")
    img_ref = np.load(tmp_path / "img_ref.npy"); g_ref = np.load(tmp_path / "g_ref.npy")
    pimg_ref = np.load(tmp_path / "pimg_ref.npy"); pg_ref = np.load(tmp_path / "pg_ref.npy")
    assert img_ref.shape == (H, W, 3) and img_ref.max() > 0 and np.abs(g_ref).max() > 0 and pimg_ref.max() > 0 and np.abs(pg_ref).max() > 0
    for r in range(3):
        assert np.allclose(np.load(tmp_path / ("img%d.npy" % r)), img_ref, rtol=1e-5, atol=1e-6)
        assert np.allclose(np.load(tmp_path / ("g%d.npy" % r)), g_ref, rtol=1e-4, atol=1e-6)
        assert np.allclose(np.load(tmp_path / ("pimg%d.npy" % r)), pimg_ref, rtol=1e-4, atol=1e-6)
        assert np.allclose(np.load(tmp_path / ("pg%d.npy" % r)), pg_ref, rtol=1e-3, atol=1e-5 * np.abs(pg_ref).max())


def test_huge_jobs_are_split_into_passes(mi):
    """W * H * spp above 2^32: the reference divides the job into passes on its own (integrator.cpp:247-265); so does the host mirror."""
    sc = mi.load_dict({"type": "scene", "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4096, "height": 4096}},
                       "r": {"type": "rectangle"}})
    p = mi.PathIntegrator({"max_depth": 3})._params(sc, 0, 1024)
    assert p.passes >= 5 and 1024 % p.passes == 0 and 4096 * 4096 * (1024 // p.passes) <= 0xffffffff
    assert mi.PathIntegrator({"max_depth": 3})._params(sc, 0, 64).passes in (0, 1)


# ------------------------------------------------------------------------------------------------
# round 2: geometry adjoint (host check), optimizers, lens-design helpers, oracle geometry derivative
# ------------------------------------------------------------------------------------------------
def test_geo_adjoint_matches_finite_differences(tmp_path):
    """csrc/mi355_geo_adjoint.h -- the hand-written reverse mode of one light-path vertex the GPU kernel runs in fp32 -- compiled for the
    host in double precision and checked function by function against central finite differences (tests/geo_adjoint_check.cpp)."""
    import subprocess
    exe = str(tmp_path / "geo_adjoint_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "geo_adjoint_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) >= 28 and all(l.rstrip().endswith("ok") for l in lines), r.stdout


def test_optimizers_follow_the_reference(mi):
    """mi.ad.SGD / mi.ad.Adam as src/python/python/ad/optimizers.py:116-302 states them, including two quirks: SGD with momentum steps
    with the PREVIOUS velocity (:163-170, first step zero), Adam advances t[k] before it looks at the gradient (:265-273) and reset()
    zeroes moments and t."""
    import torch
    x0 = torch.tensor([1.0, -2.0])
    g1, g2 = torch.tensor([0.5, 1.0]), torch.tensor([-1.0, 0.25])
    opt = mi.SGD(lr=0.1, momentum=0.9); opt["x"] = x0
    opt["x"].grad = g1.clone(); opt.step()
    assert torch.allclose(opt["x"].detach(), x0)                             # step = lr * state(previous) = 0
    opt["x"].grad = g2.clone(); opt.step()
    assert torch.allclose(opt["x"].detach(), x0 - 0.1 * g1)                  # the velocity of step 1, applied at step 2
    assert torch.allclose(opt.state["x"], 0.9 * g1 + g2)
    plain = mi.SGD(lr=0.1); plain["x"] = x0; plain["x"].grad = g1.clone(); plain.step()
    assert torch.allclose(plain["x"].detach(), x0 - 0.1 * g1)
    masked = mi.SGD(lr=0.1, momentum=0.5, mask_updates=True); masked["x"] = x0
    masked["x"].grad = torch.tensor([0.0, 1.0]); masked.step(); masked["x"].grad = torch.tensor([0.0, 1.0]); masked.step()
    assert masked["x"][0] == x0[0] and masked.state["x"][0] == 0
    ad = mi.Adam(lr=0.01); ad["x"] = x0
    ad.step()                                                                # no gradient: t advances, nothing else
    assert ad.t["x"] == 1 and torch.allclose(ad["x"].detach(), x0)
    ad["x"].grad = g1.clone(); ad.step()
    t = 2; m = 0.1 * g1; v = 0.001 * g1 * g1
    lr_t = 0.01 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
    assert ad.t["x"] == 2 and torch.allclose(ad["x"].detach(), x0 - lr_t * m / (v.sqrt() + 1e-8), atol=1e-7)
    ad["x"] = torch.tensor([1.0, 2.0, 3.0])                                  # shape change: state and t are reset
    assert ad.t["x"] == 0 and ad.state["x"][0].shape == (3,) and float(ad.state["x"][0].abs().sum()) == 0
    ad["x"] = torch.tensor([4.0, 5.0, 6.0])                                  # same shape: state kept
    ad["x"].grad = torch.ones(3); ad.step(); assert ad.t["x"] == 1
    ad["x"] = torch.tensor([0.0, 0.0, 0.0]); assert ad.t["x"] == 1
    with pytest.raises(Exception):
        ad["y"] = 3


def test_lens_design_geometry_helpers(mi):
    """notebooks/scripts/lens_design/geometry.py restated (mitsuba3-camera_amd/lens_design.py): icosphere counts, the hemisphere cut, the
    ordered boundary loop, the spherical cap's radius and sag, the surface + baffle meshes of create_surface_geometry."""
    ld = mi.lens_design
    for n in range(4):
        V, F = ld.icosphere(n)
        assert V.shape == (10 * 4 ** n + 2, 3) and F.shape == (20 * 4 ** n, 3) and np.allclose(np.linalg.norm(V, axis=1), 1)
        e = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), axis=1)
        assert (np.unique(e, axis=0, return_counts=True)[1] == 2).all()      # closed manifold
        nrm = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]]); assert ((nrm * V[F].mean(axis=1)).sum(axis=1) > 0).all()   # outward
    V, F = ld.get_hemisphere_ico(3)
    loop = ld.boundary_loop(F)
    assert V[:, 2].min() > -0.2 and len(loop) == len(set(loop.tolist())) and np.abs(V[loop, 2]).max() < 0.2
    edges = {(int(a), int(b)) for f in F for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0]))}
    assert all((int(loop[i]), int(loop[(i + 1) % len(loop)])) in edges for i in range(len(loop)))
    surf = ld.EvenAsphericSurface(radial_extent=8.0, c=1 / 24, K=0.0, z0=52.0, a4=1e-6)
    Vl, Fl, Va, Fa, mask = ld.create_surface_geometry(3, 8.0, surf.compute_z_np, c=surf.get_curvature(), baffle_radius=8.8, flip_normals=True)
    r = np.linalg.norm(Vl[:, :2], axis=1)
    # (the hemisphere is cut along icosphere edges: its rim zig-zags between two rings of vertices, as in the fork's own construction)
    assert abs(r.max() - 8.0) < 1e-6 and r[mask].min() > 0.9 * 8.0 and r[~mask].max() < r[mask].max() and mask.sum() == len(ld.boundary_loop(Fl))
    sag = 52.0 - (r ** 2 / 24) / (1 + np.sqrt(1 - r ** 2 / 576)) - 1e-6 * r ** 4
    assert np.allclose(Vl[:, 2], sag, atol=2e-4)
    n = np.cross(Vl[Fl[:, 1]] - Vl[Fl[:, 0]], Vl[Fl[:, 2]] - Vl[Fl[:, 0]]); assert (n[:, 2] < 0).all()          # flipped: facing the film
    assert Va.shape[0] == 2 * mask.sum() and Fa.shape[0] == 4 * mask.sum() and np.abs(Va[:, :2]).max() <= 8.8 + 1e-9
    # the torch sag function is the one the optimisation differentiates: d z / d z0 = 1, d z / d c < 0 off axis
    import torch
    z0 = torch.tensor([52.0], requires_grad=True); c = torch.tensor([8.0 / 24], requires_grad=True)
    surf.params["z0"] = z0; surf.params["c"] = c
    z = surf.compute_z(torch.tensor([0.0, 4.0]), torch.tensor([0.0, 0.0]))
    gz0, gc = torch.autograd.grad(z.sum(), [z0, c])
    assert abs(float(gz0) - 2.0) < 1e-6 and float(gc) < 0
    # angle-weighted vertex normals (mesh.cpp:376-413) of the cap point along the analytic normal of the sphere
    vn = mi.scene.compute_vertex_normals(Vl.astype(np.float32), Fl.astype(np.uint32))
    centre = np.array([0, 0, 52.0 - 24.0]); ana = (centre - Vl); ana /= np.linalg.norm(ana, axis=1, keepdims=True)
    inner = ~mask
    assert (np.abs((vn[inner] * ana[inner]).sum(axis=1)) > 0.999).all()


def test_oracle_geometry_derivative_against_finite_differences(mi, orc):
    """The oracle's forward-mode (multi-dual) geometry derivative of the particle tracer (oracle/mi_oracle_geo.inc) against central finite
    differences of the oracle's own primal, with identical seeds: a flat tilted glass interface (two triangles) between a collimated disk
    source and a diffuse screen -- vertex gradient contracted with the tilt, and the IOR gradient.  The image functional vanishes towards
    the frame (film-border splats are a boundary term that AD does not see)."""
    T = mi.ScalarTransform4f
    res = 24

    def wedge(ior, tilt):
        q = np.array([[-0.8, -0.8, 1.0 + 0.8 * tilt], [0.8, -0.8, 1.0 - 0.8 * tilt], [0.8, 0.8, 1.0 - 0.8 * tilt], [-0.8, 0.8, 1.0 + 0.8 * tilt]], dtype=np.float32)
        return {'type': 'scene', 'white': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': (0.8, 0.7, 0.6)}}, 'black': {'type': 'diffuse', 'reflectance': 0.0},
                'screen': {'type': 'rectangle', 'to_world': T.look_at(target=[0, 0, 1], origin=[0, 0, 0], up=[0, 1, 0]).scale([2, 2, 1]), 'bsdf': {'type': 'ref', 'id': 'white'}},
                'lens': {'type': 'mesh', 'vertex_positions': q, 'faces': np.array([[0, 2, 1], [0, 3, 2]], dtype=np.uint32),
                         'bsdf': {'type': 'dielectric', 'int_ior': ior, 'ext_ior': 1.0, 'specular_reflectance': 0.0}},
                'source': {'type': 'disk', 'to_world': T.look_at(target=[0, 0, 1.0], origin=[0, 0, 3.0], up=[0, 1, 0]) @ T.scale([0.2, 0.2, 1]),
                           'bsdf': {'type': 'ref', 'id': 'black'}, 'emitter': {'type': 'directionalarea', 'radiance': 1.0}},
                'sensor': {'type': 'perspective', 'near_clip': 0.01, 'far_clip': 10.0, 'fov': 60.0, 'to_world': T.look_at(target=[0.1, 0.05, 0.0], origin=[0.3, 0.2, 0.6], up=[0, 1, 0]),
                           'film': {'type': 'hdrfilm', 'width': res, 'height': res, 'pixel_format': 'rgb', 'rfilter': {'type': 'gaussian'}}}}
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 5})
    yy, xx = np.mgrid[0:res, 0:res]
    window = (np.sin(np.pi * (xx + 0.5) / res) * np.sin(np.pi * (yy + 0.5) / res)) ** 2
    gin = np.repeat(((0.5 + xx / res + 0.3 * yy / res) * window)[:, :, None], 3, axis=2).astype(np.float32)
    spp, ior0, tilt0, eps = 512, 1.6, 0.35, 4e-3

    def f(ior, tilt):
        o = orc.OracleScene(mi.load_dict(wedge(ior, tilt)), mi.abi)
        return float((o.ptracer_render(o.params(integ, 4, spp))[0].astype(np.float64) * gin).sum())
    sc = mi.load_dict(wedge(ior0, tilt0)); o = orc.OracleScene(sc, mi.abi)
    p = o.params(integ, 4, spp); p.flags |= mi.abi.FLAG_GEOMETRY_GRADS
    g = o.ptracer_backward(p, gin)
    lay, _ = o.param_layout()
    off_v = [o_ for k, i, o_, s in lay if k == mi.abi.PARAM_VERTEX_POSITIONS and sc.shapes[i].id == "lens"][0]
    off_i = [o_ for k, i, o_, s in lay if k == mi.abi.PARAM_BSDF_IOR][0]
    dq = np.array([[0, 0, 0.8], [0, 0, -0.8], [0, 0, -0.8], [0, 0, 0.8]], dtype=np.float64)
    g_tilt, fd_tilt = float((g[off_v:off_v + 12].reshape(4, 3) * dq).sum()), (f(ior0, tilt0 + eps) - f(ior0, tilt0 - eps)) / (2 * eps)
    g_ior, fd_ior = float(g[off_i]), (f(ior0 + eps, tilt0) - f(ior0 - eps, tilt0)) / (2 * eps)
    assert abs(fd_tilt) > 0 and abs(g_tilt - fd_tilt) < 4e-2 * abs(fd_tilt), (g_tilt, fd_tilt)
    assert abs(fd_ior) > 0 and abs(g_ior - fd_ior) < 6e-2 * abs(fd_ior), (g_ior, fd_ior)       # (the finite difference sees ~100 lobe flips: noisy)
    assert g[off_i + 1] != 0                                                 # ext_ior: the ratio's denominator


def test_baseline_config_1_runs_at_its_stated_size(mi, orc):
    """BASELINE configs[0]: the Cornell box at 256 x 256 @ 16 spp on the host CPU ("plumbing, no GPU") -- the oracle at that size, whole image,
    with the sanity identities of a converging render: finite, the light is the brightest thing, red bleeds left and green right, two
    disjoint halves of the samples agree within Monte-Carlo noise."""
    d = mi.cornell_box()      # 256 x 256 as mi.cornell_box() ships it (util.py:565-699)
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
    integ = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5})
    img, n = o.render(o.params(integ, 0, 16))
    assert img.shape == (256, 256, 3) and n == 256 * 256 * 16 and np.isfinite(img).all()
    assert img[30:36, 112:144].mean() > 20 * img[128:, :].mean()             # the emitter, top centre
    left, right = img[100:200, 8:40], img[100:200, 216:248]
    assert left[..., 0].mean() > 2 * left[..., 1].mean() and right[..., 1].mean() > 2 * right[..., 0].mean()
    a, _ = o.render(o.params(integ, 1, 8)); b, _ = o.render(o.params(integ, 2, 8))
    assert abs(a.mean() - b.mean()) < 0.02 * a.mean() and abs(0.5 * (a + b).mean() - img.mean()) < 0.02 * img.mean()


def test_bench_roofline_reads_the_committed_pmc_pass():
    """bench.py's roofline blocks are computed from the committed PMC passes (profiles/r03_pmc*_summary.json, one per leg) and the committed
    microbenchmark (profiles/r02_valu_issue.json).  The passes must have been taken from the kernel sources of this checkout (the sha256 over
    csrc/ they record equals the one bench.py computes: a later kernel edit without a new pass makes every fraction `pmc_stale`), the kernels
    bench.py looks up must be in those files under the names the current build gives them, the VALU-issue fraction must be physical
    (0 < frac <= 1) at the kernel times of the committed bench line, and the all-instruction issue model must sit between the VALU-only
    figure and 1.  Every leg of the committed line carries a block that can be recomputed from these files."""
    import importlib.util, json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    line = json.load(open(os.path.join(ROOT, "profiles", "r03_final_bench.json")))
    sha = bench.csrc_sha256()
    for f in [bench.PMC_FILE] + list(bench.PMC_FILES.values()):
        d = json.load(open(os.path.join(ROOT, f)))
        assert d["csrc_sha256"] == sha, "%s was taken from other kernel sources: re-run tools/pmc.sh" % f
        assert not bench.pmc_stale(f)
    for pmc_name, ms, lanes, f in (("k_render_compact<0, 0>", line["roofline"]["kernel_ms"], 512 * 512 * 256, None),
                                   ("k_prb_backward_1p<0, false>", line["prb"]["kernel_ms"], 512 * 512 * 64, None),
                                   ("k_ptracer<0, false>", line["ptracer"]["box"]["kernel_ms"], 512 * 512 * 64, bench.PMC_FILES["ptracer"]),
                                   ("k_ptracer<1, false>", line["lens"]["roofline"]["render"]["kernel_ms"], line["lens"]["roofline"]["render"]["pmc"]["lanes_profiled"], bench.PMC_FILES["lens"]),
                                   ("k_ptracer_geo_bwd<1, false>", line["lens"]["roofline"]["geometry_adjoint"]["kernel_ms"], line["lens"]["roofline"]["geometry_adjoint"]["pmc"]["lanes_profiled"], bench.PMC_FILES["lens"])):
        assert bench.pmc_kernel(pmc_name, f) is not None, pmc_name
        r = bench.valu_roofline(pmc_name, pmc_name, ms, lanes, 1.0, f)
        assert "pmc_stale" not in r
        assert r["bound"] == "valu_issue" and 0.0 < r["frac"] <= 1.0 and r["traffic"] > 0 and r["hbm"]["frac_of_peak"] < 1.0
        assert abs(r["frac"] * r["cycles_per_inst_measured"] - bench.VALU_CYCLES_PEAK) < 0.01          # frac = peak cycles / measured cycles
        assert r["frac"] < r["frac_of_peak_at_sustained_clock"] <= 1.0                                  # the sustained clock is below the nominal one
        im = r["issue_model"]
        assert r["frac"] < im["frac_of_model"] <= 1.0 and im["valu"] == r["insts_per_launch"] and im["scalar"] > 0
    # the regenerating kernel runs a persistent grid: its lanes come from the workload, not from the grid
    r = bench.valu_roofline("k_render_regen<2>", "k_render_regen<2>", line["veach"]["kernel_ms"], 1024 * 1024 * 512, 1.0, bench.PMC_FILES["veach"], lanes_profiled=1024 * 1024 * 512)
    assert 0.0 < r["frac"] <= 1.0 and r["pmc"]["scaled_by_lanes"] == 1.0
    # a pass from other sources is flagged and states no fraction
    stale = os.path.join(ROOT, "profiles", "r02_pmc_summary.json")
    if os.path.exists(stale):
        r = bench.valu_roofline("k", "k_render_compact<0, 0>", 9.0, 512 * 512 * 256, 1.0, "profiles/r02_pmc_summary.json")
        assert r.get("pmc_stale") is True and r["frac"] is None
    # the committed line itself carries what the contract asks for
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["roofline"]["frac"] <= 1.0 and line["cpu_baseline"]["kind"] == "port" and line["config"]["workload"].startswith("Cornell box")
    for blk in (line["roofline"], line["prb"]["roofline"], line["veach"]["roofline"], line["ptracer"]["box"]["roofline"], line["ptracer"]["gaussian"]["roofline"],
                line["lens"]["roofline"]["render"], line["lens"]["roofline"]["geometry_adjoint"]):
        assert 0.0 < blk["frac"] <= 1.0 and blk["traffic"] > 0 and blk["pmc"]["file"].startswith("profiles/r03_pmc"), blk["kernel"]
        assert blk["contract"]["algorithmic_bytes_per_launch"] > 0 and "pmc_stale" not in blk
    for leg in ("prb", "veach", "ptracer", "lens"):
        assert leg in line, leg
