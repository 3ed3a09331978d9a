"""CPU tests of the host-side mirror of the fork's lens-design scripts (mitsuba3-camera_amd/lens_design.py) and of the oracle's twin of
the geometric ray tracer `prb_basic.trace` -- what BASELINE config 5 (notebooks/doublet_full_optim.ipynb) is set up with.

Pins (all reference-held):
  * notebooks/doublet_full_optim.ipynb cell 5 ("Exact Vd: 6.861593997805768"), cell 6 (LensSystem.print(): Cauchy pairs, r_c, t) and cell 9
    (optimizer variables lens0_c, lens0_z0 after the focal-length constraint) -- stored OUTPUTS of the reference, float32 Dr.Jit arithmetic;
  * notebooks/scripts/lens_design/lens.py:1270-1282 (test_focal_lengths_singlet: BFL / FFL of two singlets) and :1284-1330 (Petzval and
    Cooke triplets: EFL 49.999855 / 50.000541);
  * include/mitsuba/core/qmc.h:189-237 (radical_inverse_2, sobol_2: the first points of the two sequences);
  * notebooks/scripts/lens_design/design_problem.py:430-472 (__test_geo_tracer: exact trace of the lens MESHES against the paraxial matrices,
    atol 2e-5; cell 7 of the notebook prints what the reference itself measured: u within 8.6e-6 -- "u is close: True" -- and y off by
    3.8e-4 -- "y is close: False").
"""
import math

import numpy as np
import pytest


def _doublet(ld, subdiv, constrained=True):
    surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
    mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
    system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, aperture_index=-1, subdiv_levels=subdiv)
    system.disable_all_materials()
    if constrained:                              # cell 5: "apply focal length constraint"
        s1 = system.get_rear_surface_params(50.0)
        system.elements[0].surface.params['c'] = float(s1['c']); system.elements[0].surface.params['z0'] = float(s1['z0'])
        system.compute_paraxial_quantities()
    return system


def test_doublet_matches_the_notebooks_stored_outputs(mi):
    ld = mi.lens_design
    system = _doublet(ld, 2)
    # cell 9: problem.optimizer.variables (float32 in the reference)
    assert abs(system.elements[0].surface.params['c'] - (-0.04761601984500885)) < 2e-8
    assert abs(system.elements[0].surface.params['z0'] - 46.7470817565918) < 1e-5
    assert abs(system.elements[1].surface.params['c'] - (-0.3333333432674408)) < 1e-7 and abs(system.elements[2].surface.params['c'] - 0.3333333432674408) < 1e-7
    # cell 6: LensSystem.print()
    d = system.describe()
    assert np.allclose(d["materials"][0], (1.456079838959, 0.021086562958), rtol=0, atol=1e-11)
    assert np.allclose(d["materials"][1], (1.568952539042, 0.026305160873), rtol=0, atol=1e-11)
    assert d["materials"][2] == (1.000277, 0.0)
    assert np.allclose(d["r_c"], (24.0, -24.0, -168.01068267), rtol=1e-7)
    assert np.allclose(d["t"], (3.0, 2.25291824, 46.74708176), rtol=0, atol=1e-5)
    assert d["r_e"] == [8.0, 8.0, 8.0]
    # the constraint holds: focal length 50 mm, paraxial focus on the film plane
    assert abs(system.compute_EFL() - 50.0) < 1e-9 and abs(system.compute_BFL() - system.elements[0].surface.params['z0']) < 1e-9
    # cell 4 / 5: get_exact_Vd -> "Exact Vd: 6.861593997805768"
    inv_c = system.elements[0].surface.get_curvature(); inv_b = -system.elements[1].surface.get_curvature(); inv_a = system.elements[2].surface.get_curvature()
    n1, V1, n2 = system.materials[2].params['ior'], system.materials[2].params['V_d'], system.materials[1].params['ior']
    P1 = (n1 - 1) * (inv_a + inv_b); P2 = -(n2 - 1) * (inv_b + inv_c)
    assert abs(-(P2 / P1) * V1 - 6.861593997805768) < 2e-5
    # no aperture stop: the entrance pupil sits on the front surface, the sources aim there (design_problem.py:152-153)
    assert system.entrance_pupil_position == 52.0 and system.size() == 3


def test_focal_lengths_of_the_references_own_lens_tests(mi):
    """lens.py:1258-1330"""
    ld = mi.lens_design

    def singlet(c0, c1):
        s = [ld.ConicSurface(radial_extent=0.8, c=c0, K=0.0, z0=2.5), ld.ConicSurface(radial_extent=0.8, c=c1, K=0.0, z0=2.8)]
        l = ld.LensSystem(s, [ld.LensMaterial("nbk7", 1.5047, 64.17)], subdiv_levels=1)
        return np.array([l.compute_BFL(), l.compute_FFL()])
    assert np.allclose(singlet(-0.3, 0.3), [3.25439588, 3.25439588])
    assert np.allclose(singlet(-0.277226358652119, 0.421164870262146), [2.76563327, 2.80743881])

    # :1284-1330 PRINTS (does not assert) the EFL of two published prescriptions next to the catalogue values 50.000541 / 49.999855; those
    # hold for n_air = 1 -- with the scripts' own air (1.000277) both come out 0.04 mm longer -- so the comparison is made in vacuum
    def compound(surfs, mats):
        surfaces = [ld.ConicSurface(K=0, radial_extent=s[0], c=(0.0 if math.isinf(s[1]) else 1.0 / s[1]), z0=s[2]) for s in surfs]
        l = ld.LensSystem(surfaces, mats, subdiv_levels=1)
        for m in l.materials:
            if m.name == "air":
                m.params['ior'] = 1.0
        l.compute_paraxial_quantities()
        return l
    air = ld.LensMaterial("air")
    sk16, f4 = ld.LensMaterial("sk16", 1.62041, 60.32), ld.LensMaterial("f4", 1.616592, 36.63)
    cooke = compound([(5.0, -17.285, 42.95), (5.0, 141.25, 44.95), (5.0, 19.3, 50.95), (5.0, -20.25, 51.95), (5.0, -158.65, 57.95), (5.0, 21.25, 59.95)], [sk16, air, f4, air, sk16])
    assert abs(cooke.compute_EFL() - 50.000541) < 5e-4
    k7, f2 = ld.LensMaterial("k7", 1.51112, 60.41), ld.LensMaterial("f2", 1.62004, 36.37)
    petzval = compound([(0.8, -188.1, 20.4), (0.8, -22.1, 22.4), (0.8, 28.42, 28.9), (0.8, math.inf, 64.67), (0.8, -37.87, 66.67), (0.8, 36.27, 72.67)], [f2, k7, air, f2, k7])
    assert abs(petzval.compute_EFL() - 49.999855) < 5e-4
    # ... and what the function PRINTS with the scripts' own air is stored in notebooks/design-problem-multi-spot.ipynb (output of cell 5):
    #       BFL matches image plane:  [20.566852569580078] 20.4      [50.02990341] [49.999855]          (Petzval)
    #       BFL matches image plane:  [43.12704086303711] 42.95      [50.04488754] [50.000541]          (Cooke triplet)
    for l in (cooke, petzval):
        for m in l.materials:
            if m.name == "air":
                m.params['ior'] = 1.000277
        l.compute_paraxial_quantities()
    assert abs(petzval.compute_BFL() - 20.566852569580078) < 2e-5 and abs(petzval.compute_EFL() - 50.02990341) < 5e-5
    assert abs(cooke.compute_BFL() - 43.12704086303711) < 2e-5 and abs(cooke.compute_EFL() - 50.04488754) < 5e-5


def test_low_discrepancy_points_of_the_tracer_check(mi):
    """qmc.h:189-237: van der Corput / the second Sobol' dimension (0, 1/2, 1/4, 3/4, 1/8 ... and, with the direction numbers 1/2, 3/4, 5/8 in natural index order, 0, 1/2, 3/4, 1/4, 5/8 ...)."""
    ld = mi.lens_design
    assert [ld.radical_inverse_2(i) for i in range(6)] == [0.0, 0.5, 0.25, 0.75, 0.125, 0.625]
    assert [ld.sobol_2(i) for i in range(6)] == [0.0, 0.5, 0.75, 0.25, 0.625, 0.125]


def test_field_sources_are_laid_out_as_in_the_reference(mi):
    """FieldSource.createSourceArray (design_problem.py:142-181): angles 0 .. 5 degrees, aimed at the entrance pupil, far enough apart that
    neighbouring disks do not overlap; names as the reference formats them."""
    ld = mi.lens_design
    system = _doublet(ld, 1)
    tracer = lambda ray: (None, None, np.ones(1, dtype=bool), np.array([[0.1, 0.0, 0.0]]))
    fs = ld.FieldSource.createSourceArray(system, (16, 16), tracer, num_sources=3, max_field_angle=5.0, init_camera_z=0.01)
    assert [f.source_key for f in fs] == ["source_00_0-deg", "source_02_5-deg", "source_05_0-deg"]
    assert [f.sensor_key for f in fs] == ["sensor_00_0-deg", "sensor_02_5-deg", "sensor_05_0-deg"]
    dist = 1.1 * 8.0 / math.tan(0.5 * math.radians(2.5))
    for f, ang in zip(fs, (0.0, 2.5, 5.0)):
        assert np.allclose(f.target, [0, 0, 52.0]) and np.allclose(f.origin, np.array([0, 0, 52.0]) + dist * np.array([math.sin(math.radians(ang)), 0, math.cos(math.radians(ang))]))
        assert f.radius == 8.0 and f.near_clip == 0.005 and f.fov == 135.0
        z_cam = max(0.01 * system.rear_z, 1.1 * 0.005)
        assert np.allclose(f.camera_pos, [0.1, 0.0, z_cam])
        assert abs(f.get_pixel_size() - z_cam * math.tan(math.radians(67.5)) / 8) < 1e-12
    centres = np.array([f.origin for f in fs])
    assert np.linalg.norm(centres[1] - centres[0]) >= 2 * 8.0                      # the disks do not overlap
    # center_camera_on_pixel (:258-265): the reference's one-sided test -- only a spot left of the centre by more than a tenth moves the camera
    f = fs[1]; x0 = f.camera_pos[0]
    f.center_camera_on_pixel(7.5, 12.0); assert not f.pos_changed and f.camera_pos[0] == x0
    f.center_camera_on_pixel(7.5, 2.0); assert f.pos_changed and abs(f.camera_pos[0] - (x0 - 5.5 * f.get_pixel_size())) < 1e-12
    # upsample_sensor (:135-140): a 2x zoom halves tan(fov / 2)
    t0 = math.tan(math.radians(f.fov / 2)); f.upsample_sensor(); assert f.fov_changed and abs(math.tan(math.radians(f.fov / 2)) - t0 / 2) < 1e-12


@pytest.mark.parametrize("subdiv", [5])
def test_oracle_geometric_tracer_against_the_paraxial_matrices(mi, orc, subdiv):
    """design_problem.py:430-472 on the oracle's twin of prb_basic.trace: 100 near-axis rays through the three lens MESHES land on the film
    where the paraxial matrices say, and leave the rear surface with the paraxial slope.  The reference asks for atol 2e-5 on both and
    reports (cell 7 of the notebook, subdivision 7) 8.6e-6 for the slope -- passed -- and 3.8e-4 for the position -- failed, a mesh-normal
    effect: the slope error times the 47 mm to the film.  The same two magnitudes are asked for here."""
    ld = mi.lens_design
    system = _doublet(ld, subdiv)
    problem = ld.DesignProblem(system, resolution=(16, 16), spp=4, learning_rate=1e-4, iters=1, num_sources=3, max_field_angle=5.0, init_camera_z=0.01)
    geo = mi.load_dict(problem._build_preliminary_scene("prb_basic"))
    assert geo.integrator().name == "prb_basic" and geo.integrator().max_depth == 5
    o = orc.OracleScene(geo, mi.abi)

    def tracer(ray):
        rays = np.zeros((ray.o.shape[0], 8), dtype=np.float32); rays[:, :3] = ray.o; rays[:, 3:6] = ray.d; rays[:, 6] = np.float32(3.4028234663852886e38)
        out = o.trace(rays, system.size() + 1, geo.integrator().max_depth)
        z = np.abs(out[:, 8])
        valid = (out[:, 9] != 0) & bool(np.all(z <= 1e-8 + 1e-5 * z))
        return out[:, 0:3], out[:, 3:6], valid, out[:, 6:9]
    err_y, err_u = problem.test_geo_tracer(tracer)
    assert err_u < 2e-5, err_u
    assert err_y < 1e-3, err_y
    # the three central rays (one per field angle) reach the film: where get_sensor_dict puts the cameras (:192-247)
    fs = ld.FieldSource.createSourceArray(system, (16, 16), tracer, num_sources=3, max_field_angle=5.0, init_camera_z=0.01)
    xs = [f.camera_pos[0] for f in fs]
    assert abs(xs[0]) < 1e-3 and abs(fs[0].camera_pos[1]) < 1e-3
    # image height of a field angle theta for a lens of focal length 50 focused at infinity: -f tan(theta) (collimated source, inverted image)
    for f, ang in zip(fs, (0.0, 2.5, 5.0)):
        assert abs(f.camera_pos[0] - (-50.0 * math.tan(math.radians(ang)))) < 2e-2, (ang, f.camera_pos)
    # a ray that misses the lens dies on the black baffle / housing: invalid for the design problem
    out = o.trace(np.array([[30.0, 0.0, 60.0, 0.0, 0.0, -1.0, 3e38, 0.0]], dtype=np.float32), 4, 5)
    assert out[0, 9] == 0.0 or out[0, 10] < 4


def test_two_sensors_share_one_oracle_scene(mi, orc):
    """Scene::sensors(): rendering sensor i of a two-sensor scene is rendering a one-sensor scene that holds just that sensor."""
    from conftest import small_cbox
    T = mi.ScalarTransform4f
    d = small_cbox(mi, 16)
    second = {'type': 'perspective', 'fov': 50.0, 'near_clip': 0.01, 'far_clip': 50.0, 'to_world': T.look_at(origin=[0.5, 0.3, 3.0], target=[0, 0, 0], up=[0, 1, 0]),
              'sampler': {'type': 'independent', 'sample_count': 8}, 'film': {'type': 'hdrfilm', 'width': 24, 'height': 12, 'rfilter': {'type': 'tent'}, 'pixel_format': 'rgb'}}
    both = dict(d); both["second"] = second
    sc = mi.load_dict(both)
    assert [s.id for s in sc.sensors()] == ["sensor", "second"] and sc.sensor is sc.sensors()[0]
    assert sc.sensor_index(1) == 1 and sc.sensor_index(sc.sensors()[1]) == 1
    with pytest.raises(mi.PluginError):
        sc.sensor_index(2)
    only = dict(d); del only["sensor"]; only["second"] = second
    integ = mi.PathIntegrator({"max_depth": 3})
    o2 = orc.OracleScene(sc, mi.abi); o1 = orc.OracleScene(mi.load_dict(only), mi.abi)
    img2, _ = o2.render(o2.params(integ, 1, 8, sensor=1)); img1, _ = o1.render(o1.params(integ, 1, 8))
    assert img2.shape == (12, 24, 3) and np.allclose(img1, img2, rtol=1e-5, atol=1e-8)      # (same samples; the oracle's OpenMP threads add their films in any order)
    first, _ = o2.render(o2.params(integ, 1, 8, sensor=0))
    assert first.shape == (16, 16, 3)
    # a sensor parameter update (perspective.cpp:155-171) on the oracle twin: x_fov
    sc.sensors()[1].x_fov = 35.0; o2.set_sensor(1)
    zoomed, _ = o2.render(o2.params(integ, 1, 8, sensor=1))
    assert zoomed.shape == (12, 24, 3) and not np.allclose(zoomed, img2, rtol=1e-3)


def test_tracer_check_reproduces_the_notebooks_numbers_at_its_mesh_level(mi, orc, kats):
    """Cell 7 of the notebook prints what the reference's own check measured on the subdivision-7 doublet: slope error 8.578e-6, position
    error 3.80e-4 (both against atol 2e-5: "u is close: True", "y is close: False").  The oracle twin on this build's meshes: the slope
    error within a few per cent of that number, the position error of the same size (it is the slope error times the 47 mm to the film)."""
    ld = mi.lens_design; ref = kats["lens_design_notebook"]["tracer_check_max_abs_err"]
    system = _doublet(ld, 7)
    problem = ld.DesignProblem(system, resolution=(16, 16), spp=4, learning_rate=1e-4, iters=1, num_sources=3, max_field_angle=5.0, init_camera_z=0.01)
    geo = mi.load_dict(problem._build_preliminary_scene("prb_basic"))
    o = orc.OracleScene(geo, mi.abi)

    def tracer(ray):
        rays = np.zeros((ray.o.shape[0], 8), dtype=np.float32); rays[:, :3] = ray.o; rays[:, 3:6] = ray.d; rays[:, 6] = np.float32(3.4028234663852886e38)
        out = o.trace(rays, system.size() + 1, geo.integrator().max_depth)
        return out[:, 0:3], out[:, 3:6], out[:, 9] != 0, out[:, 6:9]
    err_y, err_u = problem.test_geo_tracer(tracer)
    assert abs(err_u - ref["u"]) < 0.05 * ref["u"], (err_u, ref["u"])
    assert abs(err_y - ref["y"]) < 0.3 * ref["y"], (err_y, ref["y"])
    assert (err_u < ref["atol_asked"]) == ref["u_is_close"] and (err_y < ref["atol_asked"]) == ref["y_is_close"]


def test_focal_length_constraint_is_differentiable(mi):
    """ConstrainedEFLProblem (design_problem.py:656-703) re-derives the rear surface's c and z0 from the OTHER surfaces in every iteration
    (LensSystem.fixed_EFL_update -> get_rear_surface_params, lens.py:970-997, :1023-1042), and in the reference Dr.Jit's AD follows that
    constraint.  Here: with optimiser-like scalars (1-element tensors that require grad) get_rear_surface_params returns tensors whose
    gradients equal central finite differences of the plain-float evaluation, and the constrained system keeps its focal length."""
    import torch
    ld = mi.lens_design
    system = _doublet(ld, 3, constrained=False)
    base = {(e, k): float(system.elements[e].surface.params[k]) for e in (1, 2) for k in ('c', 'z0')}
    plain = system.get_rear_surface_params(50.0)
    assert isinstance(plain['c'], float) and isinstance(plain['z0'], float)
    leaves = {}
    for (e, k), v in base.items():
        leaves[(e, k)] = torch.tensor([v], dtype=torch.float64, requires_grad=True); system.elements[e].surface.params[k] = leaves[(e, k)]
    out = system.get_rear_surface_params(50.0)
    assert torch.is_tensor(out['c']) and out['c'].shape == (1,) and abs(float(out['c'].detach()) - plain['c']) < 1e-12 and abs(float(out['z0'].detach()) - plain['z0']) < 1e-10
    for name in ('c', 'z0'):
        for t in leaves.values():
            t.grad = None
        out = system.get_rear_surface_params(50.0); out[name].sum().backward()
        for (e, k), t in leaves.items():
            h = 1e-6 * max(1.0, abs(base[(e, k)]))
            vals = []
            for sgn in (+1, -1):
                for (e2, k2), v2 in base.items():
                    system.elements[e2].surface.params[k2] = v2 + (sgn * h if (e2, k2) == (e, k) else 0.0)
                vals.append(system.get_rear_surface_params(50.0)[name])
            fd = (vals[0] - vals[1]) / (2 * h)
            for (e2, k2), t2 in leaves.items():
                system.elements[e2].surface.params[k2] = t2
            assert abs(float(t.grad) - fd) <= 1e-5 * max(abs(fd), 1e-6) + 1e-9, (name, e, k, float(t.grad), fd)
    # the constrained system has the focal length it was asked for, and its paraxial focus on the film plane (z = 0)
    for (e, k), v in base.items():
        system.elements[e].surface.params[k] = v
    s1 = system.get_rear_surface_params(50.0)
    system.elements[0].surface.params['c'] = s1['c']; system.elements[0].surface.params['z0'] = s1['z0']; system.compute_paraxial_quantities()
    assert abs(system.compute_EFL() - 50.0) < 1e-9 and abs(system.compute_BFL() - s1['z0']) < 1e-9


def _cooke_triplet(ld, subdiv):
    """notebooks/design-problem-multi-spot-efl.ipynb cell 6: the Cooke triplet "from OSLO", six surfaces, three glasses with air between them."""
    surfs = [{'radial_extent': 6.5, 'c': -17.285, 'z0': 42.95}, {'radial_extent': 6.5, 'c': 141.25, 'z0': 44.95}, {'radial_extent': 5.0, 'c': 19.3, 'z0': 50.95},
             {'radial_extent': 5.0, 'c': -20.25, 'z0': 51.95}, {'radial_extent': 6.5, 'c': -158.65, 'z0': 57.95}, {'radial_extent': 6.5, 'c': 21.25, 'z0': 59.95}]
    surfaces = [ld.EvenAsphericSurface(K=0, radial_extent=s['radial_extent'], c=1.0 / s['c'], z0=s['z0']) for s in surfs]
    mats = [ld.LensMaterial("sk16", 1.62041, 60.32), ld.LensMaterial(), ld.LensMaterial("f4", 1.616592, 36.63), ld.LensMaterial(), ld.LensMaterial("sk16", 1.62041, 60.32)]
    system = ld.LensSystem(surfaces, mats, aperture_index=None, subdiv_levels=subdiv)
    system.disable_all_materials()
    s1 = system.get_rear_surface_params(50.0)
    system.elements[0].surface.params['c'] = s1['c']; system.elements[0].surface.params['z0'] = s1['z0']
    system.compute_paraxial_quantities()
    return system


def test_cooke_triplet_matches_the_second_notebooks_stored_outputs(mi):
    """design-problem-multi-spot-efl.ipynb, stored output of cell 6 (float32 Dr.Jit arithmetic there, double here):
        Focal length target:  [49.99998092651367]  vs.  50.0
        BFL focus target:  [43.05345153808594]  vs.  [43.05345916748047]
        Exit pupil position:  -25.17422180175781
        Entrance pupil position:  -7.629394502828291e-07"""
    ld = mi.lens_design
    system = _cooke_triplet(ld, 3)
    assert abs(system.compute_EFL() - 49.99998092651367) < 5e-5
    assert abs(system.compute_BFL() - 43.05345153808594) < 2e-5 and abs(ld._f(system.elements[0].surface.params['z0']) - 43.05345916748047) < 2e-5
    assert abs((system.rear_z - system.exit_pupil_position) - (-25.17422180175781)) < 1e-4
    assert abs(system.front_z - system.entrance_pupil_position) < 1e-5
    assert system.size() == 6 and [m.name for m in system.materials] == ['air', 'sk16', 'air', 'f4', 'air', 'sk16']     # (film-side air first; the world side wraps around to it)


def test_tracer_check_on_the_cooke_triplet_at_the_notebooks_mesh_level(mi, orc):
    """The same notebook's cell 9 prints what the reference's tracer check measured on the six subdivision-7 surfaces:
        [0.0016724765300750732]   y is close:  False          [3.399315755814314e-05]   u is close:  False
    (ConstrainedEFLProblem(num_sources=5, max_field_angle=7.5, init_camera_z=0.001); both ABOVE its atol 2e-5 on this lens.)  The oracle
    twin of prb_basic.trace on this build's meshes gives the same verdict and the same magnitudes (see the comment at the assertions)."""
    ld = mi.lens_design
    system = _cooke_triplet(ld, 7)
    problem = ld.DesignProblem(system, resolution=(16, 16), spp=4, learning_rate=1e-4, iters=1, num_sources=5, max_field_angle=7.5, init_camera_z=0.001)
    geo = mi.load_dict(problem._build_preliminary_scene("prb_basic"))
    o = orc.OracleScene(geo, mi.abi)

    def tracer(ray):
        rays = np.zeros((ray.o.shape[0], 8), dtype=np.float32); rays[:, :3] = ray.o; rays[:, 3:6] = ray.d; rays[:, 6] = np.float32(3.4028234663852886e38)
        out = o.trace(rays, system.size() + 1, geo.integrator().max_depth)
        return out[:, 0:3], out[:, 3:6], out[:, 9] != 0, out[:, 6:9]
    err_y, err_u = problem.test_geo_tracer(tracer)
    ref_y, ref_u = 0.0016724765300750732, 3.399315755814314e-05
    # On the doublet this build reproduces the reference's slope error to 0.2 % (test above); on the triplet it measures 6.0e-5 / 2.8e-3 against the
    # notebook's 3.4e-5 / 1.7e-3, and here the error GROWS with the subdivision level (2.7e-5, 3.0e-5, 6.0e-5 at levels 5, 6, 7): at level 7
    # it is float32 rounding of the 0.05 mm edges the vertex normals next to the pole are formed from (ulp(50) / 0.05 = 8e-5), a quantity that
    # depends on the mesh's vertex order, not a smooth function of the design.  So: the same verdict as the notebook printed, and its magnitude.
    assert 0.5 * ref_u < err_u < 2.5 * ref_u, (err_u, ref_u)
    assert 0.5 * ref_y < err_y < 2.5 * ref_y, (err_y, ref_y)
    assert err_u > 2e-5 and err_y > 2e-5                                      # "u is close: False", "y is close: False"


def _oracle_tracer(mi, orc, problem, system):
    geo = mi.load_dict(problem._build_preliminary_scene("prb_basic"))
    o = orc.OracleScene(geo, mi.abi)

    def tracer(ray):
        rays = np.zeros((ray.o.shape[0], 8), dtype=np.float32); rays[:, :3] = ray.o; rays[:, 3:6] = ray.d; rays[:, 6] = np.float32(3.4028234663852886e38)
        out = o.trace(rays, system.size() + 1, geo.integrator().max_depth)
        return out[:, 0:3], out[:, 3:6], out[:, 9] != 0, out[:, 6:9]
    return tracer


def test_tracer_check_on_the_singlet_of_the_first_design_notebook(mi, orc):
    """notebooks/design-problem-multi-spot.ipynb cells 4 and 6: a biconvex singlet (c = -+0.3 / mm, radial extent 0.8, nbk7) in
    DesignProblem(init_camera_z=0.03, num_sources=3, max_field_angle=10); the stored output of cell 6 is the tracer check at subdivision 7:
        [9.58610326051712e-06]   y is close:  True          [1.3655517250299454e-06]   u is close:  True
    NOT used as a pin of the magnitudes: that notebook was executed with an OLDER version of the scripts than the tree holds -- its stored
    final parameters (cell 10: 'c': [-0.2629818...]) carry the curvature in 1 / mm, while lens.py stores c times the radial extent -- and this
    build measures 2.6e-5 / 8.5e-6 on the meshes of the CURRENT geometry.py (the doublet and the Nikon design, whose notebooks are current,
    agree with the reference to 0.2 % and 16 %).  Asserted: the slope passes the reference's atol as it did there, and the position error is
    the slope error times the ~3 mm to the film."""
    ld = mi.lens_design
    surfaces = [ld.ConicSurface(radial_extent=0.8, c=-0.3, K=0.0, z0=2.5), ld.ConicSurface(radial_extent=0.8, c=0.3, K=0.0, z0=2.8)]
    system = ld.LensSystem(surfaces, [ld.LensMaterial("nbk7", 1.5047, 64.17)], subdiv_levels=7)
    system.disable_all_materials(); system.disable_surface_vars("z0"); system.disable_surface_vars("K")
    problem = ld.DesignProblem(system, resolution=(16, 16), spp=4, learning_rate=1e-3, iters=1, init_camera_z=0.03, num_sources=3, max_field_angle=10)
    err_y, err_u = problem.test_geo_tracer(_oracle_tracer(mi, orc, problem, system))
    assert err_u < 2e-5, err_u                                                # "u is close: True"
    assert err_y < 4.0 * err_u * 3.0, (err_y, err_u)


def _nikon(ld, subdiv):
    """notebooks/report_nikon_exp.ipynb cells 12 and 14: a 22-surface wide-angle design (its two aspheres zeroed, "sphere-only"), aperture stop at
    table row 12, air gaps as LensMaterial(), surfaces listed from the world towards the film and reversed by the parser."""
    curvature_radii = [5.2674, 0.9607, 1.442, 10.2799, 1.2154, -1.0987, 2.9183, -1.6693, 1.6429, -1.4116, -2.5725, np.inf, -0.973, -24.0801, 2.3756, -1.3055, -7.3169, -2.2, -1.5449,
                       -7.2572, np.inf, np.inf]
    thicknesses = [0.102, 0.309, 0.246, 0.083, 0.411, 0.088, 0.258, 0.009, 0.379, 0.069, 0.118, 0.604, 0.051, 0.009, 0.282, 0.239, 0.122, 0.154, 0.083, 0.75, 0.074, 0.0425]
    elem_radii = [0.847, 0.696, 0.661, 0.625, 0.546, 0.524, 0.586, 0.601, 0.624, 0.613, 0.607, 0.555, 0.476, 0.49, 0.543, 0.569, 0.604, 0.627, 0.662, 0.712, 1.2, 1.2]
    M = ld.LensMaterial
    mats = [M("glass_A", 1.5168, 64.12), M(), M("glass_B", 1.9027, 35.72), M("glass_C", 1.5955, 39.21), M(), M("glass_D", 1.699, 30.05), M("glass_E", 1.9108, 35.25), M(),
            M("glass_F", 1.5928, 68.62), M("glass_G", 1.7205, 34.7), M(), M(), M("glass_H", 1.5927, 35.31), M(), M("glass_F", 1.5928, 68.62), M(), M("glass_I", 1.6935, 53.2), M(),
            M("glass_J", 1.4875, 70.44), M(), M("glass_A", 1.5168, 64.12), M()]
    aperture_tbl_index = 12; aspheric = {17: [0.0] * 8, 18: [0.0] * 8}
    n = len(elem_radii); surfaces = []; z0 = 0.0
    for i in reversed(range(n)):
        tbl = i + 1; c = 1.0 / curvature_radii[i]; z0 += thicknesses[i]; r = elem_radii[i]
        if tbl in aspheric:
            K, a4, a6, a8, a10, a12, a14, a16 = aspheric[tbl]
            surfaces.append(ld.EvenAsphericSurface(radial_extent=r, c=c, K=K, z0=z0, a4=a4, a6=a6, a8=a8, a10=a10, a12=a12, a14=a14, a16=a16))
        elif tbl == aperture_tbl_index:
            surfaces.append(ld.ConicSurface(radial_extent=r, c=0.0, K=0.0, z0=z0))
        else:
            surfaces.append(ld.ConicSurface(radial_extent=r, c=c, K=0.0, z0=z0))
    aperture_index = n - aperture_tbl_index
    system = ld.LensSystem(surfaces, mats[:-1][::-1], aperture_index=aperture_index, subdiv_levels=subdiv)
    system.disable_all_materials(); system.disable_surface_at_index(aperture_index)
    for k in ("K", "z0", "c"):
        system.disable_surface_vars(k)
    return system, aperture_index


def test_nikon_design_matches_the_report_notebooks_stored_outputs(mi, orc):
    """report_nikon_exp.ipynb, stored outputs of cell 14 (paraxial quantities of the 22-surface design with its aperture stop) and cell 16 (the
    tracer check of DesignProblem(num_sources=6, max_field_angle=32.45, init_camera_z=0.99) at subdivision 7):
        aperture_index=10     Focal length target:  [1.5734491348266602]  vs.  1.57     BFL focus target:  [0.045883748680353165]  vs.  0.0425
        Exit pupil position:  2.3486416339874268     Entrance pupil position:  3.3314104080200195
        [9.683542884886265e-05]   y is close:  False          [4.520252696238458e-05]   u is close:  False"""
    ld = mi.lens_design
    system, aperture_index = _nikon(ld, 7)
    assert aperture_index == 10
    assert abs(system.compute_EFL() - 1.5734491348266602) < 2e-6 and abs(system.compute_BFL() - 0.045883748680353165) < 2e-6
    assert abs(ld._f(system.elements[0].surface.params['z0']) - 0.0425) < 1e-9
    assert abs(system.exit_pupil_position - 2.3486416339874268) < 2e-6 and abs(system.entrance_pupil_position - 3.3314104080200195) < 2e-6
    problem = ld.DesignProblem(system, resolution=(16, 16), spp=4, learning_rate=1e-4, iters=1, num_sources=6, max_field_angle=0.5 * 64.9, init_camera_z=0.99)
    err_y, err_u = problem.test_geo_tracer(_oracle_tracer(mi, orc, problem, system))
    ref_y, ref_u = 9.683542884886265e-05, 4.520252696238458e-05
    assert err_u > 2e-5 and err_y > 2e-5                                      # "u is close: False", "y is close: False"
    assert abs(err_u - ref_u) < 0.3 * ref_u, (err_u, ref_u)                    # measured: 5.2e-5
    assert 0.5 * ref_y < err_y < 2.0 * ref_y, (err_y, ref_y)                   # measured: 1.5e-4


def test_a_surface_with_unchanged_scalars_is_not_pushed_again(mi):
    """LensElement.update (lens.py:509-568) writes every surface's vertices each iteration; here a surface that keeps plain floats equal to
    the last push is left alone (20 of the 22 surfaces of report_nikon_exp.ipynb), and one whose float is edited by hand is pushed again."""
    import torch
    ld = mi.lens_design
    el = ld.LensElement(0, ld.EvenAsphericSurface(radial_extent=4.0, c=1 / 30, K=0.0, z0=10.0), ld.LensMaterial("nbk7", 1.5168, 64.17), ld.LensMaterial(), N=2)
    el.initialize_geometry(); el.lens_key = "lens0"; el.baffle_key = "baffle0"

    class Recorder(dict):
        writes = 0
        def __setitem__(self, k, v):
            if k.endswith("vertex_positions"):
                Recorder.writes += 1
            dict.__setitem__(self, k, v)
    params = Recorder()
    dict.__setitem__(params, "lens0.vertex_positions", torch.tensor(el.V_lens, dtype=torch.float32).reshape(-1))
    dict.__setitem__(params, "baffle0.vertex_positions", torch.tensor(el.V_ap, dtype=torch.float32).reshape(-1))
    el.save_init_state(params); el.remove_all_optvars()
    opt = {}; el.add_to_optimizer(opt, "cpu"); assert opt == {}
    el.update(params, opt); assert Recorder.writes == 2                       # first push: the lens mesh and its baffle
    first = params["lens0.vertex_positions"].clone()
    el.update(params, opt); el.update(params, opt); assert Recorder.writes == 2
    el.surface.params['z0'] = 10.5
    el.update(params, opt); assert Recorder.writes == 4
    assert torch.allclose(params["lens0.vertex_positions"].reshape(-1, 3)[:, 2], first.reshape(-1, 3)[:, 2] + 0.5, atol=1e-6)
    el.surface.params['c'] = torch.tensor([1 / 30])                           # a tensor may change in place: always pushed
    el.update(params, opt); el.update(params, opt); assert Recorder.writes == 8
    el.surface.params['c'] = 1 / 30
    el.update(params, opt); writes = Recorder.writes; el.update(params, opt); assert Recorder.writes == writes
    other = Recorder()                                                        # the parameters of ANOTHER scene over the same element: pushed afresh
    dict.__setitem__(other, "lens0.vertex_positions", first.clone()); dict.__setitem__(other, "baffle0.vertex_positions", params["baffle0.vertex_positions"].clone())
    el.update(other, opt); assert Recorder.writes == writes + 2
