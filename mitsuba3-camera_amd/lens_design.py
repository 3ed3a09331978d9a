"""Host-side mirror of the fork's lens-design scripts -- the caller of BASELINE config 5 (notebooks/doublet_full_optim.ipynb):

  notebooks/scripts/lens_design/geometry.py   create_surface_geometry :296-390, get_spherical_cap :122-158, get_hemisphere_ico :44-70
  notebooks/scripts/lens_design/lens.py       ConicSurface :55-97, EvenAsphericSurface :166-249, LensMaterial :252-339,
                                              LensElement :343-581 (add_to_scene :418-470, add_to_optimizer :473-498, update :509-568),
                                              LensSystem :752-808
  notebooks/scripts/lens_design/design_problem.py   FieldSource :12-279 (directionalarea disk source + perspective sensor per field angle,
                                              createSourceArray :142-181, get_sensor_dict :192-247, center_camera_on_pixel :258-265,
                                              upsample_sensor :135-140, update :86-132),
                                              DesignProblem :283-584 (_build_scene :403-428 with the `prb_basic` geometric tracer,
                                              __test_geo_tracer :430-472, prepare :481-492, optimize :495-584)
  notebooks/scripts/lens_design/lens.py       LensSystem.compute_paraxial_quantities :810-880, compute_pupils :956-968, compute_BFL / FFL / EFL
                                              :932-954, get_rear_surface_params :970-997, abbe_to_cauchy :1251-1255, ApertureElement :585-744
  notebooks/scripts/lens_design/losses.py     rms_loss :8-29, rms_loss_and_center :31-41

What the notebook does per iteration, and what runs where here:
  lens scalars (c, K, z0, a4..a16 per surface; ior / V_d per glass)           torch leaf tensors held by mi.Adam
    -> surface.compute_z(x0, y0) on the fixed (x0, y0) of every mesh vertex     torch (autograd keeps d(vertex)/d(scalar))
    -> params['lensK.vertex_positions'] = ...; params.update()                  C ABI: MI355_PARAM_VERTEX_POSITIONS -> device refit + normals
    -> mi.render(scene, params, ...) with the `ptracer` integrator              HIP: k_ptracer (+ deferred splat)
    -> RMS-spot loss on the image; loss.backward()                              torch -> _RenderOp.backward -> HIP: k_ptracer_geo_bwd
                                                                                (reverse-mode geometry / IOR adjoint) -> torch chain rule
    -> optimizer.step()
Dr.Jit arrays become torch tensors; units are millimetres, the film is the plane z = 0 and +z faces the world, as in the fork.
The icosphere / boundary-loop helpers of `gpytoolbox` (absent here) are restated below; vertex numbering differs from that
library's, nothing depends on it.  `num_sources` field sources per scene, each with its own sensor; a render names its sensor
(mi.render(..., sensor=i), design_problem.py:519-526) and the device replica is shared by all of them.
"""
import math
import numpy as np

from . import _abi as abi
from .transforms import ScalarTransform4f


def _torch():
    import torch
    return torch


# ------------------------------------------------------------------------------------------------
# geometry.py
# ------------------------------------------------------------------------------------------------
def icosphere(n):
    """Unit icosahedron subdivided n times (each triangle -> 4), vertices pushed onto the sphere: 10 * 4^n + 2 vertices, 20 * 4^n faces
    (what gpytoolbox.icosphere returns, geometry.py:55)."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    V = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    F = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(n):
        e = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), axis=1)
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = V[ue[:, 0]] + V[ue[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        nf = F.shape[0]; base = V.shape[0]
        m01, m12, m20 = base + inv[:nf], base + inv[nf:2 * nf], base + inv[2 * nf:]
        V = np.vstack([V, mid])
        F = np.concatenate([np.stack([F[:, 0], m01, m20], 1), np.stack([F[:, 1], m12, m01], 1), np.stack([F[:, 2], m20, m12], 1),
                            np.stack([m01, m12, m20], 1)])
    return V, F


def remove_unreferenced(V, F):
    used = np.zeros(V.shape[0], dtype=bool); used[F.ravel()] = True
    remap = np.cumsum(used) - 1
    return V[used], remap[F]


def get_hemisphere_ico(n):
    """geometry.py:44-70: the +z half of the icosphere, open at z = 0."""
    V, F = icosphere(n)
    avg_edge_length = math.sqrt((4.0 * math.pi) / F.shape[0])
    V_mask = V[:, 2] < -avg_edge_length / 2.0
    valid = ~V_mask[F].any(axis=1)
    return remove_unreferenced(V, F[valid])


def boundary_loop(F):
    """Ordered loop of the boundary vertices of a triangle mesh with one open boundary (gpytoolbox.boundary_loops(F)[0])."""
    e = np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]])
    key = np.sort(e, axis=1)
    _, idx, cnt = np.unique(key, axis=0, return_index=True, return_counts=True)
    be = e[idx[cnt == 1]]                              # boundary edges, oriented as in their (only) face
    nxt = {int(a): int(b) for a, b in be}
    start = int(be[:, 0].min()); loop = [start]; cur = nxt[start]
    while cur != start:
        loop.append(cur); cur = nxt[cur]
    return np.array(loop, dtype=np.int64)


def get_spherical_cap(n, elem_radius, abs_curvature):
    """geometry.py:122-158: the hemisphere rescaled in angle to the cap a sphere of the given curvature cuts at the element radius."""
    V, F = get_hemisphere_ico(n)
    target_angle = math.asin(elem_radius * max(abs_curvature, 1e-3))
    angle_rescale = target_angle / (0.5 * math.pi)
    V_ = V.copy()
    vlens = np.linalg.norm(V_, axis=1)
    angles = np.arccos(np.clip(V_[:, 2] / vlens, -1.0, 1.0)) * angle_rescale
    V_[:, 2] = vlens * np.cos(angles)
    with np.errstate(divide="ignore", invalid="ignore"):
        V_[:, :2] *= (vlens * np.sin(angles) / np.linalg.norm(V[:, :2], axis=1))[:, None]
    V_[np.isnan(V_)] = 0.0
    V_[:, 2] -= math.cos(target_angle)
    V_ *= elem_radius / math.sin(target_angle)
    return V_, F


def _draw_quad(v0, v1, v2, v3, flip=False):
    f1 = [v0, v1, v2]; f2 = [v2, v3, v0]
    if flip:
        f1[0], f1[1] = f1[1], f1[0]; f2[0], f2[1] = f2[1], f2[0]
    return [f1, f2]


def create_surface_geometry(N, r_element, compute_z, c=None, baffle_radius=5.0, flip_normals=False):
    """geometry.py:296-390: the mesh of one lens surface (spherical-cap template, z from the sag function) and its baffle (a two-sided
    frame from the surface's open edge out to a square of half width `baffle_radius`).  Returns V_lens, F_lens, V_ap, F_ap, ov_mask."""
    if c is None:
        h = 1e-2 * r_element
        c = abs(float(compute_z(np.array([h]), np.array([h]))[0]) - float(compute_z(np.array([0.0]), np.array([0.0]))[0])) / (h ** 2)
    V_lens, F_lens = get_spherical_cap(N, r_element, abs(c))
    F_lens = F_lens.copy()
    if flip_normals:
        F_lens[:, [0, 1]] = F_lens[:, [1, 0]]
    V_lens[:, 2] = compute_z(V_lens[:, 0], V_lens[:, 1])
    open_vertices = boundary_loop(F_lens)
    Nvo = len(open_vertices)
    ov_mask = np.zeros(V_lens.shape[0], dtype=bool); ov_mask[open_vertices] = True

    def clip_to_box(V, xmax):
        V_ = V.copy()
        V_[:, :2] *= xmax / np.linalg.norm(V[:, :2], axis=1, ord=np.inf)[:, None]
        return V_
    V_ap_el = V_lens[open_vertices]
    V_ap_edge = clip_to_box(V_ap_el, baffle_radius)
    F_ap = []
    for idx in range(Nvo):
        nidx = (idx + 1) if (idx + 1 < Nvo) else 0
        F_ap += _draw_quad(idx, nidx, Nvo + nidx, Nvo + idx, True)
        F_ap += _draw_quad(Nvo + idx, Nvo + nidx, nidx, idx, True)
    V_ap = np.vstack((V_ap_el, V_ap_edge))
    return V_lens, F_lens, V_ap, np.array(F_ap, dtype=np.int64), ov_mask


# ------------------------------------------------------------------------------------------------
# lens.py
# ------------------------------------------------------------------------------------------------
def _t(v, like=None):
    torch = _torch()
    if torch.is_tensor(v):
        return v
    return torch.tensor(float(v), dtype=torch.float32, device=like.device if like is not None else None)


class Surface:
    def __init__(self, radial_extent, params):
        self.params = params; self.radial_extent = float(radial_extent)

    def get_curvature(self):
        """Curvature in 1 / mm (the stored `c` is dimensionless), lens.py:40-44."""
        c = self.params['c']
        return float(c) / self.radial_extent if not _torch().is_tensor(c) else float(c.detach().reshape(-1)[0]) / self.radial_extent

    def get_z0(self):
        return self.params['z0']

    def compute_z_np(self, x_, y_):
        torch = _torch()
        with torch.no_grad():
            z = self.compute_z(torch.as_tensor(np.asarray(x_, dtype=np.float64), dtype=torch.float32), torch.as_tensor(np.asarray(y_, dtype=np.float64), dtype=torch.float32))
        return z.cpu().numpy().astype(np.float64)


class ConicSurface(Surface):
    """lens.py:55-97 (dimensionless form)."""

    def __init__(self, radial_extent, c, K, z0):
        super().__init__(radial_extent, {'c': c * radial_extent, 'K': K, 'z0': z0})

    def compute_z(self, x, y):
        torch = _torch()
        P = {k: _t(v, x).to(x.device) for k, v in self.params.items()}
        r2 = (x * x + y * y) * (1.0 / (self.radial_extent * self.radial_extent))
        safe = torch.clamp(1 - (1 + P['K']) * P['c'] * P['c'] * r2, min=0.0)
        return P['z0'] - self.radial_extent * r2 * P['c'] / (1 + torch.sqrt(safe))


class EvenAsphericSurface(Surface):
    """lens.py:166-249: conic + even polynomial a4 r^4 ... a16 r^16, all coefficients non-dimensionalised by the radial extent."""
    COEFFS = ('a4', 'a6', 'a8', 'a10', 'a12', 'a14', 'a16')

    def __init__(self, radial_extent, c, K, z0, a4=0.0, a6=0.0, a8=0.0, a10=0.0, a12=0.0, a14=0.0, a16=0.0):
        r = radial_extent
        super().__init__(radial_extent, {'c': c * r, 'K': K, 'z0': z0, 'a4': a4 * r ** 3, 'a6': a6 * r ** 5, 'a8': a8 * r ** 7, 'a10': a10 * r ** 9,
                                         'a12': a12 * r ** 11, 'a14': a14 * r ** 13, 'a16': a16 * r ** 15})

    def compute_z(self, x, y):
        torch = _torch()
        P = {k: _t(v, x).to(x.device) for k, v in self.params.items()}
        r2_ = (x * x + y * y) * (1.0 / (self.radial_extent * self.radial_extent))
        z = P['a16'] + 0 * r2_
        for k in ('a14', 'a12', 'a10', 'a8', 'a6', 'a4'):
            z = z * r2_ + P[k]
        safe = torch.clamp(1 - (1 + P['K']) * P['c'] * P['c'] * r2_, min=0.0)
        z = z * r2_ + P['c'] / (1 + torch.sqrt(safe))
        return z * (-self.radial_extent * r2_) + P['z0']


class LensMaterial:
    """lens.py:252-339: a glass by its index at 589.3 nm and its Abbe number; the default is air."""

    def __init__(self, name="air", refractive_index=1.000277, abbe_number=0.0):
        self.name = name.lower()
        self.params = {'ior': refractive_index, 'V_d': abbe_number}
        self.param_keys_to_opt_keys = None
        self.active_optvars = [] if self.name == "air" else list(self.params.keys())

    def remove_all_optvars(self):
        self.active_optvars = []

    def remove_optvar(self, param_name):
        if param_name in self.active_optvars:
            self.active_optvars.remove(param_name)

    def add_to_optimizer(self, optimizer, device):
        torch = _torch()
        if self.name == "air":
            return
        self.param_keys_to_opt_keys = self.param_keys_to_opt_keys or {}
        for name, value in self.params.items():
            if name not in self.active_optvars:
                continue
            key = 'mat_%s_%s' % (self.name, name)
            if key in optimizer:
                raise KeyError("Variable %s already exists in optimizer!" % key)
            optimizer[key] = torch.tensor([float(value)], dtype=torch.float32, device=device)
            self.param_keys_to_opt_keys[name] = key

    def update(self, optimizer):
        if self.name == "air" or self.param_keys_to_opt_keys is None:
            return
        for var_key, opt_key in self.param_keys_to_opt_keys.items():
            self.params[var_key] = optimizer[opt_key]


def _pushed_states(params):
    """mesh key -> the surface scalars its vertices were last computed from, stored ON the scene's parameter object: the record lives and dies
    with the scene it describes (a second DesignProblem over the same LensSystem starts from its own freshly built meshes)."""
    try:
        return params.__dict__.setdefault("_pushed_surface_states", {})
    except AttributeError:                                                    # a plain dict stands in for SceneParameters in some host tests
        return {}


class LensElement:
    """lens.py:343-581: ONE refracting surface as a mesh (`dispersive` interface between two media) plus its black baffle mesh."""

    def __init__(self, element_id, surface, int_material, ext_material, N=5, is_world_facing=True):
        self.subdiv_level = N; self.id = element_id
        self.int_material = int_material; self.ext_material = ext_material
        self.surface = surface
        self.param_keys_to_opt_keys = None
        self.is_world_facing = is_world_facing
        self.active_optvars = list(self.surface.params.keys())
        self._pushed_materials = {}

    def remove_optvar(self, param_name):
        if param_name in self.active_optvars:
            self.active_optvars.remove(param_name)

    def remove_optvars(self, names):
        self.active_optvars = [v for v in self.active_optvars if v not in names]

    def remove_all_optvars(self):
        self.active_optvars = []

    def initialize_geometry(self):
        """lens.py:373-401 without the PLY round trip: the buffers go to the scene as a `mesh` with computed vertex normals, which is what
        the `ply` loader makes of a file without normals (ply.cpp:444-446)."""
        V_lens, F_lens, V_ap, F_ap, ovs_mask = create_surface_geometry(
            N=self.subdiv_level, r_element=self.surface.radial_extent, compute_z=self.surface.compute_z_np,
            c=self.surface.get_curvature(), flip_normals=not self.is_world_facing, baffle_radius=1.1 * self.surface.radial_extent)
        self.V_lens, self.F_lens, self.V_ap, self.F_ap, self.ovs_mask = V_lens, F_lens, V_ap, F_ap, ovs_mask

    def add_to_scene(self, scene_dict):
        """lens.py:418-470"""
        self.lens_key = "lens%d" % self.id; self.baffle_key = "baffle%d" % self.id
        scalar = lambda v: float(v.detach().reshape(-1)[0]) if _torch().is_tensor(v) else float(v)
        bsdf_dict = {'type': 'dispersive',
                     'ext_ior': scalar(self.ext_material.params['ior']), 'ext_V_d': scalar(self.ext_material.params['V_d']),
                     'int_ior': scalar(self.int_material.params['ior']), 'int_V_d': scalar(self.int_material.params['V_d']),
                     'specular_reflectance': 0.0}
        if 'black-bsdf' not in scene_dict:
            scene_dict['black-bsdf'] = {'type': 'diffuse', 'reflectance': 0.0}
        for key in (self.lens_key, self.baffle_key):
            if key in scene_dict:
                raise KeyError("`%s` already exists in scene!" % key)
        scene_dict[self.lens_key] = {'type': 'mesh', 'vertex_positions': self.V_lens.astype(np.float32), 'faces': self.F_lens.astype(np.uint32),
                                     'compute_vertex_normals': True, 'bsdf': bsdf_dict}
        scene_dict[self.baffle_key] = {'type': 'mesh', 'vertex_positions': self.V_ap.astype(np.float32), 'faces': self.F_ap.astype(np.uint32),
                                       'compute_vertex_normals': True, 'bsdf': {'type': 'ref', 'id': 'black-bsdf'}}

    def add_to_optimizer(self, optimizer, device):
        """lens.py:473-498"""
        torch = _torch()
        self.param_keys_to_opt_keys = self.param_keys_to_opt_keys or {}
        for var_name, value in self.surface.params.items():
            if var_name not in self.active_optvars:
                continue
            key = 'lens%d_%s' % (self.id, var_name)
            if key in optimizer:
                raise KeyError("Variable %s already exists in optimizer!" % key)
            optimizer[key] = torch.tensor([float(value)], dtype=torch.float32, device=device)
            self.param_keys_to_opt_keys[var_name] = key

    def save_init_state(self, params):
        """lens.py:501-506"""
        self.initial_lens_vertices = params['%s.vertex_positions' % self.lens_key].detach().reshape(-1, 3).clone()
        self.initial_baffle_vertices = params['%s.vertex_positions' % self.baffle_key].detach().reshape(-1, 3).clone()

    def update(self, params, optimizer, ext_params=None):
        """lens.py:509-568: the materials' values into the interface BSDF, the optimised surface scalars into the mesh vertices.
        `ext_params`: shape parameters set directly (the constrained rear surface, lens.py:540-543).  A material that is not optimised keeps
        plain floats: assigning an unchanged float marks nothing dirty, so a constant glass costs no upload per iteration."""
        torch = _torch()
        for comp, mat in (("int", self.int_material), ("ext", self.ext_material)):
            for short, name in (("ior", "ior_d"), ("V_d", "V_d")):
                key = '%s.bsdf.%s_%s' % (self.lens_key, comp, name); v = mat.params[short]
                if torch.is_tensor(v):
                    params[key] = _as1(v, self.initial_lens_vertices)
                elif self._pushed_materials.get(key) != float(v):
                    params[key] = _as1(v, self.initial_lens_vertices); self._pushed_materials[key] = float(v)
        if self.param_keys_to_opt_keys is None:
            return
        for var_key, opt_key in self.param_keys_to_opt_keys.items():
            self.surface.params[var_key] = optimizer[opt_key]
        if ext_params is not None:
            for k, v in ext_params.items():
                self.surface.params[k] = v
        # A surface none of whose scalars is optimised keeps plain floats; with the same floats as at the last push its two meshes are
        # already in place (20 of the 22 surfaces of the wide-angle design of report_nikon_exp.ipynb): no sag launch, no upload, no new
        # normals, and the refit finds the meshes clean.
        if all(not torch.is_tensor(v) for v in self.surface.params.values()):
            state = tuple(sorted((k, float(v)) for k, v in self.surface.params.items()))
            pushed = _pushed_states(params)                                   # (kept with the scene's parameters: another scene starts from its own vertices)
            if pushed.get(self.lens_key) == state:
                return
            pushed[self.lens_key] = state
        params['%s.vertex_positions' % self.lens_key] = surface_positions(self.surface, self.initial_lens_vertices).reshape(-1)
        # the baffle follows the surface's rim in z (lens.py:553-565): every baffle vertex takes z(radial_extent, 0)
        if getattr(self, "_rim_xy", None) is None or self._rim_xy.shape[0] != self.initial_baffle_vertices.shape[0] or self._rim_xy.device != self.initial_baffle_vertices.device:
            self._rim_xy = torch.tensor([self.surface.radial_extent, 0.0], dtype=torch.float32, device=self.initial_baffle_vertices.device).repeat(self.initial_baffle_vertices.shape[0], 1)
        params['%s.vertex_positions' % self.baffle_key] = surface_positions(self.surface, self.initial_baffle_vertices, self._rim_xy).reshape(-1)


_SAG_KEYS = ('c', 'K', 'z0', 'a4', 'a6', 'a8', 'a10', 'a12', 'a14', 'a16')


def surface_positions(surface, base_positions, eval_xy=None):
    """New vertex positions of a surface mesh, (x0, y0, z(x, y)) (LensElement.update, lens.py:509-568).  On the GPU one launch of
    mi355_lens_sag (and one of mi355_lens_sag_backward when the loss is back-propagated) instead of the ~25 elementwise torch launches of
    surface.compute_z and as many again in its autograd graph; on the CPU, or for a surface type the kernel does not know, plain torch.
    `eval_xy` (n x 2): evaluate the sag there instead of at the vertices' own (x, y) -- a baffle follows the surface's rim."""
    torch = _torch()
    if base_positions.is_cuda and set(surface.params.keys()) <= set(_SAG_KEYS):
        like = base_positions
        p10 = torch.cat([_as1(surface.params.get(k, 0.0), like) for k in _SAG_KEYS])
        return _sag_op().apply(p10, base_positions, eval_xy, float(surface.radial_extent))
    x, y = (base_positions[:, 0], base_positions[:, 1]) if eval_xy is None else (eval_xy[:, 0], eval_xy[:, 1])
    z = surface.compute_z(x, y)
    return torch.stack([base_positions[:, 0], base_positions[:, 1], z.expand(base_positions.shape[0])], dim=1)


_SAG_OP = None


def _sag_op():
    global _SAG_OP
    if _SAG_OP is not None:
        return _SAG_OP
    import ctypes as C
    torch = _torch()

    class _SagOp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, p10, base, eval_xy, R):
            base = base.detach().contiguous(); p = p10.detach().contiguous().float()
            exy = None if eval_xy is None else eval_xy.detach().contiguous()
            out = torch.empty_like(base)
            st = C.c_void_p(torch.cuda.current_stream(base.device).cuda_stream)
            abi.check(abi.lib().mi355_lens_sag(C.c_void_p(exy.data_ptr()) if exy is not None else None, C.c_void_p(base.data_ptr()), base.shape[0], R,
                                                C.c_void_p(p.data_ptr()), C.c_void_p(out.data_ptr()), st))
            ctx.saved = (p, base, exy, R)
            return out

        @staticmethod
        def backward(ctx, grad_out):
            p, base, exy, R = ctx.saved
            g = grad_out.contiguous().float(); gp = torch.zeros(10, dtype=torch.float32, device=base.device)
            st = C.c_void_p(torch.cuda.current_stream(base.device).cuda_stream)
            abi.check(abi.lib().mi355_lens_sag_backward(C.c_void_p(exy.data_ptr()) if exy is not None else None, C.c_void_p(base.data_ptr()), base.shape[0], R,
                                                         C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(gp.data_ptr()), st))
            return gp, None, None, None
    _SAG_OP = _SagOp
    return _SAG_OP


def _as1(v, like):
    torch = _torch()
    if torch.is_tensor(v):
        return v.reshape(1).to(like.device)
    return torch.tensor([float(v)], dtype=torch.float32, device=like.device)


class ApertureElement:
    """lens.py:585-744: an aperture stop -- no refracting surface, only the black baffle frame around the opening."""

    def __init__(self, element_id, surface, N=5, is_world_facing=True):
        self.subdiv_level = N; self.id = element_id; self.surface = surface
        self.param_keys_to_opt_keys = None; self.is_world_facing = is_world_facing
        self.active_optvars = list(self.surface.params.keys())

    def initialize_geometry(self):
        _, _, V_ap, F_ap, _ = create_surface_geometry(N=self.subdiv_level, r_element=self.surface.radial_extent, compute_z=self.surface.compute_z_np, c=0.0,
                                                      flip_normals=not self.is_world_facing, baffle_radius=1.1 * self.surface.radial_extent)
        self.V_ap, self.F_ap = V_ap, F_ap

    def add_to_scene(self, scene_dict):
        self.ap_key = "lens%d_AP" % self.id
        if 'black-bsdf' not in scene_dict:
            scene_dict['black-bsdf'] = {'type': 'diffuse', 'reflectance': 0.0}
        if self.ap_key in scene_dict:
            raise KeyError("Aperture `%s` already exists in scene!" % self.ap_key)
        scene_dict[self.ap_key] = {'type': 'mesh', 'vertex_positions': self.V_ap.astype(np.float32), 'faces': self.F_ap.astype(np.uint32),
                                   'compute_vertex_normals': True, 'bsdf': {'type': 'ref', 'id': 'black-bsdf'}}

    def add_to_optimizer(self, optimizer, device):
        torch = _torch()
        self.param_keys_to_opt_keys = self.param_keys_to_opt_keys or {}
        for var_name, value in self.surface.params.items():
            if var_name not in self.active_optvars:
                continue
            key = 'ap%d_%s' % (self.id, var_name)
            if key in optimizer:
                raise KeyError("Variable %s already exists in optimizer!" % key)
            optimizer[key] = torch.tensor([float(value)], dtype=torch.float32, device=device)
            self.param_keys_to_opt_keys[var_name] = key

    def save_init_state(self, params):
        self.initial_ap_vertices = params['%s.vertex_positions' % self.ap_key].detach().reshape(-1, 3).clone()

    def update(self, params, optimizer, ext_params=None):
        torch = _torch()
        if self.param_keys_to_opt_keys is None:
            return
        if all(not torch.is_tensor(v) for v in self.surface.params.values()):      # as LensElement.update: an unchanged stop stays in place
            state = tuple(sorted((k, float(v)) for k, v in self.surface.params.items()))
            pushed = _pushed_states(params)
            if pushed.get(self.ap_key) == state:
                return
            pushed[self.ap_key] = state
        x0 = self.initial_ap_vertices
        rim_xy = torch.tensor([self.surface.radial_extent, 0.0], dtype=torch.float32, device=x0.device).repeat(x0.shape[0], 1)
        params['%s.vertex_positions' % self.ap_key] = surface_positions(self.surface, x0, rim_xy).reshape(-1)

    def remove_optvar(self, param_name):
        if param_name in self.active_optvars:
            self.active_optvars.remove(param_name)

    def remove_all_optvars(self):
        self.active_optvars = []


def _f(v):
    """A lens scalar as a Python float (optimised scalars are 1-element tensors)."""
    return float(v.detach().reshape(-1)[0]) if _torch().is_tensor(v) else float(v)


def abbe_to_cauchy(nd, Vd, lbdas=(0.5893, 0.48613, 0.65627)):
    """lens.py:1251-1255: Cauchy coefficients (A, B) of n(lambda) = A + B / lambda^2 from the index at the d line and the Abbe number."""
    lbda_D, lbda_F, lbda_C = lbdas
    B = (nd - 1) / (Vd * (lbda_F ** -2 - lbda_C ** -2))
    A = nd - B * lbda_D ** -2
    return A, B


class LensSystem:
    """lens.py:752-1255: surfaces ordered from the film towards the world; materials[i] lies between surface i and i + 1, air outside.
    Carries the paraxial ray-transfer matrices of the system (compute_paraxial_quantities), which the design problem uses to place its
    sources (entrance pupil) and to check the geometric tracer."""
    SUBDIV_LEVELS = 7

    def __init__(self, surfaces, materials, aperture_index=None, subdiv_levels=None):
        if len(materials) != len(surfaces) - 1:
            raise AssertionError("Material and surface lists do not match: len(materials)=%d, len(surfaces)=%d" % (len(materials), len(surfaces)))
        if aperture_index is None or aperture_index < 0:
            self.has_aperture = False; aperture_index = len(surfaces) - 1
        else:
            self.has_aperture = True
        materials = [LensMaterial()] + list(materials)
        n_mat = len(materials)
        N = self.SUBDIV_LEVELS if subdiv_levels is None else subdiv_levels
        self.elements = []
        for i in range(len(surfaces)):
            if self.has_aperture and i == aperture_index:
                self.elements.append(ApertureElement(N=N, element_id=len(self.elements), surface=surfaces[i], is_world_facing=False))
            else:
                self.elements.append(LensElement(N=N, element_id=len(self.elements), surface=surfaces[i], ext_material=materials[i], int_material=materials[(i + 1) % n_mat],
                                                 is_world_facing=False))
        self.materials = materials
        self.rear_z = _f(surfaces[0].params['z0']); self.front_z = _f(surfaces[-1].params['z0'])
        self.front_radial_extent = surfaces[-1].radial_extent
        self.aperture_index = aperture_index
        self.compute_paraxial_quantities()

    # ---- paraxial optics (2 x 2 ray-transfer matrices acting on (y, u); float64 on the host) ------------------------------------------
    def _element_matrices(self, idx):
        """lens.py:821-840: refraction at surface idx (film-side medium -> world-side medium) after the transit from the previous one."""
        n_mat = len(self.materials)
        z_prev = 0.0 if idx == 0 else _f(self.elements[idx - 1].surface.params['z0'])
        thickness = _f(self.elements[idx].surface.params['z0']) - z_prev
        curvature = -self.elements[idx].surface.get_curvature()
        ior_i = _f(self.materials[idx].params['ior']); ior_f = _f(self.materials[(idx + 1) % n_mat].params['ior'])
        lens_matrix = np.array([[1.0, 0.0], [-(ior_f - ior_i) * curvature / ior_f, ior_i / ior_f]])
        transit_matrix = np.array([[1.0, thickness], [0.0, 1.0]])
        return lens_matrix, transit_matrix

    def compute_paraxial_quantities(self):
        """lens.py:810-880"""
        M_film2front = np.eye(2); M_rear2front = np.eye(2); M_film2ap = np.eye(2); M_ap2front = np.eye(2)
        for idx in range(len(self.elements)):
            lens_matrix, transit_matrix = self._element_matrices(idx)
            elem = lens_matrix @ transit_matrix
            M_film2front = elem @ M_film2front
            if idx < self.aperture_index:
                M_film2ap = elem @ M_film2ap
            elif idx == self.aperture_index:
                M_film2ap = transit_matrix @ M_film2ap; M_ap2front = lens_matrix @ M_ap2front
            else:
                M_ap2front = elem @ M_ap2front
            M_rear2front = (lens_matrix if idx == 0 else elem) @ M_rear2front
        self.ray_matrix_film2front = M_film2front
        self.ray_matrix_front2film = np.linalg.inv(M_film2front)
        self.ray_matrix_rear2front = M_rear2front
        self.pupil_rear_matrix = M_film2ap; self.pupil_front_matrix = M_ap2front
        self.exit_pupil_position, self.entrance_pupil_position, self.exit_pupil_radius, self.entrance_pupil_radius = self.compute_pupils()

    def get_EFL_paraxial_matrix(self):
        """lens.py:884-912: the system without its rear surface (and without the transit in front of surface 1)."""
        M = np.eye(2)
        for idx in range(1, len(self.elements)):
            lens_matrix, transit_matrix = self._element_matrices(idx)
            M = (lens_matrix if idx == 1 else lens_matrix @ transit_matrix) @ M
        return M

    def trace_paraxial_film2front(self, ray):
        """(y, u) from the film plane z = 0 to the last interface; `ray` = 2 x n array or a pair (lens.py:915-921)."""
        return self.ray_matrix_film2front @ np.asarray(ray, dtype=np.float64)

    def trace_paraxial_front2film(self, ray):
        """(y, u) from the last interface (z = front_z) back onto the film plane; the ray is oriented film -> world (lens.py:923-929)."""
        return self.ray_matrix_front2film @ np.asarray(ray, dtype=np.float64)

    def compute_BFL(self):
        return -self.ray_matrix_rear2front[1, 1] / self.ray_matrix_rear2front[1, 0]

    def compute_FFL(self):
        return -self.ray_matrix_rear2front[0, 0] / self.ray_matrix_rear2front[1, 0]

    def compute_EFL(self):
        return -1.0 / self.ray_matrix_rear2front[1, 0]

    def compute_pupils(self):
        """lens.py:956-968: paraxial exit / entrance pupil positions and radii."""
        z_exit = self.pupil_rear_matrix[0, 1] / self.pupil_rear_matrix[0, 0]
        z_entrance = _f(self.elements[-1].surface.params['z0']) - self.pupil_front_matrix[0, 1] / self.pupil_front_matrix[1, 1]
        ap_radius = self.elements[self.aperture_index].surface.radial_extent
        return z_exit, z_entrance, ap_radius / self.pupil_rear_matrix[0, 0], ap_radius / self.pupil_front_matrix[1, 1]

    def _differentiable_EFL_matrix(self):
        """get_EFL_paraxial_matrix (lens.py:884-912) on the scalars AS THEY ARE: optimised scalars are 1-element tensors with a graph behind them
        (after update()), and the constrained rear surface must stay a differentiable function of them -- in the reference these are Dr.Jit
        floats and AD follows the constraint.  2 x 2 products spelled out so that floats and tensors mix."""
        def val(x):
            return x.reshape(()) if _torch().is_tensor(x) else float(x)
        M = ((1.0, 0.0), (0.0, 1.0)); n_mat = len(self.materials)
        for idx in range(1, len(self.elements)):
            thickness = val(self.elements[idx].surface.params['z0']) - val(self.elements[idx - 1].surface.params['z0'])
            curvature = -val(self.elements[idx].surface.params['c']) / self.elements[idx].surface.radial_extent
            ior_i = val(self.materials[idx].params['ior']); ior_f = val(self.materials[(idx + 1) % n_mat].params['ior'])
            m10 = -(ior_f - ior_i) * curvature / ior_f; m11 = ior_i / ior_f
            # lens_matrix = [[1, 0], [m10, m11]]; transit = [[1, t], [0, 1]]; elem = lens @ transit = [[1, t], [m10, m10 t + m11]]
            E = ((1.0, 0.0), (m10, m11)) if idx == 1 else ((1.0, thickness), (m10, m10 * thickness + m11))
            M = ((E[0][0] * M[0][0] + E[0][1] * M[1][0], E[0][0] * M[0][1] + E[0][1] * M[1][1]),
                 (E[1][0] * M[0][0] + E[1][1] * M[1][0], E[1][0] * M[0][1] + E[1][1] * M[1][1]))
        return M

    def get_rear_surface_params(self, f):
        """lens.py:970-997: curvature (dimensionless) and axial position of the rear surface that give the system the focal length `f`
        and put its paraxial focus on the film plane.  Plain floats while every scalar involved is one; 1-element tensors that carry the
        constraint's derivative as soon as one of them is an optimiser variable (ConstrainedEFLProblem)."""
        torch = _torch()
        involved = [self.materials[0].params['ior'], self.materials[1].params['ior']] + [m.params['ior'] for m in self.materials[1:]] \
            + [e.surface.params[k] for e in self.elements[1:] for k in ('c', 'z0')]
        if any(torch.is_tensor(v) and v.requires_grad for v in involved):
            M = self._differentiable_EFL_matrix(); C, D = M[1][0], M[1][1]
            val = lambda x: x.reshape(()) if torch.is_tensor(x) else float(x)
            k = val(self.materials[0].params['ior']) / val(self.materials[1].params['ior'])
            z2 = val(self.elements[1].surface.params['z0'])
            surf_z0 = f * k * (C * z2 + D) / (1.0 + f * k * C)
            surf_c = -(1.0 / f + C) / ((C * (z2 - surf_z0) + D) * (1 - k))
            as1 = lambda x: x.reshape(1) if torch.is_tensor(x) else x
            return {'c': as1(surf_c * self.elements[0].surface.radial_extent), 'z0': as1(surf_z0)}
        Mk = self.get_EFL_paraxial_matrix()
        C, D = Mk[1, 0], Mk[1, 1]
        k = _f(self.materials[0].params['ior']) / _f(self.materials[1].params['ior'])
        z2 = _f(self.elements[1].surface.params['z0'])
        surf_z0 = f * k * (C * z2 + D) / (1.0 + f * k * C)
        surf_c = -(1.0 / f + C) / ((C * (z2 - surf_z0) + D) * (1 - k))
        return {'c': surf_c * self.elements[0].surface.radial_extent, 'z0': surf_z0}

    def size(self):
        return len(self.elements) - 1 if self.has_aperture else len(self.elements)

    def initialize_geometry(self):
        for e in self.elements:
            e.initialize_geometry()

    def add_to_scene(self, scene_dict, housing=True):
        """lens.py:1065-1095: the elements, and four black walls around them (`camera_housing_i`: the unit square mesh under
        scale(rmax, -rmax, zmax / 2) . look_at(origin = z + o, target = z, up = z))."""
        rmax = 0.0
        for e in self.elements:
            e.add_to_scene(scene_dict); rmax = max(rmax, e.surface.radial_extent)
        if not housing:
            return
        T = ScalarTransform4f
        zmax = self.front_z * 1.1; rmax = rmax * 1.1
        z = np.array([0.0, 0.0, 1.0])
        for i, origin in enumerate(([1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0])):
            tw = T.scale([rmax, -rmax, zmax / 2]) @ T.look_at(origin=(z + np.array(origin, dtype=np.float64)).tolist(), target=z.tolist(), up=[0, 0, 1])
            scene_dict['camera_housing_%d' % i] = dict(unit_square_mesh(), to_world=tw, bsdf={'type': 'ref', 'id': 'black-bsdf'})

    def add_to_optimizer(self, optimizer, device):
        for m in self.materials:
            m.add_to_optimizer(optimizer, device)
        for e in self.elements:
            e.add_to_optimizer(optimizer, device)

    def save_init_state(self, params):
        for e in self.elements:
            e.save_init_state(params)

    def disable_all_surfaces(self):
        for e in self.elements:
            e.remove_all_optvars()

    def disable_all_materials(self):
        for m in self.materials:
            m.remove_all_optvars()

    def disable_surface_at_index(self, index): self.elements[index].remove_all_optvars()
    def disable_material_at_index(self, index): self.materials[index].remove_all_optvars()

    def disable_surface_vars(self, param_name):
        for e in self.elements:
            e.remove_optvar(param_name)

    def disable_material_vars(self, param_name):
        for m in self.materials:
            m.remove_optvar(param_name)

    def disable_var_in_surface(self, param_name, index): self.elements[index].remove_optvar(param_name)
    def disable_var_in_material(self, param_name, index): self.materials[index].remove_optvar(param_name)

    def update(self, params, optimizer):
        """Materials first, then the elements (lens.py:516-517 "materials update *must* be performed before LensElement update"), then the
        paraxial matrices and the scene (lens.py:1009-1021)."""
        for m in self.materials:
            m.update(optimizer)
        for e in self.elements:
            e.update(params, optimizer)
        self.compute_paraxial_quantities()
        params.update()

    def fixed_EFL_update(self, params, optimizer, efl):
        """lens.py:1023-1042: as update(), with the rear surface's c and z0 dictated by the focal-length constraint."""
        for m in self.materials:
            m.update(optimizer)
        if efl is not None:
            for e in self.elements[1:]:
                e.update(params, optimizer)
            self.elements[0].update(params, optimizer, self.get_rear_surface_params(efl))
        else:
            for e in self.elements:
                e.update(params, optimizer)
        self.compute_paraxial_quantities()
        params.update()

    def describe(self):
        """What LensSystem.print() shows (lens.py:1225-1248): Cauchy pairs of the materials world -> film, then curvature radii, thicknesses
        and extents world -> film."""
        mats = []
        for m in reversed(self.materials):
            mats.append((1.000277, 0.0) if m.name == "air" else tuple(float(x) for x in abbe_to_cauchy(_f(m.params['ior']), _f(m.params['V_d']))))
        curv = [e.surface.get_curvature() for e in self.elements]
        r_c = [1.0 / c if c != 0.0 else float("inf") for c in curv]
        z0s = [0.0] + [_f(e.surface.get_z0()) for e in self.elements]
        t = [z0s[i + 1] - z0s[i] for i in range(len(self.elements))]
        return {"materials": mats, "r_c": r_c[::-1], "t": t[::-1], "r_e": [e.surface.radial_extent for e in self.elements][::-1]}


def unit_square_mesh():
    """`meshes/rectangle.obj` of the tutorial scenes (absent from the tree): the square [-1, 1]^2 in the plane z = 0 as two triangles,
    normal +z.  A mesh, not the analytic `rectangle`: a hit point on it is the barycentric combination of vertices with z = 0, hence has
    z == 0 exactly, which the geometric tracer's `dr.allclose(0, si.p.z)` (prb_basic.py:138) relies on."""
    return {'type': 'mesh', 'vertex_positions': np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float32),
            'vertex_normals': np.array([[0, 0, 1]] * 4, dtype=np.float32), 'vertex_texcoords': np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32),
            'faces': np.array([[0, 1, 2], [2, 3, 0]], dtype=np.uint32)}


# ------------------------------------------------------------------------------------------------
# losses.py / design_problem.py
# ------------------------------------------------------------------------------------------------
_SPOT_OP = None


def _spot_op():
    """losses.py:31-41 as two launches (mi355_spot_loss / _backward) instead of ~25 elementwise and reduction kernels per field."""
    global _SPOT_OP
    if _SPOT_OP is not None:
        return _SPOT_OP
    import ctypes as C
    torch = _torch()

    class _SpotOp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, image):
            img = image.detach().contiguous().float()
            out = torch.empty(5, dtype=torch.float32, device=img.device)
            with torch.cuda.device(img.device):
                abi.check(abi.lib().mi355_spot_loss(C.c_void_p(img.data_ptr()), img.shape[0], img.shape[1], C.c_void_p(out.data_ptr()),
                                                    C.c_void_p(torch.cuda.current_stream(img.device).cuda_stream)))
            ctx.saved = (out, img.shape)
            ctx.mark_non_differentiable(out)
            return out[0].clone(), out

        @staticmethod
        def backward(ctx, grad_loss, _grad_out):
            out, shape = ctx.saved
            g = grad_loss.detach().reshape(1).contiguous().float(); gi = torch.empty(shape, dtype=torch.float32, device=out.device)
            with torch.cuda.device(out.device):
                abi.check(abi.lib().mi355_spot_loss_backward(C.c_void_p(out.data_ptr()), C.c_void_p(g.data_ptr()), shape[0], shape[1], C.c_void_p(gi.data_ptr()),
                                                             C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)))
            return gi
    _SPOT_OP = _SpotOp
    return _SPOT_OP


def rms_loss_and_center(image):
    """losses.py:31-41: squared RMS radius (in pixels) of the spot in `image` (H x W x 3 torch tensor) and its centroid (row, column); centroid
    and normalisation detached.  Device images go through one kernel each way; the torch expression below is the same arithmetic."""
    torch = _torch()
    if image.is_cuda and image.dim() == 3 and image.shape[2] == 3:
        loss, out = _spot_op().apply(image)
        return loss, out[1], out[2]
    scaled = image / image.detach().mean()
    H, W = image.shape[0], image.shape[1]
    ii, jj = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=image.device), torch.arange(W, dtype=torch.float32, device=image.device), indexing='ij')
    I = (scaled[:, :, 0] + scaled[:, :, 1] + scaled[:, :, 2]) / 3.0
    inv_I_sum = 1.0 / I.sum()
    ibar = ((ii * I).sum() * inv_I_sum).detach(); jbar = ((jj * I).sum() * inv_I_sum).detach()
    return (I * ((ii - ibar) ** 2 + (jj - jbar) ** 2)).sum() * inv_I_sum, ibar, jbar


def rms_loss(image):
    """losses.py:8-29"""
    return rms_loss_and_center(image)[0]


class Ray3f:
    """The two fields of mi.Ray3f the geometric tracer reads (o, d as n x 3 arrays; maxt = Largest)."""

    def __init__(self, o, d):
        self.o = np.asarray(o, dtype=np.float64).reshape(-1, 3); self.d = np.asarray(d, dtype=np.float64).reshape(-1, 3)


class FieldSource:
    """design_problem.py:12-279: one field angle = a `disk` with a `directionalarea` emitter in front of the lens, aimed at the entrance
    pupil, and a `perspective` sensor just in front of the film, placed where the source's central ray lands."""

    def __init__(self, name, origin, target, radius, lens_system, resolution, init_camera_z, tracer, fov=135.0, radiance=0.05):
        self.source_key = "source_" + name; self.sensor_key = "sensor_" + name
        self.origin = np.asarray(origin, dtype=np.float64); self.target = np.asarray(target, dtype=np.float64)
        self.radius = radius; self.radiance = radiance
        self.near_clip = min(0.005, lens_system.rear_z * 0.1)
        self.resolution = resolution
        self.fov = fov                                   # 135 degrees, "for nikon" (:39-40; 45 is the commented-out default)
        self.camera_pos = np.zeros(3)
        self.fov_changed = False; self.pos_changed = False
        self.lens_system = lens_system
        self.sensor_dict = self.get_sensor_dict(tracer, init_camera_z)
        self.losses = []

    def add_to_scene(self, scene_dict):
        """:51-83"""
        T = ScalarTransform4f
        for key in (self.source_key, self.sensor_key):
            if key in scene_dict:
                raise KeyError("`%s` already exists in scene!" % key)
        scene_dict[self.source_key] = {
            'type': 'disk', 'to_world': T.look_at(target=self.target.tolist(), origin=self.origin.tolist(), up=[0, 1, 0]) @ T.scale([self.radius, self.radius, 1]),
            'bsdf': {'type': 'ref', 'id': 'black-bsdf'}, 'emitter': {'type': 'directionalarea', 'radiance': self.radiance}}
        scene_dict[self.sensor_key] = self.sensor_dict

    def update(self, params):
        """:86-132: move / zoom this field's camera through the scene parameters (to_world, x_fov).  Host-side values; the device replica
        receives the two parameters at the next params.update()."""
        T = ScalarTransform4f
        if self.pos_changed:
            params["%s.to_world" % self.sensor_key] = T.look_at(target=[float(self.camera_pos[0]), float(self.camera_pos[1]), -1.0], origin=[float(x) for x in self.camera_pos], up=[0, 1, 0])
            self.pos_changed = False
        if self.fov_changed:
            params["%s.x_fov" % self.sensor_key] = self.fov
            self.fov_changed = False

    def upsample_sensor(self, zoom_factor=2.0):
        """:135-140: zoom on the centre of the image field."""
        self.fov = math.degrees(2.0 * math.atan(math.tan(0.5 * math.radians(self.fov)) / zoom_factor))
        self.fov_changed = True

    @staticmethod
    def createSourceArray(lens_system, resolution, tracer, num_sources=1, max_field_angle=0.0, init_camera_z=0.9, fov=135.0):
        """:142-181: `num_sources` field angles from 0 to max_field_angle (degrees) in the xz plane, every source aimed at the entrance
        pupil from a distance at which neighbouring disks do not overlap."""
        source_radius = lens_system.front_radial_extent
        entrance_pupil_position = np.array([0.0, 0.0, lens_system.entrance_pupil_position])
        max_field_angle = np.deg2rad(np.abs(max_field_angle))
        field_angles = np.linspace(0.0, max_field_angle, num_sources)
        if num_sources > 1:
            dtheta = max_field_angle / (num_sources - 1)
            source_distance = 1.1 * source_radius / np.tan(0.5 * dtheta)
        else:
            source_distance = max(10.0, 1.1 * lens_system.front_z)
        sources = []
        for field_angle in field_angles:
            direction = np.array([np.sin(field_angle), 0.0, np.cos(field_angle)])
            origin = entrance_pupil_position + source_distance * direction
            name = ("%04.1f-deg" % np.rad2deg(field_angle)).replace(".", "_")
            sources.append(FieldSource(name=name, origin=origin, target=entrance_pupil_position, radius=source_radius, lens_system=lens_system,
                                       resolution=resolution, tracer=tracer, init_camera_z=init_camera_z, fov=fov))
        return sources

    def get_central_ray(self):
        """:183-189"""
        d = self.target - self.origin
        return Ray3f(self.origin, d / np.linalg.norm(d))

    def get_sensor_dict(self, tracer, init_camera_z):
        """:192-247: the camera sits in front of the film where the central ray of the source lands (geometric trace through the lens)."""
        T = ScalarTransform4f
        _, _, is_valid, film_pos = tracer(self.get_central_ray())
        film_pos = np.asarray(film_pos.detach().cpu() if hasattr(film_pos, "detach") else film_pos, dtype=np.float64).ravel()
        is_valid = np.asarray(is_valid.detach().cpu() if hasattr(is_valid, "detach") else is_valid).ravel()
        if not np.all(is_valid):
            raise AssertionError("Source's central ray was not transmitted through the lens system! is_valid=%s" % is_valid)
        z_camera = max(init_camera_z * self.lens_system.rear_z, 1.1 * self.near_clip)
        self.camera_pos = np.array([film_pos[0], film_pos[1], z_camera])
        self.init_pixel_size = self.camera_pos[2] * math.tan(math.radians(0.5 * self.fov)) / (self.resolution[0] / 2)
        return {'type': 'perspective', 'near_clip': self.near_clip, 'far_clip': z_camera + 1.0, 'fov': self.fov,
                'to_world': T.look_at(target=[float(self.camera_pos[0]), float(self.camera_pos[1]), -1.0], origin=[float(x) for x in self.camera_pos], up=[0, 1, 0]),
                'sampler': {'type': 'independent', 'sample_count': 512},
                'film': {'type': 'hdrfilm', 'width': self.resolution[0], 'height': self.resolution[1], 'pixel_format': 'rgb', 'rfilter': {'type': 'tent'}}}

    def get_pixel_size(self):
        """:249-256: physical size of a camera pixel on the film plane, in millimetres."""
        return float(self.camera_pos[2]) * math.tan(math.radians(0.5 * self.fov)) / (self.resolution[0] / 2)

    def center_camera_on_pixel(self, i, j):
        """:258-265: when the spot's centroid column is more than a tenth of the image to the LEFT of the centre (the test is one-sided in
        the reference), slide the camera in x so that the spot is centred."""
        pixel_size = self.get_pixel_size()
        delta_x = self.resolution[0] / 2 - 0.5 - float(j)
        self.pos_changed = bool(delta_x / self.resolution[0] > 0.1)
        if self.pos_changed:
            self.camera_pos[0] -= delta_x * pixel_size

    def spot_loss(self, image):
        """The device half of evaluate_rms_loss: the spot's squared RMS radius in mm^2 (differentiable) and a 3-vector (pixel loss, centroid row,
        centroid column) for the host half.  Nothing is read back here, so the next field's render can be queued behind this one."""
        torch = _torch()
        pixel_loss, ibar, jbar = rms_loss_and_center(image)
        physical_loss = pixel_loss * self.get_pixel_size() ** 2
        return physical_loss, torch.stack([pixel_loss.detach().reshape(()), ibar.reshape(()), jbar.reshape(())])

    def apply_spot_stats(self, host_loss, host_i, host_j, recenter=False):
        """The host half (:258-279): re-centre the camera on the spot, record the loss, return the upsample flag (spot smaller than a tenth of
        the image).  None of it feeds back into the current iteration's renders, so it may run after all of them."""
        if recenter:
            self.center_camera_on_pixel(host_i, host_j)
        self.losses.append(host_loss * self.get_pixel_size() ** 2)
        return bool(host_loss < (self.resolution[0] * 0.1) ** 2)

    def evaluate_rms_loss(self, image, recenter=False):
        """:267-279: the spot's squared RMS radius in mm^2 and the upsample flag -- the reference's immediate form (one read-back per call)."""
        physical_loss, stats = self.spot_loss(image)
        host_loss, host_i, host_j = stats.tolist()
        return physical_loss, self.apply_spot_stats(host_loss, host_i, host_j, recenter)


class DesignProblem:
    """design_problem.py:283-584: `num_sources` field sources, one sensor each, rendered one after the other in every iteration."""

    def __init__(self, lens_system, resolution, spp, learning_rate, iters, film_diagonal=35.0, num_sources=1, max_field_angle=5.0, init_camera_z=0.9, fov=135.0):
        lens_system.initialize_geometry()
        self.lens_system = lens_system; self.resolution = resolution; self.spp = spp; self.lr = learning_rate; self.iters = iters
        self.film_diagonal = film_diagonal; self.num_sources = num_sources; self.max_field_angle = max_field_angle; self.init_camera_z = init_camera_z; self.fov = fov
        self.scene = self.params = self.optimizer = None
        self.geo_check = None

    def _get_integrator(self, integrator_type):
        return {'type': integrator_type, 'max_depth': self.lens_system.size() + 2, 'hide_emitters': False}      # :355-362

    def _build_preliminary_scene(self, integrator_type):
        """:364-401: lenses and film plane only."""
        T = ScalarTransform4f
        d = {'type': 'scene', 'integrator': self._get_integrator(integrator_type),
             'white-bsdf': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': (1, 1, 1)}},
             'black-bsdf': {'type': 'diffuse', 'reflectance': 0.0},
             'film_plane': dict(unit_square_mesh(), to_world=T.look_at(target=[0, 0, 1], origin=[0, 0, 0], up=[0, 1, 0]).scale([self.film_diagonal, self.film_diagonal, 1]),
                                bsdf={'type': 'ref', 'id': 'white-bsdf'})}
        self.lens_system.add_to_scene(d)
        return d

    def add_field_sources(self, scene_dict, geo_tracer):
        """:310-327"""
        self.field_sources = FieldSource.createSourceArray(lens_system=self.lens_system, num_sources=self.num_sources, max_field_angle=self.max_field_angle,
                                                           resolution=self.resolution, tracer=geo_tracer, init_camera_z=self.init_camera_z, fov=self.fov)
        for fs in self.field_sources:
            fs.add_to_scene(scene_dict)

    def test_geo_tracer(self, tracer, N=100):
        """:430-472 (__test_geo_tracer): N near-axis rays through the lens MESHES against the paraxial matrices.  Returns (max |y_exact -
        y_paraxial|, max |u_exact - u_paraxial|); the reference prints both and asks for atol 2e-5."""
        ls = self.lens_system
        z_init = ls.front_z * 1.1; r_max = ls.front_radial_extent * 0.001; theta_max = 0.001 * math.pi
        idx = np.arange(N, dtype=np.uint64)
        rs = r_max * (2.0 * np.array([radical_inverse_2(int(i)) for i in idx]) - 1.0)
        thetas = theta_max * (2.0 * np.array([sobol_2(int(i)) for i in idx]) - 1.0)
        d = -np.stack([np.tan(thetas), np.zeros(N), np.ones(N)], axis=1)          # "negate direction since ray is traveling backwards"
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = np.stack([rs, np.zeros(N), np.full(N, z_init)], axis=1)
        _, d_exact, _, out_exact = tracer(Ray3f(o, d))
        d_exact = np.asarray(d_exact.detach().cpu() if hasattr(d_exact, "detach") else d_exact, dtype=np.float64)
        out_exact = np.asarray(out_exact.detach().cpu() if hasattr(out_exact, "detach") else out_exact, dtype=np.float64)
        y = rs - (z_init - ls.front_z) * np.tan(thetas); u = np.tan(thetas)
        out_paraxial = ls.trace_paraxial_front2film(np.stack([y, u]))
        err_y = float(np.max(np.abs(out_exact[:, 0] - out_paraxial[0])))
        err_u = float(np.max(np.abs(d_exact[:, 0] / d_exact[:, 2] - out_paraxial[1])))
        return err_y, err_u

    def _build_scene(self):
        """:403-428: a first scene (lenses + film) carries the `prb_basic` geometric tracer that places the field cameras; the rendering
        scene uses `ptracer`."""
        from .scene import load_dict
        geo_scene = load_dict(self._build_preliminary_scene("prb_basic"))
        trace_depth = self.lens_system.size() + 1
        integ = geo_scene.integrator()
        geo_tracer = lambda ray: integ.trace(geo_scene, ray, trace_depth)
        d = self._build_preliminary_scene("ptracer")
        self.add_field_sources(d, geo_tracer)
        scene = load_dict(d)
        self.geo_check = self.test_geo_tracer(geo_tracer)
        geo_scene.release()
        return scene

    def scene_dict(self):
        """The rendering scene's dictionary (needs a device: the field cameras are placed by the geometric tracer)."""
        from .scene import load_dict
        geo_scene = load_dict(self._build_preliminary_scene("prb_basic"))
        integ = geo_scene.integrator(); depth = self.lens_system.size() + 1
        d = self._build_preliminary_scene("ptracer")
        self.add_field_sources(d, lambda ray: integ.trace(geo_scene, ray, depth))
        geo_scene.release()
        return d

    def prepare(self):
        """:481-492"""
        from .integrators import traverse
        from .optimizers import Adam
        self.scene = self._build_scene()
        self.params = traverse(self.scene)
        dev = self.params[next(iter(self.params._entries))].device
        self.optimizer = Adam(lr=self.lr)
        self.lens_system.save_init_state(self.params)
        self.lens_system.add_to_optimizer(self.optimizer, dev)
        return self

    def _update_lens_system(self):
        self.lens_system.update(self.params, self.optimizer)

    def step(self, it, recenter=True):
        """One iteration of optimize() (:508-565): update the scene from the optimizer, update the field cameras, render every field
        differentiably, back-propagate the summed spot loss (x 10000, as the notebook scales it), take the optimizer step.  Returns
        (loss, upsample_flag)."""
        from .integrators import render
        self._update_lens_system()
        for fs in self.field_sources:
            fs.update(self.params)
        import torch
        loss = None; pending = []
        # multi-GPU: `shard = (rank, world)` -- every rank traces its part of the light-sample grid, the film and the gradient buffer are
        # all-reduced inside render / render_backward (BASELINE configs[4]: "8-GPU RCCL grad all-reduce"), so every rank takes the same step
        # The fields of an iteration are rendered (and differentiated) with ONE seed (:525-537), so their cameras see the same light paths: the images
        # come from one pass over those paths and the gradients from one walk + sweep (integrators.render_fields; up to eight cameras per pass).
        shard = getattr(self, "shard", None); images = {}
        if getattr(self, "share_light_paths", True) and hasattr(self.scene.integrator(), "render_sensors"):
            from .integrators import render_fields
            per_pass = int(getattr(self, "sensors_per_pass", 8))
            for first in range(0, self.num_sources, per_pass):
                group_ids = list(range(first, min(first + per_pass, self.num_sources)))
                for k, im in zip(group_ids, render_fields(self.scene, self.params, group_ids, seed=it, spp=2 * self.spp, spp_grad=self.spp, shard=shard, group=getattr(self, "group", None))):
                    images[k] = im
        for sensor_idx in range(self.num_sources):
            image = images[sensor_idx] if sensor_idx in images else render(self.scene, self.params, seed=it, spp=2 * self.spp, spp_grad=self.spp, sensor=sensor_idx, shard=shard, group=getattr(self, "group", None))
            sensor_loss, stats = self.field_sources[sensor_idx].spot_loss(image)
            pending.append(stats)
            loss = 10000.0 * sensor_loss if loss is None else loss + 10000.0 * sensor_loss
        loss.backward()
        self.optimizer.step()
        # ONE read-back per iteration, behind everything that was queued: the centroids only move the cameras of the NEXT iteration and the
        # flag is only looked at below (the reference reads three scalars per field as it goes; same values, same order of host updates)
        host = torch.cat(pending + [loss.detach().reshape(1)]).tolist()
        upsample_flag = True
        for k, fs in enumerate(self.field_sources):
            upsample_flag = fs.apply_spot_stats(host[3 * k], host[3 * k + 1], host[3 * k + 2], recenter=recenter) and upsample_flag
        loss_value = host[-1]
        if upsample_flag:
            for fs in self.field_sources:
                fs.upsample_sensor()
        self.params.update()
        return loss_value, upsample_flag

    def optimize(self, iters=None):
        losses = []; self.upsample_steps = 0
        for it in range(self.iters if iters is None else iters):
            loss, up = self.step(it)
            losses.append(loss); self.upsample_steps += int(up)
            if it in (int(0.7 * self.iters), int(0.9 * self.iters)):
                self.optimizer.set_learning_rate(0.3 * self.optimizer.lr_default)
        return losses

    def render(self, spp, sensor_idx=0, resolution=None, zoom_factor=None):
        """:586-619: a look at the film through one field's camera, optionally zoomed and at another resolution; everything is restored."""
        from .integrators import render
        import torch
        with torch.no_grad():
            if zoom_factor is not None:
                for fs in self.field_sources:
                    fs.upsample_sensor(zoom_factor); fs.update(self.params)
            key = "%s.film" % self.field_sources[sensor_idx].sensor_key
            if resolution is not None:
                saved = (self.params[key + ".size"], self.params[key + ".crop_size"], self.params[key + ".crop_offset"])
                self.params[key + ".size"] = (resolution[0], resolution[1])
            self.params.update()
            image = render(self.scene, self.params, spp=spp, sensor=sensor_idx)
            if zoom_factor is not None:
                for fs in self.field_sources:
                    fs.upsample_sensor(1.0 / zoom_factor); fs.update(self.params)
            if resolution is not None:
                self.params[key + ".size"] = saved[0]; self.params[key + ".crop_size"] = saved[1]; self.params[key + ".crop_offset"] = saved[2]
            self.params.update()
        return image


class ConstrainedEFLProblem(DesignProblem):
    """design_problem.py:656-803: the design problem with the system's effective focal length held at `target_focal_length`.  The rear surface's
    first-order parameters (c, z0) leave the optimiser; in every iteration they are re-derived from the other surfaces and the glasses
    (LensSystem.fixed_EFL_update -> get_rear_surface_params), and the gradient follows that constraint."""

    def __init__(self, lens_system, resolution, spp, learning_rate, target_focal_length, iters, film_diagonal=35.0, num_sources=1, max_field_angle=5.0,
                 init_camera_z=0.9, fov=135.0):
        self.target_focal_length = float(target_focal_length)
        lens_system.disable_var_in_surface('c', 0); lens_system.disable_var_in_surface('z0', 0)          # :673-676
        s1 = lens_system.get_rear_surface_params(self.target_focal_length)                                 # :678-682
        lens_system.elements[0].surface.params['c'] = _f(s1['c']); lens_system.elements[0].surface.params['z0'] = _f(s1['z0'])
        lens_system.compute_paraxial_quantities()
        super().__init__(lens_system, resolution, spp, learning_rate, iters, film_diagonal=film_diagonal, num_sources=num_sources,
                         max_field_angle=max_field_angle, init_camera_z=init_camera_z, fov=fov)

    def _update_lens_system(self):
        self.lens_system.fixed_EFL_update(self.params, self.optimizer, self.target_focal_length)        # :722

    def optimize(self, iters=None):
        """:705-803: as DesignProblem.optimize, the learning rate dropping to a tenth at 50 % and at 70 % of the iterations (:786-788)."""
        losses = []; self.upsample_steps = 0
        for it in range(self.iters if iters is None else iters):
            loss, up = self.step(it)
            losses.append(loss); self.upsample_steps += int(up)
            if it in (int(0.5 * self.iters), int(0.7 * self.iters)):
                self.optimizer.set_learning_rate(0.1 * self.optimizer.lr_default)
        return losses


def radical_inverse_2(index, scramble=0):
    """mi.radical_inverse_2 (include/mitsuba/core/qmc.h:189-211, single precision): van der Corput in base 2 by bit reversal; the float is
    built from the 23 high bits ((bits >> 9) | 0x3f800000) - 1."""
    index &= 0xffffffff
    index = ((index << 16) | (index >> 16)) & 0xffffffff
    index = (((index & 0x00ff00ff) << 8) | ((index & 0xff00ff00) >> 8)) & 0xffffffff
    index = (((index & 0x0f0f0f0f) << 4) | ((index & 0xf0f0f0f0) >> 4)) & 0xffffffff
    index = (((index & 0x33333333) << 2) | ((index & 0xcccccccc) >> 2)) & 0xffffffff
    index = (((index & 0x55555555) << 1) | ((index & 0xaaaaaaaa) >> 1)) & 0xffffffff
    return float(((index ^ scramble) >> 9) / float(1 << 23))


def sobol_2(index, scramble=0):
    """mi.sobol_2 (qmc.h:217-237, single precision): the second Sobol' dimension, Float(bits) / Float(2^32)."""
    index &= 0xffffffff; v = 1 << 31
    while index != 0:
        if index & 1:
            scramble ^= v
        index >>= 1
        v ^= v >> 1
    return float(np.float32(np.float32(scramble) / np.float32(2.0 ** 32)))
