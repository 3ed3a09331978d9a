"""Host-side mirror of the fork's lens-design scripts -- the caller of BASELINE config 5 (notebooks/doublet_full_optim.ipynb):

  notebooks/scripts/lens_design/geometry.py   create_surface_geometry :296-390, get_spherical_cap :122-158, get_hemisphere_ico :44-70
  notebooks/scripts/lens_design/lens.py       ConicSurface :55-97, EvenAsphericSurface :166-249, LensMaterial :252-339,
                                              LensElement :343-581 (add_to_scene :418-470, add_to_optimizer :473-498, update :509-568),
                                              LensSystem :752-808
  notebooks/scripts/lens_design/design_problem.py   FieldSource :12-279 (directionalarea disk source + perspective sensor),
                                              DesignProblem :283-584 (prepare :481-492, optimize :495-584)
  notebooks/scripts/lens_design/losses.py     rms_loss :8-29

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
library's, nothing depends on it.  One field source (on axis) per scene: this build's Scene holds one sensor.
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


class LensElement:
    """lens.py:343-581: ONE refracting surface as a mesh (`dispersive` interface between two media) plus its black baffle mesh."""

    def __init__(self, element_id, surface, int_material, ext_material, N=5, is_world_facing=True):
        self.subdiv_level = N; self.id = element_id
        self.int_material = int_material; self.ext_material = ext_material
        self.surface = surface
        self.param_keys_to_opt_keys = None
        self.is_world_facing = is_world_facing
        self.active_optvars = list(self.surface.params.keys())

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

    def update(self, params, optimizer):
        """lens.py:509-568: the materials' values into the interface BSDF, the optimised surface scalars into the mesh vertices."""
        torch = _torch()
        params['%s.bsdf.int_ior_d' % self.lens_key] = _as1(self.int_material.params['ior'], self.initial_lens_vertices)
        params['%s.bsdf.int_V_d' % self.lens_key] = _as1(self.int_material.params['V_d'], self.initial_lens_vertices)
        params['%s.bsdf.ext_ior_d' % self.lens_key] = _as1(self.ext_material.params['ior'], self.initial_lens_vertices)
        params['%s.bsdf.ext_V_d' % self.lens_key] = _as1(self.ext_material.params['V_d'], self.initial_lens_vertices)
        if self.param_keys_to_opt_keys is None:
            return
        for var_key, opt_key in self.param_keys_to_opt_keys.items():
            self.surface.params[var_key] = optimizer[opt_key]
        x0, y0 = self.initial_lens_vertices[:, 0], self.initial_lens_vertices[:, 1]
        new_pos = torch.stack([x0, y0, self.surface.compute_z(x0, y0)], dim=1)
        params['%s.vertex_positions' % self.lens_key] = new_pos.reshape(-1)
        # the baffle follows the surface's rim in z (lens.py:553-565)
        rim = self.surface.compute_z(torch.full((1,), self.surface.radial_extent, device=x0.device), torch.zeros(1, device=x0.device))
        bx, by = self.initial_baffle_vertices[:, 0], self.initial_baffle_vertices[:, 1]
        params['%s.vertex_positions' % self.baffle_key] = torch.stack([bx, by, rim.expand(bx.shape[0])], dim=1).reshape(-1)


def _as1(v, like):
    torch = _torch()
    if torch.is_tensor(v):
        return v.reshape(1).to(like.device)
    return torch.tensor([float(v)], dtype=torch.float32, device=like.device)


class LensSystem:
    """lens.py:752-808: surfaces ordered from the film towards the world; materials[i] lies between surface i and i + 1, air outside."""
    SUBDIV_LEVELS = 7

    def __init__(self, surfaces, materials, subdiv_levels=None):
        if len(materials) != len(surfaces) - 1:
            raise AssertionError("Material and surface lists do not match: len(materials)=%d, len(surfaces)=%d" % (len(materials), len(surfaces)))
        materials = [LensMaterial()] + list(materials)
        n_mat = len(materials)
        N = self.SUBDIV_LEVELS if subdiv_levels is None else subdiv_levels
        self.elements = [LensElement(N=N, element_id=i, surface=surfaces[i], ext_material=materials[i], int_material=materials[(i + 1) % n_mat],
                                     is_world_facing=False) for i in range(len(surfaces))]
        self.materials = materials
        self.rear_z = float(surfaces[0].params['z0']); self.front_z = float(surfaces[-1].params['z0'])
        self.front_radial_extent = surfaces[-1].radial_extent

    def size(self):
        return len(self.elements)

    def initialize_geometry(self):
        for e in self.elements:
            e.initialize_geometry()

    def add_to_scene(self, scene_dict):
        for e in self.elements:
            e.add_to_scene(scene_dict)

    def add_to_optimizer(self, optimizer, device):
        for m in self.materials:
            m.add_to_optimizer(optimizer, device)
        for e in self.elements:
            e.add_to_optimizer(optimizer, device)

    def save_init_state(self, params):
        for e in self.elements:
            e.save_init_state(params)

    def disable_all_materials(self):
        for m in self.materials:
            m.remove_all_optvars()

    def update(self, params, optimizer):
        """Materials first, then the elements (lens.py:516-517 "materials update *must* be performed before LensElement update")."""
        for m in self.materials:
            m.update(optimizer)
        for e in self.elements:
            e.update(params, optimizer)


# ------------------------------------------------------------------------------------------------
# losses.py / design_problem.py
# ------------------------------------------------------------------------------------------------
def rms_loss(image):
    """losses.py:8-29: squared RMS radius (in pixels) of the spot in `image` (H x W x 3 torch tensor); centroid and normalisation detached."""
    torch = _torch()
    scaled = image / image.detach().mean()
    H, W = image.shape[0], image.shape[1]
    ii, jj = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=image.device), torch.arange(W, dtype=torch.float32, device=image.device), indexing='ij')
    I = (scaled[:, :, 0] + scaled[:, :, 1] + scaled[:, :, 2]) / 3.0
    inv_I_sum = 1.0 / I.sum()
    ibar = ((ii * I).sum() * inv_I_sum).detach(); jbar = ((jj * I).sum() * inv_I_sum).detach()
    return (I * ((ii - ibar) ** 2 + (jj - jbar) ** 2)).sum() * inv_I_sum


class FieldSource:
    """design_problem.py:12-279 for the on-axis field: a `disk` with a `directionalarea` emitter in front of the lens, aimed at it, and a
    `perspective` sensor just in front of the film looking back at it."""

    def __init__(self, lens_system, resolution, init_camera_z=0.9, fov=135.0, source_distance=None, radiance=0.05):
        self.radius = lens_system.front_radial_extent
        self.target = np.array([0.0, 0.0, lens_system.front_z])
        dist = max(10.0, 1.1 * lens_system.front_z) if source_distance is None else source_distance
        self.origin = self.target + dist * np.array([0.0, 0.0, 1.0])
        self.near_clip = min(0.005, lens_system.rear_z * 0.1)
        self.resolution = resolution; self.fov = fov; self.radiance = radiance
        z_camera = max(init_camera_z * lens_system.rear_z, 1.1 * self.near_clip)
        self.camera_pos = np.array([0.0, 0.0, z_camera])

    def add_to_scene(self, scene_dict):
        T = ScalarTransform4f
        scene_dict['source_00_0-deg'] = {
            'type': 'disk', 'to_world': T.look_at(target=self.target.tolist(), origin=self.origin.tolist(), up=[0, 1, 0]) @ T.scale([self.radius, self.radius, 1]),
            'bsdf': {'type': 'ref', 'id': 'black-bsdf'}, 'emitter': {'type': 'directionalarea', 'radiance': self.radiance}}
        scene_dict['sensor_00_0-deg'] = {
            'type': 'perspective', 'near_clip': self.near_clip, 'far_clip': float(self.camera_pos[2]) + 1.0, 'fov': self.fov,
            'to_world': T.look_at(target=[float(self.camera_pos[0]), float(self.camera_pos[1]), -1.0], origin=self.camera_pos.tolist(), up=[0, 1, 0]),
            'sampler': {'type': 'independent', 'sample_count': 512},
            'film': {'type': 'hdrfilm', 'width': self.resolution[0], 'height': self.resolution[1], 'pixel_format': 'rgb', 'rfilter': {'type': 'tent'}}}

    def get_pixel_size(self):
        return float(self.camera_pos[2]) * math.tan(math.radians(0.5 * self.fov)) / (self.resolution[0] / 2)

    def evaluate_rms_loss(self, image):
        """design_problem.py:267-279: the spot's squared RMS radius in mm^2."""
        return rms_loss(image) * self.get_pixel_size() ** 2


class DesignProblem:
    """design_problem.py:283-584 (prepare :481-492, optimize :495-584) for one on-axis source."""

    def __init__(self, lens_system, resolution, spp, learning_rate, iters, film_diagonal=35.0, init_camera_z=0.9, fov=135.0, source_distance=None):
        lens_system.initialize_geometry()
        self.lens_system = lens_system; self.resolution = resolution; self.spp = spp; self.lr = learning_rate; self.iters = iters
        self.film_diagonal = film_diagonal; self.init_camera_z = init_camera_z; self.fov = fov; self.source_distance = source_distance
        self.scene = self.params = self.optimizer = None

    def scene_dict(self):
        """_build_preliminary_scene (:364-401) + add_field_sources (:310-327)."""
        T = ScalarTransform4f
        d = {'type': 'scene',
             'integrator': {'type': 'ptracer', 'max_depth': self.lens_system.size() + 2, 'hide_emitters': False},
             'white-bsdf': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': (1, 1, 1)}},
             'black-bsdf': {'type': 'diffuse', 'reflectance': 0.0},
             'film_plane': {'type': 'rectangle', 'to_world': T.look_at(target=[0, 0, 1], origin=[0, 0, 0], up=[0, 1, 0]).scale([self.film_diagonal, self.film_diagonal, 1]),
                            'bsdf': {'type': 'ref', 'id': 'white-bsdf'}}}
        self.lens_system.add_to_scene(d)
        self.field_source = FieldSource(self.lens_system, self.resolution, self.init_camera_z, self.fov, self.source_distance)
        self.field_source.add_to_scene(d)
        return d

    def prepare(self):
        from .scene import load_dict
        from .integrators import traverse
        from .optimizers import Adam
        import torch
        self.scene = load_dict(self.scene_dict())
        self.params = traverse(self.scene)
        dev = self.params[next(iter(self.params._entries))].device
        self.optimizer = Adam(lr=self.lr)
        self.lens_system.save_init_state(self.params)
        self.lens_system.add_to_optimizer(self.optimizer, dev)
        return self

    def step(self, it):
        """One iteration of optimize() (:508-565): update the scene from the optimizer, render differentiably, back-propagate the spot loss,
        take the optimizer step.  Returns the loss (mm^2 x 10000, as the notebook scales it)."""
        from .integrators import render
        self.lens_system.update(self.params, self.optimizer)
        # multi-GPU: `shard = (rank, world)` -- every rank traces its part of the light-sample grid, the film and the gradient buffer are
        # all-reduced inside render / render_backward (BASELINE configs[4]: "8-GPU RCCL grad all-reduce"), so every rank takes the same step
        image = render(self.scene, self.params, seed=it, spp=2 * self.spp, spp_grad=self.spp, shard=getattr(self, "shard", None), group=getattr(self, "group", None))
        loss = 10000.0 * self.field_source.evaluate_rms_loss(image)
        loss.backward()
        self.optimizer.step()
        self.params.update()
        return float(loss.detach())

    def optimize(self, iters=None):
        losses = []
        for it in range(self.iters if iters is None else iters):
            losses.append(self.step(it))
            if it in (int(0.7 * self.iters), int(0.9 * self.iters)):
                self.optimizer.set_learning_rate(0.3 * self.optimizer.lr_default)
        return losses
