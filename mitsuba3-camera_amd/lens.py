"""Host side of the `realisticlens` sensor (src/sensors/realisticlens.cpp, fork-specific): lens prescriptions,
dispersive glass models, and the thick-lens autofocus the reference runs in its constructor.  The per-sample work
(film point -> rear element -> sequential refraction through the interfaces) runs on the device; this module only
produces the flat interface table handed over through the C ABI (mi355_sensor_desc.lens_data).

Interface table row (16 floats, metres): type (0 sphere, 1 plane, 2 aperture stop, 3 asphere), curvature radius (signed, sphere
only), element radius, z intercept on the optical axis, IOR on the film side ("left"), IOR on the world side ("right"); for an
asphere additionally kappa, the unitless curvature c and the seven scaled polynomial coefficients (include/mi355.h).
In RGB variants every glass is evaluated at the nominal wavelength 589.3 nm (realisticlens.cpp:169-179).
"""
import math

import numpy as np

SPHERE, PLANE, APERTURE, ASPHERE = 0, 1, 2, 3
NOMINAL_WAVELENGTH_UM = 0.5893


class DispersiveMaterial:
    """realisticlens.cpp:147-241: Cauchy n = A + B / lambda^2, or 3-term Sellmeier; wavelengths in micrometres."""

    def __init__(self, name, cauchy=None, sellmeier=None):
        self.name = name
        self.cauchy = None if cauchy is None else (np.float32(cauchy[0]), np.float32(cauchy[1]))
        self.sellmeier = None if sellmeier is None else [(np.float32(b), np.float32(c)) for b, c in sellmeier]

    def compute_ior(self, wavelength_um=NOMINAL_WAVELENGTH_UM):
        w = np.float32(wavelength_um)
        if self.cauchy is not None:
            return np.float32(self.cauchy[0] + self.cauchy[1] / (w * w))
        w2 = w * w
        n = np.float32(1.0)
        for b, c in self.sellmeier:
            n = np.float32(n + b * w2 / (w2 - c))
        return np.float32(np.sqrt(n))

    def abbe_number(self):
        return float((self.compute_ior(0.58756) - 1.0) / (self.compute_ior(0.4861) - self.compute_ior(0.6563)))


AIR = DispersiveMaterial("Air", cauchy=(1.000277, 0.0))


class Interface:
    def __init__(self, kind, element_radius, z, left, right, curvature_radius=0.0, kappa=0.0, ai=None):
        self.kind = kind
        self.curvature_radius = float(curvature_radius)
        self.element_radius = float(element_radius)
        self.z = float(z)
        self.left, self.right = left, right      # materials on the film side / world side
        # AsphericalLens ctor (realisticlens.cpp:559-572): curvature radius and a_i arrive in millimetres
        self.kappa = float(kappa)
        self.c = 0.0
        self.ai = [0.0] * 7
        if kind == ASPHERE:
            r_scale = 1000.0 * self.element_radius
            self.c = r_scale / self.curvature_radius
            self.ai = [r_scale ** (2 * i + 3) * float(a) for i, a in enumerate(list(ai) + [0.0] * (7 - len(ai)))]

    def row(self):
        return [float(self.kind), self.curvature_radius, self.element_radius, self.z,
                float(self.left.compute_ior()), float(self.right.compute_ior()), self.kappa, self.c] + [float(a) for a in self.ai] + [0.0]

    # ---- asphere geometry in float64 (realisticlens.cpp:574-731), used by the constructor-time autofocus
    def asph_eval(self, r2):
        r2_ = r2 / self.element_radius ** 2
        z_ = 0.0
        for a in reversed(self.ai):
            z_ = z_ * r2_ + a
        z_ *= r2_ * r2_
        z_ += self.c * r2_ / (1 + math.sqrt(1 - (1 + self.kappa) * self.c ** 2 * r2_))
        return -z_ * self.element_radius + self.z

    def asph_normal(self, p):
        rx, ry = p[0] / self.element_radius, p[1] / self.element_radius
        r2_ = rx * rx + ry * ry
        if r2_ >= 4 * 5.96e-8:
            s = 1 / math.sqrt(r2_); rx *= s; ry *= s
        else:
            rx = ry = 0.0
        r_ = math.sqrt(r2_)
        z_ = 0.0
        for i in range(6, -1, -1):
            z_ = z_ * r2_ + (2.0 * i + 4.0) * self.ai[i]
        z_ *= r2_ * r_
        cr = self.c * r_
        z_ += cr / math.sqrt(1 - (1 + self.kappa) * cr * cr)
        n = np.array([-z_ * rx, -z_ * ry, -1.0])
        return n / np.linalg.norm(n)

    def asph_intersect(self, o, d):
        R, c, K, z0 = self.element_radius, self.c, self.kappa, self.z
        on = (o - np.array([0.0, 0.0, z0])) / R
        A = c * (1 + K * d[2] ** 2); B = 2 * (c * (float(on @ d) + K * on[2] * d[2]) - d[2]); C = c * (float(on @ on) + K * on[2] ** 2) - 2 * on[2]
        t = None
        disc = B * B - 4 * A * C
        if A == 0 and B != 0:
            roots = (-C / B, -C / B)
        elif A != 0 and disc >= 0:
            q = -0.5 * (B + math.copysign(math.sqrt(disc), B)); r0, r1 = q / A, C / q; roots = (min(r0, r1), max(r0, r1))
        else:
            roots = None
        if roots is not None:
            t0, t1 = roots[0] * R, roots[1] * R
            lim = R / (1 + K) if K > -1 else math.inf
            v0 = 0 < c * (o[2] + t0 * d[2] - z0) <= lim; v1 = 0 < c * (o[2] + t1 * d[2] - z0) <= lim
            if v0 or v1:
                t = (t0 if t0 > 0 else t1) if (v0 and v1) else (t0 if v0 else t1)
        if t is None:
            t = (z0 - o[2]) / d[2]
        p = o + t * d; err = math.inf
        for _ in range(10):
            z_asph = self.asph_eval(p[0] ** 2 + p[1] ** 2)
            err = abs(p[2] - z_asph)
            pn = self.asph_normal(p)
            t = float(pn @ (np.array([p[0], p[1], z_asph]) - o)) / float(pn @ d)
            p = o + t * d
            if not err > 5.96e-7:
                break
        if not err < 5.96e-7 or not math.isfinite(t):
            return None
        return p, self.asph_normal(p)


# ---- scalar double-precision trace used only by the constructor-time autofocus ---------------------------------
def _fresnel(cos_theta_i, eta):
    outside = cos_theta_i >= 0
    eta_it, eta_ti = (eta, 1 / eta) if outside else (1 / eta, eta)
    ctt2 = 1 - (1 - cos_theta_i * cos_theta_i) * eta_ti * eta_ti
    cia, cta = abs(cos_theta_i), math.sqrt(max(ctt2, 0.0))
    if eta == 1:
        r = 0.0
    elif cia == 0:
        r = 1.0
    else:
        a_s = (cia - eta_it * cta) / (cia + eta_it * cta); a_p = (cta - eta_it * cia) / (cta + eta_it * cia)
        r = 0.5 * (a_s * a_s + a_p * a_p)
    return r, (-cta if cos_theta_i >= 0 else cta), eta_ti


def _interact(itf, o, d):
    """LensInterface::compute_interaction (realisticlens.cpp:269-297) in float64.  Returns (o, d) or None."""
    if itf.kind == ASPHERE:
        r = itf.asph_intersect(o, d)
        if r is None:
            return None
        p, n = r
    elif itf.kind == SPHERE:
        R = itf.curvature_radius
        c = np.array([0.0, 0.0, itf.z + R])
        pc = c - o
        proj = float(pc @ d)
        perp = np.linalg.norm(pc - proj * d)
        disc = R * R - perp * perp
        if disc < 0:
            return None
        sq = math.sqrt(disc); near, far = proj - sq, proj + sq
        if far < 0:
            return None
        convex = R > 0
        t = (near if near >= 0 else math.inf) if (convex != (d[2] < 0)) else far
        if not math.isfinite(t):
            return None
        p = o + t * d
        n = (p - c) / np.linalg.norm(p - c)
        if not convex:
            n = -n
    else:
        n = np.array([0.0, 0.0, -1.0])
        nd = float(n @ d)
        if abs(nd) < 5.96e-8:
            return None
        t = (n[2] * itf.z - float(n @ o)) / nd
        if t < 0:
            return None
        p = o + t * d
    if p[0] * p[0] + p[1] * p[1] >= itf.element_radius ** 2:
        return None
    if itf.kind == APERTURE:
        return p, d
    ext_ior, int_ior = float(itf.left.compute_ior()), float(itf.right.compute_ior())
    cos_i = float(-d @ n)
    r, cos_t, eta_ti = _fresnel(cos_i, int_ior / ext_ior)
    if r > 1 - 5.96e-8:
        return None
    wi = -d
    d_out = n * (float(wi @ n) * eta_ti + cos_t) - wi * eta_ti      # refract(), fresnel.h:309-313
    return p, d_out


def trace(interfaces, o, d, from_film=True):
    o = np.asarray(o, dtype=np.float64); d = np.asarray(d, dtype=np.float64)
    for itf in (interfaces if from_film else reversed(interfaces)):
        r = _interact(itf, o, d)
        if r is None:
            return None
        o, d = r
    return o, d


def _cardinal(start_o, end_o, end_d):
    """compute_cardinal_points, realisticlens.cpp:2107-2113"""
    t_focus = -end_o[0] / end_d[0]
    z_f = end_o[2] + t_focus * end_d[2]
    t_plane = (start_o[0] - end_o[0]) / end_d[0]
    z_p = end_o[2] + t_plane * end_d[2]
    return z_p, z_f


def thick_lens_approximation(interfaces, film_half_width):
    """compute_thick_lens_approximation, realisticlens.cpp:2115-2159"""
    r = 0.001 * film_half_width
    o = np.array([r, 0.0, interfaces[-1].z + 1.0])
    res = trace(interfaces, o, [0, 0, -1.0], from_film=False)
    if res is None:
        raise RuntimeError("compute_thick_lens_approximation: world ray was not transmitted through lens!")
    obj_plane, obj_focus = _cardinal(o, *res)
    o2 = np.array([r, 0.0, interfaces[0].z - 1.0])
    res2 = trace(interfaces, o2, [0, 0, 1.0], from_film=True)
    if res2 is None:
        raise RuntimeError("compute_thick_lens_approximation: film ray was not transmitted through lens!")
    img_plane, img_focus = _cardinal(o2, *res2)
    return obj_plane, obj_plane - obj_focus, img_plane, img_focus - img_plane      # p_img(back), f_img, p_obj(front), f_obj


def focus_thick_lens(interfaces, film_half_width, focus_distance):
    """focus_thick_lens, realisticlens.cpp:2161-2171"""
    p_img, f_img, p_obj, f_obj = thick_lens_approximation(interfaces, film_half_width)
    tmp = max(focus_distance, 4.01 * f_img) - p_obj
    return 0.5 * (p_img - tmp + math.sqrt((p_img + tmp) ** 2 - 4 * f_img * (p_img + tmp)))


# ---- prescriptions ------------------------------------------------------------------------------------------------
def build_singlet(object_distance, curvature_radius, lens_radius):
    """realisticlens.cpp:936-955"""
    distance = max(object_distance, 4.001 * curvature_radius)
    z = 0.5 * distance * (1 - math.sqrt(1 - 4 * curvature_radius / distance))
    thickness = 2 * curvature_radius * (1 - math.sqrt(1 - (lens_radius / curvature_radius) ** 2))
    glass = DispersiveMaterial("NBK7", cauchy=(1.5046, 5 * 0.00420))
    return [Interface(SPHERE, lens_radius, z, AIR, glass, curvature_radius),
            Interface(SPHERE, lens_radius, z + thickness, glass, AIR, -curvature_radius)]


def build_doublet(object_distance, R, lens_radius):
    """realisticlens.cpp:957-975"""
    f = 2 * R
    distance = max(object_distance, 4.001 * f)
    z = 0.5 * distance * (1 - math.sqrt(1 - 4 * f / distance))
    thickness = 2 * R * (1 - math.sqrt(1 - (lens_radius / R) ** 2))
    g1 = DispersiveMaterial("NBK7", cauchy=(1.5046, 0.00420)); g2 = DispersiveMaterial("mod-NBK7", cauchy=(1.5046, 0.00860948454))
    return [Interface(PLANE, lens_radius, z, AIR, g2), Interface(SPHERE, lens_radius, z + thickness, g2, g1, R),
            Interface(SPHERE, lens_radius, z + 2 * thickness, g1, AIR, -R), Interface(APERTURE, lens_radius, z + 3 * thickness, AIR, AIR)]


def build_from_data(kinds, thicknesses, curv_radii, elem_radii, mats, stop_ratio=1.0, scale=1.0, kappas=None, ai_list=None):
    """build_lens_from_data, realisticlens.cpp:1568-1652: tables are listed from the world side, in millimetres."""
    out = []
    z_pos = 0.0
    for i in range(len(thicknesses) - 1, -1, -1):
        z_pos += thicknesses[i]
        er, z = 0.001 * scale * elem_radii[i], 0.001 * scale * z_pos
        if kinds[i] == APERTURE:
            out.append(Interface(APERTURE, er * stop_ratio, z, mats[i + 1], mats[i + 1]))
        elif kinds[i] == PLANE:
            out.append(Interface(PLANE, er, z, mats[i + 1], mats[i]))
        elif kinds[i] == ASPHERE:                                   # :1607-1626: R and a_i rescaled with the design scale
            ai = [a / scale ** (2 * j + 3) for j, a in enumerate(ai_list[i])]
            out.append(Interface(ASPHERE, er, z, mats[i + 1], mats[i], curvature_radius=scale * curv_radii[i], kappa=kappas[i], ai=ai))
        else:
            out.append(Interface(SPHERE, er, z, mats[i + 1], mats[i], -0.001 * scale * curv_radii[i]))
    return out


def build_exp1_doublet(cauchy_a, cauchy_b, stop_ratio=1.0):
    """realisticlens.cpp:1283-1320 (the doublet of notebooks/doublet_full_optim)"""
    glass_a = DispersiveMaterial("glass_A", cauchy=(1.4560798389592584, 0.021086562958141358))
    glass_b = DispersiveMaterial("glass_B", cauchy=(cauchy_a, cauchy_b))
    return build_from_data([APERTURE, SPHERE, SPHERE, SPHERE], [0.0, 3.0, 2.25291824, 46.74708176],
                           [1000.0, 24.0, -24.0, -168.01068267], [8.0, 8.0, 8.0, 8.0], [AIR, AIR, glass_a, glass_b, AIR], stop_ratio)


S, A = SPHERE, APERTURE


def _cauchy(name, a, b):
    return DispersiveMaterial(name, cauchy=(a, b))


def build_tessar(stop_ratio=1.0):
    """realisticlens.cpp:977-1029 (prescription: henryquach.org/tessar.html); Sellmeier glasses"""
    nlak9 = DispersiveMaterial("NLAK9", sellmeier=[(1.462319050, 0.007242702), (0.344399589, 0.0243353131), (1.155083720, 85.46868680)])
    k10 = DispersiveMaterial("K10", sellmeier=[(1.156870820, 0.008094243), (0.064262544, 0.0386051284), (0.872376139, 104.74773000)])
    f2 = DispersiveMaterial("F2", sellmeier=[(1.397570370, 0.009959061), (0.159201403, 0.0546931752), (1.268654300, 119.24834600)])
    return build_from_data([S, S, S, S, A, S, S, S], [3.567, 2.286, 2.290, 2.289, 1.999, 2.286, 9.941, 86.917],
                           [35.034, -296.111, -63.028, 31.297, 1e8, -86.620, 45.344, -43.567], [9.0, 9.0, 6.0, 6.0, 4.65, 8.0, 8.0, 8.0],
                           [AIR, nlak9, AIR, f2, AIR, AIR, k10, nlak9, AIR], stop_ratio)


def build_helios(stop_ratio=1.0):
    """realisticlens.cpp:1032-1086 (Helios 44M-4, photonstophotos.net OpticalBench GOI/ST01FB06)"""
    bf16, tk14, lf7 = _cauchy("BF16", 1.648275034, 0.007806736), _cauchy("TK14", 1.597547619, 0.005351918), _cauchy("LF7", 1.551328271, 0.008025103)
    return build_from_data([S, S, S, S, S, A, S, S, S, S, S], [4.81, 2.26, 9.07, 1.31, 4.7, 4.63, 1.32, 6.25, 0.5, 4.94, 38.08],
                           [38.07, 136.365, 25.33, -124.225, 15.995, 1e8, -16.62, 66.085, -22.21, 191.54, -52.725],
                           [14.75, 14.75, 13.2, 11.6, 9.75, 9.575, 9.35, 10.2, 10.6, 12.35, 12.35],
                           [AIR, tk14, AIR, tk14, lf7, AIR, AIR, lf7, tk14, AIR, bf16, AIR], stop_ratio)


def build_jupiter(stop_ratio=1.0):
    """realisticlens.cpp:1089-1149 (Jupiter-9, photonstophotos.net OpticalBench GOI/ST01FB43)"""
    tk16, bf13, k1 = _cauchy("TK16", 1.596466676, 0.00558386), _cauchy("BF13", 1.618336084, 0.007302944), _cauchy("K1", 1.486688667, 0.00398663)
    tf2, of1, bf7 = _cauchy("TF2", 1.637217608, 0.012112489), _cauchy("OF1", 1.513488027, 0.005500433), _cauchy("BF7", 1.562693323, 0.005811246)
    return build_from_data([S, S, S, S, S, S, A, S, S, S, S], [5.6, 0.4, 5.8, 7.5, 1.8, 10.0, 3.8, 2.8, 10.6, 2.9, 40.53],
                           [46.45, 268.5, 25.94, 52.0, -264.2, 16.444, 1e8, -1043.65, 44.51, -15.031, -95.06],
                           [22.0, 22.0, 19.015, 19.015, 19.015, 12.68, 12.275, 24.57 * 0.5, 13.5, 13.5, 15.15],
                           [AIR, tk16, AIR, bf13, k1, tf2, AIR, AIR, of1, bf13, bf7, AIR], stop_ratio)


def build_fisheye(stop_ratio=1.0):
    """realisticlens.cpp:1152-1220 (JP1988-017421 example 3, photonstophotos.net OpticalBench)"""
    ga, gb, gc = _cauchy("glass_A", 1.5881276381075704, 0.005202992085188941), _cauchy("glass_B", 1.793242496642434, 0.018550536235572006), \
        _cauchy("glass_C", 1.4770186893501427, 0.003636419065560783)
    gd, ge = _cauchy("glass_D", 1.6021851259042148, 0.005144827846028017), _cauchy("glass_E", 1.4983808648479255, 0.004423976662977713)
    return build_from_data([S] * 8 + [A] + [S] * 7, [2.5, 11.83, 2.5, 7.54, 5.34, 1.85, 6.71, 3.84, 3.53, 2.72, 0.15, 3.99, 5.14, 0.15, 4.84, 39.67],
                           [78.06, 15.9, 22.22, 13.27, 127.88, 22.35, 32.04, -190.22, -1000000.0, -289.77, -29.1, -100.42, 29.39, -25.73, 43.88, -43.88],
                           [31.725, 15.9, 13.89, 10.69, 9.955, 7.61, 6.73, 6.73, 6.659, 7.21, 7.21, 9.52, 9.52, 9.52, 11.71, 11.71],
                           [AIR, ga, AIR, gd, AIR, ga, AIR, gb, AIR, AIR, ge, AIR, gb, gc, AIR, gc, AIR], stop_ratio)


def build_double_gauss(stop_ratio=1.0):
    """realisticlens.cpp:1223-1280 (double Gauss from Smith, Modern Optical Engineering)"""
    ga, gb, gc = _cauchy("glass_A", 1.64855004723031, 0.00744902140861971), _cauchy("glass_B", 1.66398266226799, 0.0121606281020403), \
        _cauchy("glass_C", 1.57907201321296, 0.00830957940819446)
    gd, ge = _cauchy("glass_D", 1.64068415393588, 0.00601335161083744), _cauchy("glass_E", 1.69447574875623, 0.00782209786331075)
    return build_from_data([S, S, S, S, S, A, S, S, S, S, S], [7.52, 0.24, 8.05, 6.55, 11.41, 9.0, 2.36, 12.13, 0.38, 6.44, 72.228],
                           [58.95, 169.66, 38.55, 81.54, 25.5, 10000000.0, -28.99, 81.54, -40.77, 874.13, -79.46],
                           [25.2, 25.2, 23.0, 23.0, 18.0, 17.1, 17.0, 20.0, 20.0, 20.0, 20.0],
                           [AIR, ga, AIR, ga, gb, AIR, AIR, gc, gd, AIR, ge, AIR], stop_ratio)


def build_exp2_nikon(mode):
    """realisticlens.cpp:1452-1566: 22-surface Nikon design of notebooks' experiment 2 (mode 0 = as published, 1 = aspheres
    zeroed, 2 = re-optimised aspheres); design scale 35 / 1.57, aperture stopped down to 0.5."""
    g = [_cauchy("glass_%s" % n, a, b) for n, a, b in (
        ("A", 1.5046464980225065, 0.004220601200948007), ("B", 1.8645930283464647, 0.013233579146425922), ("C", 1.5725988459469284, 0.007952986594191858),
        ("D", 1.6639243971469755, 0.012180862092226684), ("E", 1.8718384392978837, 0.013530356122431513), ("F", 1.5797734345478602, 0.0045237938738434446),
        ("G", 1.689190440676433, 0.010873011246216379), ("H", 1.5673889795706786, 0.008789871710972213), ("I", 1.6738434507974307, 0.006826218099482183),
        ("J", 1.4770641526601842, 0.00362410356265244))]
    gA, gB, gC, gD, gE, gF, gG, gH, gI, gJ = g
    kinds = [S] * 11 + [A] + [S] * 4 + [ASPHERE, ASPHERE] + [S, S, PLANE, PLANE]
    curv = [5.2674, 0.9607, 1.442, 10.2799, 1.2154, -1.0987, 2.9183, -1.6693, 1.6429, -1.4116, -2.5725, 1e8, -0.973, -24.0801, 2.3756, -1.3055,
            -7.3169, -2.2, -1.5449, -7.2572, 1e8, 1e8]
    thick = [0.102, 0.309, 0.246, 0.083, 0.411, 0.088, 0.258, 0.009, 0.379, 0.069, 0.118, 0.604, 0.051, 0.009, 0.282, 0.239, 0.122, 0.154, 0.083, 0.75,
             0.074, 0.0425]
    radii = [0.847, 0.696, 0.661, 0.625, 0.546, 0.524, 0.586, 0.601, 0.624, 0.613, 0.607, 0.555, 0.476, 0.49, 0.543, 0.569, 0.604, 0.627, 0.662, 0.712,
             1.2, 1.2]
    ai = [[0.0] * 7 for _ in range(22)]
    if mode == 0:
        ai[16] = [-0.24, -0.4268, 0.0, 0.0, 0.0, 0.0, 0.0]; ai[17] = [-0.05053, -0.3491, 0.1459, 0.07718, 0.0, 0.0, 0.0]
    elif mode == 2:
        ai[16] = [-0.09490284323692322, -0.10803833603858948, -0.08069773763418198, -0.01646527275443077, -0.007415791507810354,
                  -0.503877580165863, -3.2629644870758057]
        ai[17] = [0.07706142961978912, 0.06080370396375656, -0.015299857594072819, -0.16691075265407562, -0.38500601053237915,
                  -0.5226516723632812, 0.1124117523431778]
    mats = [AIR, gA, AIR, gB, gC, AIR, gD, gE, AIR, gF, gG, AIR, AIR, gH, AIR, gF, AIR, gI, AIR, gJ, AIR, gA, AIR]
    return build_from_data(kinds, thick, curv, radii, mats, stop_ratio=0.5, scale=35.0 / 1.57, kappas=[0.0] * 22, ai_list=ai)


def build_asph(stop_ratio=1.0):
    """realisticlens.cpp:1373-1449: 13-asphere phone-camera lens (no fine focus, :876-878)."""
    gA, gB, gC = _cauchy("glass_A", 1.52923858553436, 0.00509603519451627), _cauchy("glass_B", 1.59833561885235, 0.0143836590443554), \
        _cauchy("glass_C", 1.5205859418518, 0.00497611850070528)
    nbk7 = _cauchy("NBK7", 1.50478491769513, 0.00421697320906858)
    Q = ASPHERE
    kinds = [Q, A, Q, Q, Q, Q, Q, Q, Q, Q, Q, Q, Q, PLANE, PLANE]
    curv = [1.952, 1e8, 2.978, 2.003, 2.167, 4.143, -3.947, -0.969, -2.089, 3.867, -0.847, -16.711, 0.785, 1e8, 1e8]
    thick = [0.118, 0.28, 0.069, 0.28, 0.205, 0.551, 0.211, 0.421, 0.03, 1.105, 0.03, 0.44, 0.7, 0.3, 0.235]
    radii = [0.72391, 0.70781, 0.79221, 0.79661, 0.91103, 0.94416, 1.05504, 1.07363, 1.26654, 1.33913, 1.56417, 1.81331, 2.37438, 2.72353, 2.72353]
    kappas = [-2.1201, 0.0, -20.0, -12.382, -3.006, 1.0, 0.0, -1.4213, -3.0569, -6.3368, -4.2091, -20.0, -5.1631, 0.0, 0.0]
    ai = [[3.5988e-03, 3.7387e-01, -1.3929e+00, 1.4094e+00, 2.0282e+00, -3.6199e+00, 0.0], [0.0] * 7,
          [-1.7492e-01, 9.8443e-02, 6.6244e-01, -1.7257e+00, 6.6955e-01, -3.4243e-01, 0.0],
          [-2.6551e-01, 5.7642e-01, -1.1253e+00, 1.1555e+00, 1.4439e-02, -1.9141e+00, 0.0],
          [-2.5306e-01, 3.1573e-01, -2.1258e-01, 9.4103e-02, -2.7001e-01, 1.1110e-01, 0.0],
          [-1.4680e-01, -1.9199e-02, 1.1489e-01, -1.6910e-01, -1.1287e-01, 1.2615e-01, 0.0],
          [-4.4328e-02, -1.6599e-01, 4.8223e-02, 9.4653e-03, 1.2137e-02, -8.1558e-03, 0.0],
          [3.3847e-01, -8.7891e-01, 1.1445e+00, -9.1578e-01, 5.9482e-01, -2.0143e-01, 0.0],
          [1.4102e-01, -2.7755e-01, 3.2138e-01, -2.4968e-01, 1.2901e-01, -2.7976e-02, 0.0],
          [-1.6984e-01, 1.4944e-01, -9.9124e-02, 1.3353e-02, 1.6198e-03, 1.2524e-03, 0.0],
          [-6.7672e-02, -3.0645e-02, 1.0444e-01, -8.3869e-02, 2.6536e-02, -2.7869e-03, 0.0],
          [-1.3442e-01, 3.8933e-02, -6.2383e-03, 7.5356e-04, 2.2005e-04, -5.2957e-05, 0.0],
          [-7.5325e-02, 2.4095e-02, -5.8920e-03, 1.0269e-03, -1.1564e-04, 5.9417e-06, 0.0], [0.0] * 7, [0.0] * 7]
    mats = [AIR, gA, gA, AIR, gB, AIR, gA, AIR, gB, AIR, gC, AIR, gC, AIR, nbk7, AIR]
    return build_from_data(kinds, thick, curv, radii, mats, stop_ratio=stop_ratio, kappas=kappas, ai_list=ai)


def build_hypercentric(ap_radius=1.0, focal_lengths=2.0):
    """realisticlens.cpp:1322-1370: one thick biconvex element behind a far aperture (hypercentric / telecentric optics).
    Returns (interfaces, film_z_position): the film does not sit at z = 0 of the lens frame (:1366, :2263)."""
    sf6 = _cauchy("SF6", 1.757435971209294, 0.016580283248837778)
    target_img_size, lens_radius, focal_length, track_length, R, lens_thickness = 5.0, 120.0, 210.0, 450.0, 310.505660861, 50.0
    z_ap_to_lens = focal_length * focal_lengths
    z_film_to_ap = z_ap_to_lens * (target_img_size - ap_radius) / (lens_radius + ap_radius)
    curv, thick, radii = [R, -R, math.inf], [lens_thickness, z_ap_to_lens - lens_thickness * 0.5, z_film_to_ap], [lens_radius, lens_radius, ap_radius]
    mats = [AIR, sf6, AIR, AIR]
    out, z_pos = [], 0.0
    for i in (2, 1, 0):
        z_pos += thick[i]
        if i == 2:
            out.append(Interface(APERTURE, 0.001 * radii[i], 0.001 * z_pos, AIR, AIR))
        else:
            out.append(Interface(SPHERE, 0.001 * radii[i], 0.001 * z_pos, mats[i + 1], mats[i], -0.001 * curv[i]))
    return out, 0.001 * (track_length - sum(thick))


TABLE_DESIGNS = {"tessar": build_tessar, "helios": build_helios, "jupiter": build_jupiter, "fisheye": build_fisheye, "gauss": build_double_gauss}
EXP1 = {"exp1a": (1.5689525390422485, 0.0263051608728981), "exp1b": (1.507009332993809, 0.04781645616233479),
        "exp1c": (1.5030211256389316, 0.04920145883751409)}     # realisticlens.cpp:883-888
EXP2 = {"exp2a": 0, "exp2b": 1, "exp2c": 2}
UNSUPPORTED_DESIGNS = ()


def build_lens(props, film_half_width):
    """RealisticLensCamera ctor, realisticlens.cpp:832-907: prescription + fine focus.  Returns the interface list."""
    design = str(props.get("lens_design", "singlet"))
    object_distance = float(props.get("object_distance", 6.0))
    focal_length = float(props.get("lens_focal_length", 0.05))
    lens_diameter = float(props.get("lens_diameter", 0.01))
    stop_ratio = float(props.get("stop_ratio", 1.0))
    if design == "custom":          # extension: an explicit interface list (film -> world), e.g. to exercise aspheres on their own
        import copy
        itf = [copy.copy(i) for i in props["lens_interfaces"]]
        delta = focus_thick_lens(itf, film_half_width, object_distance) if props.get("fine_focus", True) else float("nan")
        if not math.isnan(delta):
            for i in itf:
                i.z -= delta
        return itf, delta
    if design in UNSUPPORTED_DESIGNS:
        raise NotImplementedError('lens_design "%s" is not carried by this build (carried: singlet, doublet, tessar, helios, jupiter, '
                                  'fisheye, gauss, exp1a/b/c, exp2a/b/c)' % design)
    if design == "doublet":
        itf = build_doublet(object_distance, focal_length / 2, lens_diameter / 2)
    elif design in TABLE_DESIGNS:
        itf = TABLE_DESIGNS[design](stop_ratio=stop_ratio)
    elif design in EXP2:
        itf = build_exp2_nikon(EXP2[design])
    elif design == "asph":                  # no fine focus (:876-878)
        return build_asph(stop_ratio=stop_ratio), float("nan")
    elif design == "hypercentric":          # no fine focus either (:866-875); the caller shifts the camera frame by film_z_position
        itf, film_z = build_hypercentric(float(props.get("ap_size", 1.0)), float(props.get("ap_offset", 2.0)))
        for i in itf:
            i.film_z_position = film_z
        return itf, float("nan")
    elif design in EXP1:
        itf = build_exp1_doublet(*EXP1[design], stop_ratio=stop_ratio)
    else:
        itf = build_singlet(object_distance, focal_length, lens_diameter / 2)
    delta = focus_thick_lens(itf, film_half_width, object_distance)
    if not math.isnan(delta):
        for i in itf:
            i.z -= delta
    return itf, delta


def lens_table(interfaces):
    return np.ascontiguousarray(np.array([i.row() for i in interfaces], dtype=np.float32))
