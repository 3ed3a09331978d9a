"""Host-side scene description: the subset of Mitsuba's dict/plugin front-end that the hot path
needs, flattened into the POD records of include/mi355.h.

Mirrors (names, argument meaning, error behaviour):
  mi.load_dict                         src/python/python/... (dict loader), src/core/xml.cpp:1147-1163 (rgb expansion)
  mi.cornell_box()                     src/python/python/util.py:565-699
  parse_fov                            src/render/sensor.cpp:149-199
  Cube                                 src/shapes/cube.cpp:105-163 (24 vertices / 12 triangles, world space)
  Rectangle / PerspectiveCamera / HDRFilm / GaussianFilter / IndependentSampler property names
Anything outside the path's scope raises, like PluginManager does for a missing plugin
(src/core/plugin.cpp:144-158).
"""
import math
import os
import ctypes as C
import numpy as np

from . import _abi as abi
from .transforms import Transform4f, ScalarTransform4f


class PluginError(RuntimeError):
    pass


# ------------------------------------------------------------------------------------------------
# parse_fov: src/render/sensor.cpp:149-199
# ------------------------------------------------------------------------------------------------
def parse_fov(props, aspect):
    if "fov" in props and "focal_length" in props:
        raise PluginError("Please specify either a focal length ('focal_length') or a field of view ('fov')!")
    if "fov" in props:
        fov = float(props["fov"])
        fov_axis = str(props.get("fov_axis", "x")).lower()
        if fov_axis == "smaller":
            fov_axis = "y" if aspect > 1 else "x"
        elif fov_axis == "larger":
            fov_axis = "x" if aspect > 1 else "y"
    else:
        f = str(props.get("focal_length", "50mm"))
        if f.endswith("mm"):
            f = f[:-2]
        try:
            value = float(f)
        except ValueError:
            raise PluginError("Could not parse the focal length (must be of the form <x>mm, where <x> is a positive integer)!")
        fov = 2.0 * math.degrees(math.atan(math.sqrt(36.0 * 36 + 24 * 24) / (2.0 * value)))
        fov_axis = "diagonal"
    if fov_axis == "x":
        result = fov
    elif fov_axis == "y":
        result = math.degrees(2.0 * math.atan(math.tan(0.5 * math.radians(fov)) * aspect))
    elif fov_axis == "diagonal":
        diagonal = 2.0 * math.tan(0.5 * math.radians(fov))
        width = diagonal / math.sqrt(1.0 + 1.0 / (aspect * aspect))
        result = math.degrees(2.0 * math.atan(width * 0.5))
    else:
        raise PluginError("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!")
    if result <= 0.0 or result >= 180.0:
        raise PluginError("The horizontal field of view must be in the range [0, 180]!")
    return result


# ------------------------------------------------------------------------------------------------
# host objects
# ------------------------------------------------------------------------------------------------
class Texture:
    """`bitmap` texture from a tensor (raw=True semantics, bitmap.cpp:183-190)."""

    def __init__(self, data, filter_type="bilinear", wrap_mode="repeat", to_uv=None, id=None):
        data = np.ascontiguousarray(np.asarray(data, dtype=np.float32))
        if data.ndim == 2:
            data = data[:, :, None]
        if data.ndim != 3 or data.shape[2] not in (1, 3):
            raise PluginError("Unsupported tensor channel count: %s (expected 1 or 3)" % (data.shape,))
        if data.shape[0] < 2 or data.shape[1] < 2:
            raise PluginError("bitmap texture must be at least 2x2 pixels in size!")
        try:
            self.filter = {"nearest": abi.FILTER_NEAREST, "bilinear": abi.FILTER_BILINEAR}[filter_type]
        except KeyError:
            raise PluginError('Invalid filter type "%s", must be one of: "nearest", or "bilinear"!' % filter_type)
        try:
            self.wrap = {"repeat": abi.WRAP_REPEAT, "mirror": abi.WRAP_MIRROR, "clamp": abi.WRAP_CLAMP}[wrap_mode]
        except KeyError:
            raise PluginError('Invalid wrap mode "%s", must be one of: "repeat", "mirror", or "clamp"!' % wrap_mode)
        self.data = data
        self.to_uv = np.array([1, 0, 0, 0, 1, 0], dtype=np.float32) if to_uv is None else np.asarray(to_uv, dtype=np.float32).reshape(6)
        self.id = id


_IOR = {"vacuum": 1.0, "helium": 1.000036, "hydrogen": 1.000132, "air": 1.000277, "carbon dioxide": 1.00045, "water": 1.3330,
        "acetone": 1.36, "ethanol": 1.361, "carbon tetrachloride": 1.461, "glycerol": 1.4729, "benzene": 1.501,
        "silicone oil": 1.52045, "bromine": 1.661, "water ice": 1.31, "fused quartz": 1.458, "pyrex": 1.470,
        "acrylic glass": 1.49, "polypropylene": 1.49, "bk7": 1.5046, "sodium chloride": 1.544, "amber": 1.55, "pet": 1.5750,
        "diamond": 2.419}      # include/mitsuba/render/ior.h:21-47


def lookup_ior(props, name, default):
    v = props.get(name, default)
    if isinstance(v, str):
        if v.lower() not in _IOR:
            raise PluginError('Unable to find an IOR value for "%s"!' % v.lower())
        return float(_IOR[v.lower()])
    return float(v)


class BSDF:
    def __init__(self, id, reflectance=(0.5, 0.5, 0.5), texture=None, type=abi.BSDF_DIFFUSE):
        self.id = id
        self.type = type
        self.reflectance = np.asarray(reflectance, dtype=np.float32).reshape(3).copy()
        self.texture = texture      # Texture or None
        # roughconductor (roughconductor.cpp:163-212) / dielectric (dielectric.cpp:208-236)
        self.alpha_u = self.alpha_v = 0.1
        self.distribution = 0       # 0 beckmann, 1 ggx
        self.sample_visible = True
        self.eta = np.zeros(3, dtype=np.float32); self.k = np.ones(3, dtype=np.float32)
        self.int_ior, self.ext_ior = 1.5046, 1.000277
        self.transmittance = np.ones(3, dtype=np.float32)
        self.dispersive = None      # `dispersive` plugin: its four parameters (int_ior_d, int_V_d, ext_ior_d, ext_V_d)

    def flags_smooth(self):
        return self.type not in (abi.BSDF_DIELECTRIC, abi.BSDF_CONDUCTOR)


class Emitter:
    def __init__(self, radiance):
        self.type = abi.EMITTER_AREA
        self.radiance = np.asarray(radiance, dtype=np.float32).reshape(3).copy()
        self.shape = None


class Shape:
    def __init__(self, id, kind, to_world, bsdf, emitter=None, flip_normals=False):
        self.id = id
        self.type = kind
        self.to_world = to_world
        self.bsdf = bsdf
        self.emitter = emitter
        self.flip_normals = flip_normals
        self._vp = self._vn = None
        self._normals_stale = False      # vertex_positions were updated: the engine recomputed the normals on the device; the host copy catches up lazily
        self._positions_device = None    # ... and the positions themselves still live in a device tensor (SceneParameters._push)
        self.vertex_texcoords = self.faces = None

    # The host copies of an updated mesh are brought up to date on first use (positions from the device tensor, normals recomputed as
    # Mesh::parameters_changed does, mesh.cpp:138-139): an optimisation loop that never looks at them never pays for the download.
    @property
    def vertex_positions(self):
        if self._positions_device is not None:
            self._vp = np.ascontiguousarray(self._positions_device.detach().cpu().numpy().reshape(-1, 3).astype(np.float32)); self._positions_device = None
        return self._vp

    @vertex_positions.setter
    def vertex_positions(self, v):
        self._vp = v; self._positions_device = None

    @property
    def vertex_normals(self):
        if self._normals_stale and self._vn is not None:
            self._vn = compute_vertex_normals(self.vertex_positions, self.faces)
        self._normals_stale = False
        return self._vn

    @vertex_normals.setter
    def vertex_normals(self, v):
        self._vn = v; self._normals_stale = False

    def sync_host(self):
        return self.vertex_positions, self.vertex_normals

    def primitive_count(self):
        return 1 if self.type != abi.SHAPE_MESH else int(self.faces.shape[0])

    def surface_area(self):
        if self.type == abi.SHAPE_RECTANGLE:      # rectangle.cpp:158-160
            s = self.to_world.transform_vector([2, 0, 0]); t = self.to_world.transform_vector([0, 2, 0])
            return float(np.linalg.norm(np.cross(s, t)))
        if self.type == abi.SHAPE_SPHERE:         # sphere.cpp:197-199
            return float(4 * math.pi * np.linalg.norm(self.to_world.transform_vector([1, 0, 0])) ** 2)
        if self.type == abi.SHAPE_DISK:           # disk.cpp:163-167: pi * |dp_du| * height of the ellipse
            du = self.to_world.transform_vector([1, 0, 0]); dv = self.to_world.transform_vector([0, 1, 0])
            lu, lv = np.linalg.norm(du), np.linalg.norm(dv)
            return float(math.pi * lu * math.sqrt(max(lv * lv - float(np.dot(dv, du / lu)) ** 2, 0.0)))
        p = self.vertex_positions; f = self.faces
        return float(0.5 * np.linalg.norm(np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]]), axis=1).sum())

    def bbox(self):
        if self.type in (abi.SHAPE_RECTANGLE, abi.SHAPE_DISK):      # rectangle.cpp:128-138 (a disk's box is bounded by its square)
            c = np.array([self.to_world.transform_affine(q) for q in ([-1, -1, 0], [-1, 1, 0], [1, -1, 0], [1, 1, 0])])
        elif self.type == abi.SHAPE_SPHERE:       # sphere.cpp:190-195
            ce = self.to_world.transform_affine([0, 0, 0]); r = np.linalg.norm(self.to_world.transform_vector([1, 0, 0]))
            return ce - r, ce + r
        else:
            c = self.vertex_positions
        return c.min(axis=0), c.max(axis=0)


def compute_vertex_normals(pos, faces):
    """Mesh::recompute_vertex_normals (mesh.cpp:321-413, the JIT branch :376-413): face normal n = normalize(cross(v1 - v0, v2 - v0)), weighted by
    the corner angle safe_acos(dot(normalize(v[i+1] - v[i]), normalize(v[i+2] - v[i]))), summed per vertex, normalised; float32.  obj.cpp:401-403
    and ply.cpp:444-446 call it when the file holds no normals (and face_normals is false); Mesh::parameters_changed (mesh.cpp:138-139) whenever
    the positions of a mesh with vertex normals change."""
    f32 = np.float32
    pos = np.asarray(pos, dtype=f32); faces = np.asarray(faces, dtype=np.int64)
    v = [pos[faces[:, k]] for k in range(3)]

    def nrm(a):
        with np.errstate(divide="ignore", invalid="ignore"):
            return (a * (f32(1) / np.sqrt((a * a).sum(axis=1, dtype=f32, keepdims=True)))).astype(f32)
    n = nrm(np.cross(v[1] - v[0], v[2] - v[0]).astype(f32))
    out = np.zeros_like(pos)
    ok = np.isfinite(n).all(axis=1)
    for i in range(3):
        d0 = nrm(v[(i + 1) % 3] - v[i]); d1 = nrm(v[(i + 2) % 3] - v[i])
        ang = np.arccos(np.clip((d0 * d1).sum(axis=1, dtype=f32), f32(-1), f32(1))).astype(f32)
        np.add.at(out, faces[ok, i], (n[ok] * ang[ok, None]).astype(f32))
    ln = np.sqrt((out * out).sum(axis=1, dtype=f32))
    res = np.where(ln[:, None] != 0, out / np.where(ln == 0, f32(1), ln)[:, None], np.array([1, 0, 0], dtype=f32)[None, :])
    return np.ascontiguousarray(res.astype(f32))


_CUBE_V = np.array([[1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, 1, 1],
                    [1, -1, -1], [1, 1, -1], [1, 1, 1], [1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1],
                    [-1, -1, 1], [-1, 1, 1], [-1, 1, -1], [-1, -1, -1], [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], dtype=np.float32)
_CUBE_N = np.repeat(np.array([[0, -1, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1], [-1, 0, 0], [0, 0, -1]], dtype=np.float32), 4, axis=0)
_CUBE_UV = np.tile(np.array([[0, 1], [1, 1], [1, 0], [0, 0]], dtype=np.float32), (6, 1))
_CUBE_F = np.array([[0, 1, 2], [3, 0, 2], [4, 5, 6], [7, 4, 6], [8, 9, 10], [11, 8, 10], [12, 13, 14], [15, 12, 14],
                    [16, 17, 18], [19, 16, 18], [20, 21, 22], [23, 20, 22]], dtype=np.uint32)


def _make_cube(shape):
    """Cube ctor: cube.cpp:105-163 -- vertices and normals pre-transformed to world space."""
    tw = shape.to_world
    m = tw.matrix; it = tw.inverse_transpose
    pos = (_CUBE_V @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
    nrm = (_CUBE_N @ it[:3, :3].T).astype(np.float32)
    nrm = (nrm / np.sqrt((nrm * nrm).sum(axis=1, keepdims=True))).astype(np.float32)
    shape.vertex_positions = np.ascontiguousarray(pos)
    shape.vertex_normals = np.ascontiguousarray(nrm)
    shape.vertex_texcoords = np.ascontiguousarray(_CUBE_UV.copy())
    shape.faces = np.ascontiguousarray(_CUBE_F.copy())


def _load_obj(filename):
    """Minimal Wavefront OBJ reader (v / vt / vn / f with triangles or convex polygons): the vertex/face
    buffer format of src/shapes/obj.cpp (unique (v,vt,vn) triplets become vertices; v is flipped like obj.cpp)."""
    vs, vts, vns, keys, tris = [], [], [], {}, []
    with open(filename) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                vs.append([float(x) for x in t[1:4]])
            elif t[0] == "vt":
                vts.append([float(t[1]), 1.0 - float(t[2])])
            elif t[0] == "vn":
                vns.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = []
                for w in t[1:]:
                    parts = (w.split("/") + ["", ""])[:3]
                    key = tuple(int(p) if p else 0 for p in parts)
                    key = tuple((k + (len(a) + 1) if k < 0 else k) for k, a in zip(key, (vs, vts, vns)))
                    if key not in keys:
                        keys[key] = len(keys)
                    idx.append(keys[key])
                for k in range(1, len(idx) - 1):
                    tris.append([idx[0], idx[k], idx[k + 1]])
    order = sorted(keys.items(), key=lambda kv: kv[1])
    pos = np.array([vs[k[0] - 1] for k, _ in order], dtype=np.float32)
    has_uv = all(k[1] > 0 for k, _ in order) and len(vts) > 0
    has_n = all(k[2] > 0 for k, _ in order) and len(vns) > 0
    uv = np.array([vts[k[1] - 1] for k, _ in order], dtype=np.float32) if has_uv else None
    nrm = np.array([vns[k[2] - 1] for k, _ in order], dtype=np.float32) if has_n else None
    return pos, nrm, uv, np.array(tris, dtype=np.uint32)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _load_ply(filename):
    """Stanford PLY reader (ascii / binary_little_endian / binary_big_endian), the vertex/face buffer format of
    src/shapes/ply.cpp: vertex x y z [nx ny nz] [u v | s t], faces as `vertex_indices` / `vertex_index` lists (triangles, or
    convex polygons fanned like obj.cpp).  Other elements and properties are skipped."""
    with open(filename, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise PluginError('"%s" is not a PLY file' % filename)
        fmt, elements = None, []
        while True:
            line = fh.readline()
            if not line:
                raise PluginError("PLY header of \"%s\" is truncated" % filename)
            t = line.decode("ascii", "replace").split()
            if not t or t[0] in ("comment", "obj_info"):
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                elements.append({"name": t[1], "count": int(t[2]), "props": []})
            elif t[0] == "property":
                if t[1] == "list":
                    elements[-1]["props"].append(("list", _PLY_TYPES[t[2]], _PLY_TYPES[t[3]], t[4]))
                else:
                    elements[-1]["props"].append(("scalar", _PLY_TYPES[t[1]], None, t[2]))
            elif t[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise PluginError('PLY format "%s" is not supported' % fmt)
        end = "<" if fmt != "binary_big_endian" else ">"
        verts, faces = {}, []
        for el in elements:
            scalar_only = all(p[0] == "scalar" for p in el["props"])
            if fmt == "ascii":
                rows = [fh.readline().split() for _ in range(el["count"])]
                if el["name"] == "vertex":
                    for j, p in enumerate(el["props"]):
                        verts[p[3]] = np.array([float(r[j]) for r in rows], dtype=np.float64)
                elif el["name"] == "face":
                    for r in rows:
                        k = 0
                        for p in el["props"]:
                            if p[0] == "list":
                                n = int(r[k]); vals = [int(x) for x in r[k + 1:k + 1 + n]]; k += 1 + n
                                if p[3] in ("vertex_indices", "vertex_index"):
                                    faces.extend([vals[0], vals[i], vals[i + 1]] for i in range(1, n - 1))
                            else:
                                k += 1
                continue
            if scalar_only:
                dt = np.dtype([(p[3], end + p[1]) for p in el["props"]])
                data = np.frombuffer(fh.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
                if el["name"] == "vertex":
                    for p in el["props"]:
                        verts[p[3]] = data[p[3]].astype(np.float64)
                continue
            for _ in range(el["count"]):                      # lists: variable-length rows
                for p in el["props"]:
                    if p[0] == "list":
                        n = int(np.frombuffer(fh.read(np.dtype(p[1]).itemsize), dtype=end + p[1])[0])
                        vals = np.frombuffer(fh.read(np.dtype(p[2]).itemsize * n), dtype=end + p[2]).astype(np.int64)
                        if el["name"] == "face" and p[3] in ("vertex_indices", "vertex_index"):
                            faces.extend([int(vals[0]), int(vals[i]), int(vals[i + 1])] for i in range(1, n - 1))
                    else:
                        fh.read(np.dtype(p[1]).itemsize)
    if not all(k in verts for k in ("x", "y", "z")) or not faces:
        raise PluginError('PLY file "%s" has no x/y/z vertices or no faces' % filename)
    pos = np.stack([verts["x"], verts["y"], verts["z"]], axis=1).astype(np.float32)
    nrm = np.stack([verts["nx"], verts["ny"], verts["nz"]], axis=1).astype(np.float32) if all(k in verts for k in ("nx", "ny", "nz")) else None
    uv = None
    for a, b in (("u", "v"), ("s", "t"), ("texture_u", "texture_v")):
        if a in verts and b in verts:
            uv = np.stack([verts[a], verts[b]], axis=1).astype(np.float32)
            break
    f = np.array(faces, dtype=np.uint32)
    if f.max() >= pos.shape[0]:
        raise PluginError('PLY file "%s": face index out of range' % filename)
    return pos, nrm, uv, f


class Film:
    def __init__(self, props):
        t = props.get("type", "hdrfilm")
        if t != "hdrfilm":
            raise PluginError('film plugin "%s" is outside the path (hdrfilm only)' % t)
        self.width = int(props.get("width", 768)); self.height = int(props.get("height", 576))
        self.crop_width = int(props.get("crop_width", self.width)); self.crop_height = int(props.get("crop_height", self.height))
        self.crop_offset_x = int(props.get("crop_offset_x", 0)); self.crop_offset_y = int(props.get("crop_offset_y", 0))
        pf = str(props.get("pixel_format", "rgb")).lower()
        if pf != "rgb":
            raise PluginError('pixel_format "%s" is outside the path (rgb only: RGBW accumulation, hdrfilm.cpp:245-259)' % pf)
        rf = props.get("rfilter", {"type": "gaussian"})
        rt = rf.get("type", "gaussian")
        if rt == "gaussian":
            self.rfilter = abi.RFILTER_GAUSSIAN; self.rfilter_param = float(rf.get("stddev", 0.5)); self.radius = 4 * self.rfilter_param
        elif rt == "box":
            self.rfilter = abi.RFILTER_BOX; self.rfilter_param = 0.5; self.radius = 0.5
        elif rt == "tent":
            self.rfilter = abi.RFILTER_TENT; self.rfilter_param = float(rf.get("radius", 1.0)); self.radius = self.rfilter_param
        else:
            raise PluginError('rfilter plugin "%s" is outside the path (box, gaussian, tent)' % rt)

        self.sample_border = bool(props.get("sample_border", False))         # film.cpp:32
        # ReconstructionFilter::border_size (rfilter.cpp:23), RayEpsilon = 1500 * 2^-24 (math.h:17-23)
        self.border_size = int(math.ceil(np.float32(self.radius) - np.float32(0.5) - np.float32(2.0) * np.float32(1500.0 * 2.0 ** -24)))
        self.diagonal = 0.001 * float(props.get("diagonal", 35.0))          # fork: physical film diagonal [m], film.cpp:46-48

    def physical_size(self):
        """Film::get_physical_size (film.h:193-198): HALF extents in metres."""
        aspect = float(self.height) / float(self.width)
        x = math.sqrt(self.diagonal * self.diagonal / (1.0 + aspect * aspect))
        return (x / 2, aspect * x / 2)

    def size(self): return (self.width, self.height)
    def crop_size(self): return (self.crop_width, self.crop_height)
    def crop_offset(self): return (self.crop_offset_x, self.crop_offset_y)


class Sensor:
    def __init__(self, props, id="sensor"):
        self.id = id
        t = props.get("type", "perspective")
        if t not in ("perspective", "realisticlens"):
            raise PluginError('sensor plugin "%s" is outside the path in this build (perspective, realisticlens)' % t)
        self.type = abi.SENSOR_PERSPECTIVE if t == "perspective" else abi.SENSOR_REALISTICLENS
        self.film = Film(props.get("film", {}))
        self.to_world = props.get("to_world", Transform4f())
        if not isinstance(self.to_world, Transform4f):
            self.to_world = Transform4f(self.to_world)
        if self.to_world.has_scale():
            raise PluginError("Scale factors in the camera-to-world transformation are not allowed!")
        self.near_clip = float(props.get("near_clip", 1e-2)); self.far_clip = float(props.get("far_clip", 1e4))
        self.lens_table = None; self.interfaces = None; self.sample_exit_pupil = False
        # Sensor::Sensor (sensor.cpp): shutter_open / shutter_close, open time = close - open
        self.shutter_open = float(props.get("shutter_open", 0.0)); self.shutter_open_time = float(props.get("shutter_close", 0.0)) - self.shutter_open
        if self.shutter_open_time < 0:
            raise PluginError("Shutter opening time must be less than or equal to the shutter closing time!")
        if self.type == abi.SENSOR_REALISTICLENS:       # realisticlens.cpp:832-907
            from . import lens as _lens
            self.sample_exit_pupil = bool(props.get("sample_exit_pupil", False))       # realisticlens.cpp:851
            try:
                self.interfaces, self.focus_delta = _lens.build_lens(props, self.film.physical_size()[0])
            except NotImplementedError as e:
                raise PluginError(str(e))
            self.lens_table = _lens.lens_table(self.interfaces)
            film_z = getattr(self.interfaces[0], "film_z_position", 0.0)
            if film_z:              # ray.o.z += m_film_z_position before to_world (realisticlens.cpp:2263)
                self.to_world = self.to_world.translate([0.0, 0.0, film_z])
            self.x_fov = 0.0
        else:
            self.x_fov = float(np.float32(parse_fov(props, self.film.width / float(self.film.height))))
            if float(props.get("principal_point_offset_x", 0.0)) != 0.0 or float(props.get("principal_point_offset_y", 0.0)) != 0.0:
                # perspective.cpp:147-149, :215-221, :296-301: not carried by the engine -- refuse rather than render with the offset ignored
                raise PluginError("perspective: a non-zero principal_point_offset is outside the path in this build")
        smp = props.get("sampler", {"type": "independent"})
        if smp.get("type", "independent") != "independent":
            raise PluginError('sampler plugin "%s" is outside the path (independent)' % smp.get("type"))
        self.sample_count = int(smp.get("sample_count", 4)); self.base_seed = int(smp.get("seed", 0))


class Scene:
    def __init__(self):
        self.shapes, self.bsdfs, self.emitters, self.textures = [], [], [], []
        self._sensors = []               # Scene::sensors(): every sensor of the scene, in the order the scene lists them
        self.integrator_props = None
        self._handle = None
        self._device = None
        self._keep = []

    # ---- Mitsuba-like accessors ------------------------------------------------------------------
    def sensors(self): return self._sensors

    @property
    def sensor(self):
        """The first sensor -- what every `sensor=0` default of the reference's API means (util.py:410-523, integrator.h:74-97)."""
        return self._sensors[0] if self._sensors else None

    @sensor.setter
    def sensor(self, value):
        if self._sensors:
            self._sensors[0] = value
        else:
            self._sensors.append(value)

    def sensor_index(self, sensor):
        """`sensor` as the reference's API takes it: an index into sensors(), or a Sensor object of this scene."""
        if isinstance(sensor, Sensor):
            for i, se in enumerate(self._sensors):
                if se is sensor:
                    return i
            raise PluginError("the sensor does not belong to this scene")
        i = int(sensor)
        if not 0 <= i < len(self._sensors):
            raise PluginError("Out-of-bound sensor index %d (the scene has %d sensors)" % (i, len(self._sensors)))     # util.py:489-491
        return i
    def integrator(self):
        from .integrators import make_integrator
        return make_integrator(self.integrator_props) if self.integrator_props else None

    # ---- flatten ---------------------------------------------------------------------------------
    def desc(self):
        """Build the ctypes mi355_scene_desc (keeps the numpy buffers alive in self._keep)."""
        keep = []
        tex = (abi.TextureDesc * max(len(self.textures), 1))()
        for i, t in enumerate(self.textures):
            tex[i].width, tex[i].height, tex[i].channels = t.data.shape[1], t.data.shape[0], t.data.shape[2]
            tex[i].filter, tex[i].wrap = t.filter, t.wrap
            tex[i].to_uv[:] = t.to_uv.tolist()
            tex[i].data = t.data.ctypes.data_as(abi.f32p); keep.append(t.data)
        bs = (abi.BsdfDesc * max(len(self.bsdfs), 1))()
        for i, b in enumerate(self.bsdfs):
            bs[i].type = b.type; bs[i].reflectance[:] = b.reflectance.tolist()
            bs[i].reflectance_texture = self.textures.index(b.texture) if b.texture is not None else -1
            bs[i].alpha_u, bs[i].alpha_v, bs[i].distribution = b.alpha_u, b.alpha_v, b.distribution
            bs[i].sample_visible = 1 if b.sample_visible else 0
            bs[i].eta[:] = b.eta.tolist(); bs[i].k[:] = b.k.tolist()
            bs[i].int_ior, bs[i].ext_ior = b.int_ior, b.ext_ior
            bs[i].transmittance[:] = b.transmittance.tolist()
        em = (abi.EmitterDesc * max(len(self.emitters), 1))()
        for i, e in enumerate(self.emitters):
            em[i].type = e.type; em[i].radiance[:] = e.radiance.tolist(); em[i].shape = self.shapes.index(e.shape)
        sh = (abi.ShapeDesc * max(len(self.shapes), 1))()
        for i, s in enumerate(self.shapes):
            sh[i].type = s.type; sh[i].bsdf = self.bsdfs.index(s.bsdf)
            sh[i].emitter = self.emitters.index(s.emitter) if s.emitter is not None else -1
            sh[i].flip_normals = 1 if s.flip_normals else 0
            sh[i].to_world[:] = s.to_world.affine12().tolist(); sh[i].to_object[:] = s.to_world.inverse_affine12().tolist()
            if s.type == abi.SHAPE_MESH:
                s.sync_host()
                sh[i].vertex_count = s.vertex_positions.shape[0]; sh[i].face_count = s.faces.shape[0]
                sh[i].vertex_positions = s.vertex_positions.ctypes.data_as(abi.f32p)
                sh[i].faces = s.faces.ctypes.data_as(abi.u32p)
                if s.vertex_normals is not None:
                    sh[i].vertex_normals = s.vertex_normals.ctypes.data_as(abi.f32p)
                if s.vertex_texcoords is not None:
                    sh[i].vertex_texcoords = s.vertex_texcoords.ctypes.data_as(abi.f32p)
                keep += [s.vertex_positions, s.faces, s.vertex_normals, s.vertex_texcoords]
        d = abi.SceneDesc()
        d.abi_version = abi.ABI_VERSION
        d.n_shapes, d.n_bsdfs, d.n_emitters, d.n_textures = len(self.shapes), len(self.bsdfs), len(self.emitters), len(self.textures)
        d.shapes, d.bsdfs, d.emitters, d.textures = sh, bs, em, tex
        sens = (abi.SensorDesc * max(len(self._sensors), 1))()
        for i in range(len(self._sensors)):
            self.sensor_desc(i, out=sens[i], keep=keep)
        d.n_sensors, d.sensors = len(self._sensors), sens
        keep += [tex, bs, em, sh, sens]
        self._keep = keep
        return d

    def sensor_desc(self, index, out=None, keep=None):
        """mi355_sensor_desc of sensor `index` (kept alive by the Sensor object's own buffers)."""
        se = self._sensors[index]; f = se.film
        d = abi.SensorDesc() if out is None else out
        d.type = se.type
        if se.lens_table is not None:
            d.lens_count = se.lens_table.shape[0]
            d.lens_data = se.lens_table.ctypes.data_as(abi.f32p)
            if keep is not None:
                keep.append(se.lens_table)
        d.film_diagonal = f.diagonal
        d.to_world[:] = se.to_world.affine12().tolist()
        d.shutter_open_time = se.shutter_open_time; d.sample_exit_pupil = 1 if se.sample_exit_pupil else 0
        d.sample_border = 1 if f.sample_border else 0
        d.x_fov, d.near_clip, d.far_clip = se.x_fov, se.near_clip, se.far_clip
        d.film_width, d.film_height = f.width, f.height
        d.crop_offset_x, d.crop_offset_y = f.crop_offset_x, f.crop_offset_y
        d.crop_width, d.crop_height = f.crop_width, f.crop_height
        d.rfilter, d.rfilter_param = f.rfilter, f.rfilter_param
        return d

    def push_sensor(self, index):
        """Sensor `index` changed on the host (film size, crop window, type ...): hand its new description to the device replica
        (mi355_scene_set_sensor -- no geometry, BVH or scratch buffer is touched; without a replica yet there is nothing to do)."""
        if self._handle is not None:
            d = self.sensor_desc(index)
            abi.check(abi.lib().mi355_scene_set_sensor(self._handle, int(index), C.byref(d)))

    # ---- engine handle ------------------------------------------------------------------------------
    def handle(self, device=None):
        """Create (once) the device replica through the C ABI.  Fails loudly without a GPU."""
        import torch
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        if self._handle is not None and self._device == device:
            return self._handle
        if self._handle is not None:
            self.release()
        h = C.c_void_p()
        d = self.desc()
        abi.check(abi.lib().mi355_scene_create(C.byref(d), int(device), C.byref(h)))
        self._handle, self._device = h, device
        return h

    def set_reduce_hook(self, fn):
        """mi355_set_reduce_hook: `fn(user, d_buffer, count, stream) -> int` sums a device buffer over all ranks in place (None removes
        it).  The host layer itself exchanges through torch.distributed and leaves the hook unset; this is the C shim's route."""
        self._reduce_cb = abi.REDUCE_FN(fn) if fn is not None else None      # (kept alive with the scene)
        abi.check(abi.lib().mi355_set_reduce_hook(self.handle(), C.cast(self._reduce_cb, C.c_void_p) if fn is not None else None, None))

    def info(self):
        """Geometry organisation of the device replica (mi355_scene_info)."""
        out = (C.c_uint32 * 8)()
        abi.check(abi.lib().mi355_scene_info(self.handle(), out))
        return {"flat_prims": out[0], "boxes": out[1], "bvh_meshes": out[2], "bvh_nodes": out[3], "triangles": out[4], "grad_floats": out[5],
                "spheres": out[6], "features": out[7]}

    def release(self):
        if self._handle is not None:
            abi.lib().mi355_scene_destroy(self._handle)
            self._handle = None
            self._grad_enabled = {}

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# load_dict
# ------------------------------------------------------------------------------------------------
def _rgb(v, what, bounded):
    """{'type':'rgb','value':..} or a plain float/list.  Reflectances must lie in [0,1] (srgb.cpp:52-66);
    emitter radiances are unbounded (xml.cpp:1147-1163, d65.cpp:146-155)."""
    if isinstance(v, dict):
        t = v.get("type")
        if t == "rgb":
            v = v["value"]
        elif t in ("uniform", "d65"):
            raise PluginError('spectrum plugin "%s" is outside the RGB path' % t)
        else:
            raise PluginError('texture plugin "%s" cannot be used as a constant colour' % t)
    a = np.asarray(v, dtype=np.float32)
    if a.ndim == 0:
        a = np.full(3, a, dtype=np.float32)
    a = a.reshape(3)
    if bounded and (np.any(a < 0) or np.any(a > 1)):
        raise PluginError("%s: invalid RGB reflectance value %s, must be in the range [0, 1]!" % (what, a.tolist()))
    return a


def _make_bsdf(scene, props, id):
    t = props.get("type")
    if t == "roughconductor":
        if props.get("material", "none") != "none":
            raise PluginError("roughconductor: named materials need the reference's IOR data files (absent); give eta / k")
        b = BSDF(id, _rgb(props.get("specular_reflectance", 1.0), "roughconductor.specular_reflectance", True), type=abi.BSDF_ROUGHCONDUCTOR)
        d = str(props.get("distribution", "beckmann")).lower()
        if d not in ("beckmann", "ggx"):
            raise PluginError('Specified an invalid distribution "%s", must be "beckmann" or "ggx"!' % d)
        b.distribution = 0 if d == "beckmann" else 1
        if "alpha_u" in props or "alpha_v" in props:
            if not ("alpha_u" in props and "alpha_v" in props):
                raise PluginError("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.")
            if "alpha" in props:
                raise PluginError("Microfacet model: please specify either 'alpha' or 'alpha_u'/'alpha_v'.")
            b.alpha_u, b.alpha_v = float(props["alpha_u"]), float(props["alpha_v"])
        else:
            b.alpha_u = b.alpha_v = float(props.get("alpha", 0.1))
        b.sample_visible = bool(props.get("sample_visible", True))
        b.eta = _rgb(props.get("eta", 0.0), "roughconductor.eta", False); b.k = _rgb(props.get("k", 1.0), "roughconductor.k", False)
        scene.bsdfs.append(b)
        return b
    if t == "conductor":                                        # conductor.cpp:222-241
        if "eta" not in props and props.get("material", "none") != "none":
            raise PluginError("conductor: named materials need the reference's IOR data files (absent); give eta / k")
        if "eta" in props and props.get("material", "none") != "none":
            raise PluginError("Should specify either (eta, k) or material, not both.")
        b = BSDF(id, _rgb(props.get("specular_reflectance", 1.0), "conductor.specular_reflectance", True), type=abi.BSDF_CONDUCTOR)
        b.eta = _rgb(props.get("eta", 0.0), "conductor.eta", False); b.k = _rgb(props.get("k", 1.0), "conductor.k", False)
        scene.bsdfs.append(b)
        return b
    if t == "dispersive":       # dispersive_dielectric.cpp: in RGB variants both media use their Cauchy IOR at 589.3 nm (:411-432)
        def eval_ior(ior_d, v_d, wl=np.float32(0.5893)):
            ior_d, v_d = np.float32(ior_d), np.float32(v_d)
            inv = lambda x: np.float32(1) / np.float32(x)
            b_ = (ior_d - np.float32(1)) * inv(v_d * (inv(np.float32(0.48613) ** 2) - inv(np.float32(0.65627) ** 2))) if v_d > np.float32(5.96e-8) else np.float32(0)
            a_ = ior_d - b_ * inv(np.float32(0.5893) ** 2)
            return np.float32(a_ + b_ * inv(wl * wl))
        b = BSDF(id, _rgb(props.get("specular_reflectance", 1.0), "dispersive.specular_reflectance", True), type=abi.BSDF_DIELECTRIC)
        b.transmittance = _rgb(props.get("specular_transmittance", 1.0), "dispersive.specular_transmittance", True)
        # eta = eval_ior(int) * rcp(eval_ior(ext))  (:250): the two indices at 589.3 nm are handed over separately (they are parameters, :236-239)
        b.dispersive = {"int_ior_d": float(props.get("int_ior", 1.5046)), "int_V_d": float(props.get("int_V_d", 64.17)),
                        "ext_ior_d": float(props.get("ext_ior", 1.0)), "ext_V_d": float(props.get("ext_V_d", 0.0))}
        b.eval_ior = eval_ior
        b.int_ior = float(eval_ior(b.dispersive["int_ior_d"], b.dispersive["int_V_d"])); b.ext_ior = float(eval_ior(b.dispersive["ext_ior_d"], b.dispersive["ext_V_d"]))
        scene.bsdfs.append(b)
        return b
    if t == "dielectric":
        b = BSDF(id, _rgb(props.get("specular_reflectance", 1.0), "dielectric.specular_reflectance", True), type=abi.BSDF_DIELECTRIC)
        b.transmittance = _rgb(props.get("specular_transmittance", 1.0), "dielectric.specular_transmittance", True)
        b.int_ior = lookup_ior(props, "int_ior", "bk7"); b.ext_ior = lookup_ior(props, "ext_ior", "air")
        if b.int_ior < 0 or b.ext_ior < 0:
            raise PluginError("The interior and exterior indices of refraction must be positive!")
        scene.bsdfs.append(b)
        return b
    if t != "diffuse":
        raise PluginError('bsdf plugin "%s" is outside the path in this build (diffuse, roughconductor, conductor, dielectric)' % t)
    refl = props.get("reflectance", 0.5)
    if isinstance(refl, dict) and refl.get("type") == "bitmap":
        if "data" not in refl:
            raise PluginError("bitmap textures must be given as a tensor ('data') in this build (no image file I/O)")
        if not refl.get("raw", True):
            raise PluginError('Bitmap "raw" parameter must be `true` when initializing using tensor data!')
        to_uv = refl.get("to_uv")
        if isinstance(to_uv, Transform4f):
            m = to_uv.matrix; to_uv = [m[0, 0], m[0, 1], m[0, 3], m[1, 0], m[1, 1], m[1, 3]]
        tex = Texture(refl["data"], refl.get("filter_type", "bilinear"), refl.get("wrap_mode", "repeat"), to_uv, id=id)
        scene.textures.append(tex)
        b = BSDF(id, (0.5, 0.5, 0.5), tex)
    else:
        b = BSDF(id, _rgb(refl, "diffuse.reflectance", True))
    scene.bsdfs.append(b)
    return b


BSDF_PLUGINS = ("diffuse", "roughconductor", "conductor", "dielectric", "dispersive")


def load_dict(d):
    """mi.load_dict for the plugins on the path.  Returns a Scene (type 'scene') or a single-shape Scene-less
    object for shape dicts used by unit tests."""
    t = d.get("type")
    if t != "scene":
        # convenience used by the reference's shape tests: mi.load_dict({'type': 'rectangle', ...})
        sc = load_dict({"type": "scene", "shape": d})
        return sc.shapes[0]
    scene = Scene()
    named = {}
    pending_shapes = []
    for key, v in d.items():
        if key == "type" or not isinstance(v, dict):
            continue
        vt = v.get("type")
        if vt in ("path", "prb", "ptracer", "prb_basic"):
            scene.integrator_props = dict(v)
        elif vt in ("perspective", "realisticlens"):
            scene._sensors.append(Sensor(v, id=key))
        elif vt in BSDF_PLUGINS:
            named[key] = _make_bsdf(scene, v, key)
        elif vt in ("rectangle", "cube", "obj", "ply", "mesh", "sphere", "disk"):
            pending_shapes.append((key, v))
        elif vt in ("thinlens", "orthographic", "radialwarp", "volpath", "direct", "aov",
                    "cylinder", "serialized", "constant", "point", "envmap", "directional",
                    "plastic", "twosided"):
            raise PluginError('plugin "%s" is outside the hot path this engine replaces' % vt)
        else:
            raise PluginError('Plugin "%s" not found!' % vt)
    default_bsdf = None
    for key, v in pending_shapes:
        vt = v["type"]
        tw = v.get("to_world", Transform4f())
        if not isinstance(tw, Transform4f):
            tw = Transform4f(tw)
        # children are recognised by their plugin class, not by their key (xml.cpp / Shape::Shape, shape.cpp:39-72)
        bp = ep = None
        for ck, cv in v.items():
            if not isinstance(cv, dict) or "type" not in cv:
                continue
            ct = cv["type"]
            if ct in BSDF_PLUGINS or (ct == "ref" and cv.get("id") in named):
                if bp is not None:
                    raise PluginError("Only a single BSDF child object can be specified per shape.")
                bp = cv
            elif ct in ("area", "directionalarea"):
                if ep is not None:
                    raise PluginError("Only a single Emitter child object can be specified per shape.")
                ep = cv
            elif ct == "ref":
                raise PluginError('Referenced object "%s" not found!' % cv.get("id"))
            elif ct in ("rgb", "bitmap", "spectrum"):
                continue
            else:
                raise PluginError('shape child plugin "%s" is outside the path (bsdf: %s; emitter: area, directionalarea)' % (ct, ", ".join(BSDF_PLUGINS)))
        if bp is None:
            if default_bsdf is None:
                default_bsdf = _make_bsdf(scene, {"type": "diffuse"}, "__default_bsdf")
            bsdf = default_bsdf
        elif bp.get("type") == "ref":
            if bp["id"] not in named:
                raise PluginError('Referenced object "%s" not found!' % bp["id"])
            bsdf = named[bp["id"]]
        else:
            bsdf = _make_bsdf(scene, bp, key + ".bsdf")
        kind = {"rectangle": abi.SHAPE_RECTANGLE, "sphere": abi.SHAPE_SPHERE, "disk": abi.SHAPE_DISK}.get(vt, abi.SHAPE_MESH)
        if vt == "disk" and v.get("flip_normals", False):      # disk.cpp:98-101: folded into to_world
            tw = tw @ ScalarTransform4f.scale([1.0, 1.0, -1.0])
        if vt == "sphere":      # sphere.cpp:131-146: to_world * translate(center) * scale(radius)
            tw = tw @ ScalarTransform4f.translate(v.get("center", [0, 0, 0])) @ ScalarTransform4f.scale(float(v.get("radius", 1.0)))
        shape = Shape(key, kind, tw, bsdf, flip_normals=bool(v.get("flip_normals", False)))
        if vt == "cube":
            _make_cube(shape)
        elif vt in ("obj", "ply"):
            pos, nrm, uv, faces = (_load_obj if vt == "obj" else _load_ply)(v["filename"])
            m, it = tw.matrix, tw.inverse_transpose
            shape.vertex_positions = np.ascontiguousarray((pos @ m[:3, :3].T + m[:3, 3]).astype(np.float32))
            shape.faces = np.ascontiguousarray(faces)
            if nrm is not None and not v.get("face_normals", False):
                n = (nrm @ it[:3, :3].T).astype(np.float32)
                shape.vertex_normals = np.ascontiguousarray((n / np.sqrt((n * n).sum(axis=1, keepdims=True))).astype(np.float32))
            elif not v.get("face_normals", False):          # obj.cpp:401-403 / ply.cpp:444-446: no normals in the file -> computed
                shape.vertex_normals = compute_vertex_normals(shape.vertex_positions, shape.faces)
            shape.vertex_texcoords = None if uv is None else np.ascontiguousarray(uv)
        elif vt == "mesh":     # extension: buffers as mi.Mesh exposes them through traverse() (mesh.h:523-527)
            m, it = tw.matrix, tw.inverse_transpose
            pos = np.asarray(v["vertex_positions"], dtype=np.float32).reshape(-1, 3)
            shape.vertex_positions = np.ascontiguousarray((pos @ m[:3, :3].T + m[:3, 3]).astype(np.float32))
            if v.get("vertex_normals") is not None:
                n = (np.asarray(v["vertex_normals"], dtype=np.float32).reshape(-1, 3) @ it[:3, :3].T).astype(np.float32)
                shape.vertex_normals = np.ascontiguousarray((n / np.sqrt((n * n).sum(axis=1, keepdims=True))).astype(np.float32))
            elif v.get("compute_vertex_normals", False):      # what the ply / obj loaders do for a file without normals
                faces_ = np.asarray(v["faces"], dtype=np.uint32).reshape(-1, 3)
                shape.vertex_normals = compute_vertex_normals(shape.vertex_positions, faces_)
            if v.get("vertex_texcoords") is not None:
                shape.vertex_texcoords = np.ascontiguousarray(np.asarray(v["vertex_texcoords"], dtype=np.float32).reshape(-1, 2))
            shape.faces = np.ascontiguousarray(np.asarray(v["faces"], dtype=np.uint32).reshape(-1, 3))
        if ep is not None:
            e = Emitter(_rgb(ep.get("radiance", 1.0), "area.radiance", False))
            e.type = abi.EMITTER_AREA if ep.get("type") == "area" else abi.EMITTER_DIRECTIONALAREA
            e.shape = shape; shape.emitter = e
            scene.emitters.append(e)
        scene.shapes.append(shape)
    if not scene._sensors:
        scene._sensors.append(Sensor({"type": "perspective", "film": {"type": "hdrfilm"}}))
    return scene


def cornell_box():
    """mi.cornell_box(): src/python/python/util.py:565-699, verbatim values."""
    T = ScalarTransform4f
    return {
        'type': 'scene',
        'integrator': {'type': 'path', 'max_depth': 8},
        'sensor': {
            'type': 'perspective', 'fov_axis': 'smaller', 'near_clip': 0.001, 'far_clip': 100.0,
            'focus_distance': 1000, 'fov': 39.3077,
            'to_world': T.look_at(origin=[0, 0, 3.90], target=[0, 0, 0], up=[0, 1, 0]),
            'sampler': {'type': 'independent', 'sample_count': 64},
            'film': {'type': 'hdrfilm', 'width': 256, 'height': 256, 'rfilter': {'type': 'gaussian'},
                     'pixel_format': 'rgb', 'component_format': 'float32'},
        },
        'white': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.885809, 0.698859, 0.666422]}},
        'green': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.105421, 0.37798, 0.076425]}},
        'red': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.570068, 0.0430135, 0.0443706]}},
        'light': {
            'type': 'rectangle',
            'to_world': T.translate([0.0, 0.99, 0.01]).rotate([1, 0, 0], 90).scale([0.23, 0.19, 0.19]),
            'bsdf': {'type': 'ref', 'id': 'white'},
            'emitter': {'type': 'area', 'radiance': {'type': 'rgb', 'value': [18.387, 13.9873, 6.75357]}},
        },
        'floor': {'type': 'rectangle', 'to_world': T.translate([0.0, -1.0, 0.0]).rotate([1, 0, 0], -90),
                  'bsdf': {'type': 'ref', 'id': 'white'}},
        'ceiling': {'type': 'rectangle', 'to_world': T.translate([0.0, 1.0, 0.0]).rotate([1, 0, 0], 90),
                    'bsdf': {'type': 'ref', 'id': 'white'}},
        'back': {'type': 'rectangle', 'to_world': T.translate([0.0, 0.0, -1.0]), 'bsdf': {'type': 'ref', 'id': 'white'}},
        'green-wall': {'type': 'rectangle', 'to_world': T.translate([1.0, 0.0, 0.0]).rotate([0, 1, 0], -90),
                       'bsdf': {'type': 'ref', 'id': 'green'}},
        'red-wall': {'type': 'rectangle', 'to_world': T.translate([-1.0, 0.0, 0.0]).rotate([0, 1, 0], 90),
                     'bsdf': {'type': 'ref', 'id': 'red'}},
        'small-box': {'type': 'cube', 'to_world': T.translate([0.335, -0.7, 0.38]).rotate([0, 1, 0], -17).scale(0.3),
                      'bsdf': {'type': 'ref', 'id': 'white'}},
        'large-box': {'type': 'cube', 'to_world': T.translate([-0.33, -0.4, -0.28]).rotate([0, 1, 0], 18.25).scale([0.3, 0.61, 0.3]),
                      'bsdf': {'type': 'ref', 'id': 'white'}},
    }


def veach_mis_scene(res=256):
    """Build-authored analogue of the Veach MIS test scene (BASELINE config 4; the reference's scene file is not in its
    tree): four `roughconductor` plates of roughness 0.005 / 0.02 / 0.05 / 0.1 tilted towards the camera, four
    sphere `area` emitters of radius 0.03 / 0.1 / 0.3 / 0.9 (x0.25) with equal power and different colours, one `dielectric`
    sphere, diffuse floor and back wall, one dim rectangular fill light."""
    T = ScalarTransform4f
    d = {
        'type': 'scene',
        'integrator': {'type': 'path', 'max_depth': 8},
        'sensor': {'type': 'perspective', 'fov': 38, 'near_clip': 0.01, 'far_clip': 100.0,
                   'to_world': T.look_at(origin=[0, 2.0, 7.5], target=[0, 0.2, 0], up=[0, 1, 0]),
                   'sampler': {'type': 'independent', 'sample_count': 64},
                   'film': {'type': 'hdrfilm', 'width': res, 'height': res, 'rfilter': {'type': 'gaussian'}, 'pixel_format': 'rgb'}},
        'grey': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.4, 0.4, 0.4]}},
        'floor': {'type': 'rectangle', 'to_world': T.translate([0, -1.2, 0]).rotate([1, 0, 0], -90).scale(6.0), 'bsdf': {'type': 'ref', 'id': 'grey'}},
        'back': {'type': 'rectangle', 'to_world': T.translate([0, 2.0, -3.5]).scale([6.0, 4.0, 1.0]), 'bsdf': {'type': 'ref', 'id': 'grey'}},
        'fill': {'type': 'rectangle', 'to_world': T.translate([0, 6.0, 2.0]).rotate([1, 0, 0], 90).scale(1.5), 'bsdf': {'type': 'ref', 'id': 'grey'},
                 'emitter': {'type': 'area', 'radiance': {'type': 'rgb', 'value': [1.0, 1.0, 1.0]}}},
        'glass': {'type': 'sphere', 'center': [1.75, -0.75, 2.6], 'radius': 0.45, 'bsdf': {'type': 'dielectric', 'int_ior': 'bk7', 'ext_ior': 'air'}},
    }
    alphas = [0.005, 0.02, 0.05, 0.1]
    for i, a in enumerate(alphas):
        tilt = -38.0 + 9.0 * i
        d['plate%d' % i] = {'type': 'rectangle',
                            'to_world': T.translate([0, -0.9 + 0.42 * i, 1.6 - 0.95 * i]).rotate([1, 0, 0], -90 - tilt).scale([2.2, 0.38, 1.0]),
                            'bsdf': {'type': 'roughconductor', 'alpha': a, 'distribution': 'ggx' if i % 2 else 'beckmann',
                                     'eta': [0.2, 0.92, 1.1], 'k': [3.9, 2.45, 2.14]}}
    radii = [0.03, 0.1, 0.3, 0.9]
    colors = [[1.0, 0.35, 0.3], [0.35, 1.0, 0.4], [0.35, 0.45, 1.0], [1.0, 0.9, 0.5]]
    for i, r in enumerate(radii):
        rr = 0.25 * r
        power = 40.0
        rad = [power * c / (4 * math.pi * rr * rr * math.pi) for c in colors[i]]
        d['lamp%d' % i] = {'type': 'sphere', 'center': [-1.65 + 1.1 * i, 1.55, -1.9], 'radius': rr, 'bsdf': {'type': 'ref', 'id': 'grey'},
                           'emitter': {'type': 'area', 'radiance': {'type': 'rgb', 'value': rad}}}
    return d
