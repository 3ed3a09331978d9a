"""`mi.load_file`: the subset of Mitsuba's XML scene format (src/core/xml.cpp) that the path's plugins use, translated into the
dictionary form `load_dict` consumes.

Supported: <scene>, <default name value> with $name substitution (overridable through keyword arguments, like
mi.load_file(path, spp=4)), plugin elements <integrator> <sensor> <sampler> <film> <rfilter> <bsdf> <shape> <emitter> <texture>
with `type` / `id` / `name`, <ref id [name]>, the property elements <float> <integer> <boolean> <string> <rgb> <spectrum> (constant)
<vector> <point>, and <transform name> with <translate> <scale> <rotate> <lookat> <matrix> children -- each child is applied
AFTER the ones before it, as xml.cpp does.  `filename` properties are resolved relative to the XML file.
Anything else raises PluginError: nothing is skipped silently.
"""
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from .transforms import ScalarTransform4f

PLUGIN_TAGS = ("integrator", "sensor", "sampler", "film", "rfilter", "bsdf", "shape", "emitter", "texture")


def _err(msg):
    from .scene import PluginError
    return PluginError(msg)


def _floats(text):
    return [float(x) for x in re.split(r"[\s,]+", text.strip()) if x]


def _vec3(el, default=0.0):
    if "value" in el.attrib:
        v = _floats(el.attrib["value"])
        return v * 3 if len(v) == 1 else v
    return [float(el.attrib.get(k, default)) for k in ("x", "y", "z")]


def _transform(el):
    t = ScalarTransform4f()
    for op in el:
        if op.tag == "translate":
            m = ScalarTransform4f.translate(_vec3(op))
        elif op.tag == "scale":
            m = ScalarTransform4f.scale(_vec3(op, 1.0))
        elif op.tag == "rotate":
            m = ScalarTransform4f.rotate(_vec3(op), float(op.attrib["angle"]))
        elif op.tag == "lookat":
            m = ScalarTransform4f.look_at(_floats(op.attrib["origin"]), _floats(op.attrib["target"]), _floats(op.attrib.get("up", "0, 1, 0")))
        elif op.tag == "matrix":
            v = _floats(op.attrib["value"])
            if len(v) == 9:
                v = v[0:3] + [0.0] + v[3:6] + [0.0] + v[6:9] + [0.0, 0.0, 0.0, 0.0, 1.0]
            if len(v) != 16:
                raise _err("<matrix> needs 9 or 16 values")
            m = ScalarTransform4f(np.array(v, dtype=np.float64).reshape(4, 4))
        else:
            raise _err('transform operation <%s> is not supported' % op.tag)
        t = m @ t                                       # later operations are applied after earlier ones (xml.cpp)
    return t


def _plugin(el, base_dir, ids, counter):
    if "type" not in el.attrib:
        raise _err("<%s> needs a type" % el.tag)
    out = {"type": el.attrib["type"]}
    if "id" in el.attrib:
        out["id"] = el.attrib["id"]
    for ch in el:
        tag, name = ch.tag, ch.attrib.get("name")
        if tag in PLUGIN_TAGS:
            key = name or ({"rfilter": "rfilter", "film": "film", "sampler": "sampler", "bsdf": "bsdf", "emitter": "emitter"}.get(tag)) or ch.attrib.get("id")
            if key is None or key in out:
                counter[0] += 1; key = "_arg_%d" % counter[0]
            out[key] = _plugin(ch, base_dir, ids, counter)
        elif tag == "ref":
            rid = ch.attrib["id"]
            if rid not in ids:
                raise _err('reference to unknown id "%s"' % rid)
            key = name or {"bsdf": "bsdf", "emitter": "emitter", "texture": "reflectance"}.get(ids[rid], rid)
            out[key] = {"type": "ref", "id": rid}
        elif name is None:
            raise _err("<%s> inside <%s> needs a name" % (tag, el.tag))
        elif tag == "float":
            out[name] = float(ch.attrib["value"])
        elif tag == "integer":
            out[name] = int(ch.attrib["value"])
        elif tag == "boolean":
            out[name] = ch.attrib["value"].strip().lower() == "true"
        elif tag == "string":
            v = ch.attrib["value"]
            out[name] = os.path.join(base_dir, v) if name == "filename" and not os.path.isabs(v) else v
        elif tag in ("rgb", "spectrum"):
            v = _floats(ch.attrib["value"])
            if tag == "spectrum" and len(v) != 1:
                raise _err("<spectrum> with wavelength:value pairs is outside this build (constant spectra and <rgb> only)")
            out[name] = {"type": "rgb", "value": v * 3 if len(v) == 1 else v}
        elif tag in ("vector", "point"):
            out[name] = _vec3(ch)
        elif tag == "transform":
            out[name] = _transform(ch)
        else:
            raise _err("XML element <%s> is not supported" % tag)
    return out


def parse_file(filename, **overrides):
    """Returns the scene dictionary of `filename` (what mi.load_dict takes)."""
    text = open(filename).read()
    root = ET.fromstring(text)
    if root.tag != "scene":
        raise _err("the root element must be <scene>")
    params = {}
    for el in root.iter("default"):
        params[el.attrib["name"]] = el.attrib["value"]
    params.update({k: str(v) for k, v in overrides.items()})

    def subst(el):
        for k, v in list(el.attrib.items()):
            if "$" in v:
                def rep(m):
                    if m.group(1) not in params:
                        raise _err('undefined parameter "$%s"' % m.group(1))
                    return params[m.group(1)]
                el.attrib[k] = re.sub(r"\$(\w+)", rep, v)
        for ch in el:
            subst(ch)
    subst(root)
    ids = {el.attrib["id"]: el.tag for el in root.iter() if "id" in el.attrib and el.tag in PLUGIN_TAGS}
    base_dir = os.path.dirname(os.path.abspath(filename))
    scene, counter = {"type": "scene"}, [0]
    for el in root:
        if el.tag == "default":
            continue
        if el.tag not in PLUGIN_TAGS:
            raise _err("XML element <%s> is not supported at scene level" % el.tag)
        key = el.attrib.get("id") or {"integrator": "integrator", "sensor": "sensor"}.get(el.tag)
        if key is None or key in scene:
            counter[0] += 1; key = "_arg_%d" % counter[0]
        scene[key] = _plugin(el, base_dir, ids, counter)
    return scene


def load_file(filename, **overrides):
    from .scene import load_dict
    return load_dict(parse_file(filename, **overrides))
