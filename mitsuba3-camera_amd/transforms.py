"""float32 affine / projective transforms with Mitsuba's Transform4f semantics.

Mirrors include/mitsuba/core/transform.h: a transform carries its matrix AND the
inverse-transpose, products multiply both analytically (transform.h:62-71), chained
Python methods right-multiply (src/core/python/transform_v.cpp:212-219), `rotate` takes
degrees (transform.h:188-191), `look_at` builds (left, up', dir, origin) columns
(transform.h:254-282).
"""
import math
import numpy as np

f32 = np.float32


def _m(a):
    return np.asarray(a, dtype=np.float32).reshape(4, 4)


def _mul(a, b):
    # float32 product, accumulated in float32 like dr::Matrix operator*
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)


class Transform4f:
    __slots__ = ("matrix", "inverse_transpose")

    def __init__(self, matrix=None, inverse_transpose=None):
        if matrix is None:
            matrix = np.eye(4, dtype=np.float32)
        if isinstance(matrix, Transform4f):
            self.matrix = matrix.matrix.copy()
            self.inverse_transpose = matrix.inverse_transpose.copy()
            return
        self.matrix = _m(matrix)
        if inverse_transpose is None:
            inverse_transpose = np.linalg.inv(self.matrix.astype(np.float64)).T.astype(np.float32)
        self.inverse_transpose = _m(inverse_transpose)

    # ---- constructors (static + chained forms) ---------------------------------------------
    @staticmethod
    def _translate(v):
        v = np.asarray(v, dtype=np.float32).reshape(3)
        m = np.eye(4, dtype=np.float32); m[:3, 3] = v
        i = np.eye(4, dtype=np.float32); i[:3, 3] = -v
        return Transform4f(m, i.T)

    @staticmethod
    def _scale(v):
        v = np.asarray(v, dtype=np.float32)
        if v.ndim == 0:
            v = np.full(3, v, dtype=np.float32)
        v = v.reshape(3)
        m = np.diag(np.append(v, f32(1))).astype(np.float32)
        i = np.diag(np.append(f32(1) / v, f32(1))).astype(np.float32)
        return Transform4f(m, i)

    @staticmethod
    def _rotate(axis, angle):
        a = np.asarray(axis, dtype=np.float64).reshape(3)
        n = np.linalg.norm(a)
        if n != 0 and abs(n - 1.0) > 1e-6:
            a = a / n
        th = math.radians(float(angle))
        s, c = math.sin(th), math.cos(th)
        x, y, z = a
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
                      [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
                      [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0],
                      [0, 0, 0, 1]], dtype=np.float64).astype(np.float32)
        return Transform4f(r, r)        # transform.h:190-191: inverse transpose of a rotation is itself

    @staticmethod
    def _look_at(origin, target, up):
        o = np.asarray(origin, dtype=np.float32); t = np.asarray(target, dtype=np.float32); u = np.asarray(up, dtype=np.float32)

        def nrm(v):
            return (v / np.sqrt(np.dot(v, v))).astype(np.float32)
        d = nrm(t - o)
        left = nrm(np.cross(u, d).astype(np.float32))
        new_up = np.cross(d, left).astype(np.float32)
        m = np.eye(4, dtype=np.float32)
        m[:3, 0] = left; m[:3, 1] = new_up; m[:3, 2] = d; m[:3, 3] = o
        inv = np.eye(4, dtype=np.float32)
        inv[0, :3] = left; inv[1, :3] = new_up; inv[2, :3] = d
        inv[:3, 3] = -(inv[:3, :3] @ o)
        return Transform4f(m, inv.T)

    @staticmethod
    def _perspective(fov, near, far):
        fov, near, far = f32(fov), f32(near), f32(far)
        recip = f32(1) / (far - near)
        tan = f32(math.tan(math.radians(float(fov) * 0.5)))
        cot = f32(1) / tan
        m = np.zeros((4, 4), dtype=np.float32)
        m[0, 0] = cot; m[1, 1] = cot; m[2, 2] = far * recip; m[2, 3] = -near * far * recip; m[3, 2] = 1
        i = np.zeros((4, 4), dtype=np.float32)
        i[0, 0] = tan; i[1, 1] = tan; i[3, 3] = f32(1) / near; i[2, 3] = 1; i[3, 2] = (near - far) / (far * near)
        return Transform4f(m, i.T)

    # Both `T.translate(v)` (static-style on the class instance mi.ScalarTransform4f) and
    # `t.translate(v)` (chained) are spelled the same way in Mitsuba's Python API.
    def translate(self, v):
        return self @ Transform4f._translate(v)

    def scale(self, v):
        return self @ Transform4f._scale(v)

    def rotate(self, axis, angle):
        return self @ Transform4f._rotate(axis, angle)

    def look_at(self, origin, target, up):
        return self @ Transform4f._look_at(origin, target, up)

    def perspective(self, fov, near, far):
        return self @ Transform4f._perspective(fov, near, far)

    # ---- algebra -----------------------------------------------------------------------------
    def __matmul__(self, other):
        if isinstance(other, Transform4f):
            return Transform4f(_mul(self.matrix, other.matrix), _mul(self.inverse_transpose, other.inverse_transpose))
        v = np.asarray(other, dtype=np.float32)
        return self.transform_affine(v)

    def inverse(self):
        return Transform4f(self.inverse_transpose.T.copy(), self.matrix.T.copy())

    def translation(self):
        return self.matrix[:3, 3].copy()

    def transform_affine(self, p):
        """Point transform (transform.h:98-106), float32 fma order is irrelevant at numpy level."""
        p = np.asarray(p, dtype=np.float32)
        return (self.matrix[:3, :3] @ p + self.matrix[:3, 3]).astype(np.float32)

    def transform_vector(self, v):
        return (self.matrix[:3, :3] @ np.asarray(v, dtype=np.float32)).astype(np.float32)

    def transform_normal(self, n):
        return (self.inverse_transpose[:3, :3] @ np.asarray(n, dtype=np.float32)).astype(np.float32)

    def has_scale(self):
        m = self.matrix[:3, :3].astype(np.float64)
        return not np.allclose(m.T @ m, np.eye(3), atol=1e-3)

    def affine12(self):
        return np.ascontiguousarray(self.matrix[:3, :]).reshape(12).astype(np.float32)

    def inverse_affine12(self):
        return np.ascontiguousarray(self.inverse_transpose.T[:3, :]).reshape(12).astype(np.float32)

    def __eq__(self, o):
        return isinstance(o, Transform4f) and np.array_equal(self.matrix, o.matrix)

    def __repr__(self):
        return "Transform4f(%s)" % np.array2string(self.matrix, precision=6)


class _TransformFactory:
    """`T = mi.ScalarTransform4f; T.translate(..).rotate(..)` -- the class object itself acts as identity."""

    def __call__(self, *a, **k):
        return Transform4f(*a, **k)

    def translate(self, v): return Transform4f._translate(v)
    def scale(self, v): return Transform4f._scale(v)
    def rotate(self, axis, angle): return Transform4f._rotate(axis, angle)
    def look_at(self, origin, target, up): return Transform4f._look_at(origin, target, up)
    def perspective(self, fov, near, far): return Transform4f._perspective(fov, near, far)


ScalarTransform4f = _TransformFactory()
