"""An independent witness of the shading arithmetic — TEST INFRASTRUCTURE.

The CPU oracle (oracle/) and the HIP kernels share include/rtr_math.h: the same Moeller-Trumbore, the same pow / ACES / UNORM8
forms, the same expression order.  That is what makes bit-exact parity testable, and it is also why a wrong constant in that
header would be invisible to every parity test.  This module restates the path a SECOND time with nothing in common:
numpy, float64, libm (np.power, np.arctan2, ...), its own ray-triangle test, brute force over all triangles (no BVH), its own
texture sampler and tone map, written from the shader text (reference src/shaders/raygen.rgen, closesthit.rchit, miss.rmiss,
opacity.rahit, cook-torrance.glsl, raycommon.glsl) and the C-ABI scene description alone.

It cannot be bit-exact against fp32 code; tests/test_witness.py holds the oracle within 1e-5 relative on the HDR radiance and
+-1 LSB on the RGBA8 bytes, on a sample of pixels, and allows the handful of pixels where a discrete decision (hit / miss at a
silhouette, occluded / visible at a shadow edge, r1 + r2 > 1 fold) flips between fp32 and fp64.
"""
import ctypes as C

import numpy as np

F = np.float64


def _arr(ptr, n, dtype, width):
    if n == 0:
        return np.zeros((0, width), dtype)
    buf = C.cast(ptr, C.POINTER(C.c_uint8 * (n * width * np.dtype(dtype).itemsize))).contents
    return np.frombuffer(buf, dtype=dtype).reshape(n, width).copy()


def pcg(seed):
    """raycommon.glsl:22-27 on uint32 arrays (numpy wraps modulo 2^32)."""
    with np.errstate(over="ignore"):
        seed = np.asarray(seed, dtype=np.uint32)
        state = seed * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
        h = (word >> np.uint32(22)) ^ word
    # float(h) / 2^32 as the shader computes it: the uint -> float conversion rounds to 24 bits
    return h.astype(np.float32).astype(F) / 4294967296.0


def normalize(v):
    return v / np.sqrt(np.sum(v * v, axis=-1, keepdims=True))


class Witness:
    def __init__(self, desc):
        d = desc
        self.verts = _arr(d.vertices, d.numVertices, np.float32, 12).astype(F)        # pos(3) pad normal(3) pad uv(2) pad(2)
        self.idx = _arr(d.indices, d.numIndices, np.uint32, 1).reshape(-1)
        self.numLights = d.numLights
        self.meshes = [d.meshes[i] for i in range(d.numMeshes)]
        self.instances = sorted([d.instances[i] for i in range(d.numInstances)], key=lambda i: i.customIndex)
        self.objects = [d.objects[i] for i in range(d.numObjects)]
        self.lights = [d.lights[i] for i in range(d.numLights)]
        self.textures = []
        for t in range(d.numTextures):
            tx = d.textures[t]
            if not tx.pixels:
                self.textures.append(None)
                continue
            px = np.frombuffer(C.cast(tx.pixels, C.POINTER(C.c_uint8 * (tx.width * tx.height * tx.channels))).contents, dtype=np.uint8)
            self.textures.append(px.reshape(tx.height, tx.width, tx.channels).astype(F) / 255.0)
        self.hdri = None
        if d.hdri:
            tx = d.hdri.contents
            px = np.frombuffer(C.cast(tx.pixels, C.POINTER(C.c_uint8 * (tx.width * tx.height * tx.channels))).contents, dtype=np.uint8)
            self.hdri = px.reshape(tx.height, tx.width, tx.channels).astype(F) / 255.0
        self.sky = np.power(np.array([d.skyColor[0], d.skyColor[1], d.skyColor[2]], F), 2.2)
        # world-space triangle soup in (instance, primitive) order + what the hit shader needs per triangle
        V0, V1, V2, cust, prim, alpha = [], [], [], [], [], []
        self.xform, self.nmat = {}, {}
        for inst in self.instances:
            me = self.meshes[inst.meshIndex]
            M = np.array(inst.transform[:], F).reshape(3, 4)
            self.xform[inst.customIndex] = M
            self.nmat[inst.customIndex] = np.linalg.inv(M[:, :3]).T                     # transpose(inverse(mat3(O2W))), closesthit.rchit:74
            tri = self.idx[me.indexOffset:me.indexOffset + me.indexCount].reshape(-1, 3).astype(np.int64) + me.vertexOffset
            P = self.verts[:, 0:3]
            w = [P[tri[:, k]] @ M[:, :3].T + M[:, 3] for k in range(3)]
            V0.append(w[0]); V1.append(w[1]); V2.append(w[2])
            cust.append(np.full(len(tri), inst.customIndex, np.int64)); prim.append(np.arange(len(tri), dtype=np.int64))
            tested = inst.customIndex >= self.numLights and self.objects[inst.customIndex - self.numLights].usesOpacityMap != 0 and me.isOpaque == 0
            alpha.append(np.full(len(tri), tested))
        cat = lambda xs, shape: np.concatenate(xs) if xs else np.zeros(shape)       # noqa: E731
        self.V0, self.V1, self.V2 = cat(V0, (0, 3)), cat(V1, (0, 3)), cat(V2, (0, 3))
        self.cust, self.prim, self.alpha = cat(cust, (0,)).astype(np.int64), cat(prim, (0,)).astype(np.int64), cat(alpha, (0,)).astype(bool)

    # ---- texture(): linear filter, repeat addressing, one mip (image_sampler.cppm:26-42) ------------------------------------
    @staticmethod
    def sample(img, u, v):
        H, W = img.shape[:2]
        x, y = (u - np.floor(u)) * W - 0.5, (v - np.floor(v)) * H - 0.5
        x0, y0 = np.floor(x), np.floor(y)
        fx, fy = (x - x0)[:, None], (y - y0)[:, None]
        x0, y0 = x0.astype(np.int64) % W, y0.astype(np.int64) % H
        x1, y1 = (x0 + 1) % W, (y0 + 1) % H
        a = img[y0, x0] * (1 - fx) + img[y0, x1] * fx
        b = img[y1, x0] * (1 - fx) + img[y1, x1] * fx
        out = a * (1 - fy) + b * fy
        if out.shape[1] == 1:                                                           # R8: (r, 0, 0, 1)
            out = np.concatenate([out, np.zeros_like(out), np.zeros_like(out), np.ones_like(out)], axis=1)
        return out

    # ---- ray / triangle, all rays against all triangles -------------------------------------------------------------------
    def _candidates(self, o, d, tmin, tmax):
        """(R, T) arrays t, u, v with t = inf where the ray misses the triangle (edge rule of intersect.rint:18-41)."""
        e1, e2 = self.V1 - self.V0, self.V2 - self.V0                                    # (T, 3)
        h = np.cross(d[:, None, :], e2[None, :, :])                                     # (R, T, 3)
        a = np.einsum("tk,rtk->rt", e1, h)
        with np.errstate(divide="ignore", invalid="ignore"):
            f = 1.0 / a
            s = o[:, None, :] - self.V0[None, :, :]
            u = f * np.einsum("rtk,rtk->rt", s, h)
            q = np.cross(s, e1[None, :, :])
            v = f * np.einsum("rk,rtk->rt", d, q)
            t = f * np.einsum("tk,rtk->rt", e2, q)
        ok = (np.abs(a) >= 1e-5) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax[:, None])
        return np.where(ok, t, np.inf), u, v

    def _alpha_reject(self, t, u, v):
        """opacity.rahit:31-64 for candidates on alpha-tested triangles: candidates whose opacity texel .r < 0.9 are ignored."""
        if not self.alpha.any():
            return t
        t = t.copy()
        for ti in np.nonzero(self.alpha)[0]:
            rows = np.nonzero(np.isfinite(t[:, ti]))[0]
            if len(rows) == 0:
                continue
            oi = self.objects[self.cust[ti] - self.numLights]
            tri = self.idx[oi.indexOffset + 3 * self.prim[ti]: oi.indexOffset + 3 * self.prim[ti] + 3].astype(np.int64) + oi.vertexOffset
            uv = self.verts[tri, 8:10]
            bu, bv = u[rows, ti], v[rows, ti]
            uu = uv[0, 0] * (1 - bu - bv) + uv[1, 0] * bu + uv[2, 0] * bv
            vv = uv[0, 1] * (1 - bu - bv) + uv[1, 1] * bu + uv[2, 1] * bv
            texel = self.sample(self.textures[oi.opacityIndex], uu, vv)
            t[rows[texel[:, 0] < 0.9], ti] = np.inf
        return t

    def closest(self, o, d, tmax):
        t, u, v = self._candidates(o, d, 0.001, np.full(len(o), tmax))
        t = self._alpha_reject(t, u, v)
        k = np.argmin(t, axis=1)                     # ties: the first in (instance, primitive) order = min (customIndex, primitiveID)
        r = np.arange(len(o))
        return np.isfinite(t[r, k]), k, t[r, k], u[r, k], v[r, k]

    def occluded(self, o, d, tmax):
        live = tmax > 0.001
        t, u, v = self._candidates(o, d, 0.001, tmax)
        t = self._alpha_reject(t, u, v)
        return live & np.isfinite(t).any(axis=1)

    # ---- cook-torrance.glsl ------------------------------------------------------------------------------------------------
    @staticmethod
    def ggx_d(n, h, alpha):
        noh = np.sum(n * h, -1)
        a2, noh2 = alpha * alpha, noh * noh
        den = np.maximum(noh2 * a2 + (1 - noh2), 0.001)
        return (noh > 0) * a2 / (np.pi * den * den)

    @staticmethod
    def ggx_g1(v, n, h, alpha):
        voh = np.clip(np.sum(v * h, -1), 0.001, 1.0)
        chi = (voh / np.clip(np.sum(v * n, -1), 0.001, 1.0)) > 0
        voh2 = voh * voh
        tan2 = (1 - voh2) / voh2
        return chi * 2 / (1 + np.sqrt(1 + alpha * alpha * tan2))

    def brdf(self, N, V, Ldir, color, metallic, roughness, mspec, ndotv_floor, ndotl_floor):
        H = normalize(V + Ldir)
        cosT = np.clip(np.sum(V * H, -1), 0, 1)
        D = self.ggx_d(N, H, roughness)
        G = self.ggx_g1(V, N, H, roughness) * self.ggx_g1(Ldir, N, H, roughness)
        Fr = mspec + (1 - mspec) * np.power(1 - cosT, 5.0)[:, None]
        ndv = np.maximum(np.sum(N * V, -1), ndotv_floor)
        ndl = np.maximum(np.sum(N * Ldir, -1), ndotl_floor)
        spec = (D * G)[:, None] * Fr / (4 * ndv * ndl)[:, None]
        diff = (1 - metallic)[:, None] * color / np.pi
        return spec + diff, ndl

    # ---- raygen.rgen for a set of pixels -------------------------------------------------------------------------------------
    def render(self, camera, info, params, xs, ys):
        """Returns (hdr (n,3) float64: pre-tonemap shadowed radiance, rgba8 (n,) uint32 packed B,G,R,255) for pixels (xs, ys)."""
        xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
        n = len(xs)
        cam = np.array(camera.position[:3], F)
        TL, dH, dV = (np.array(a[:3], F) for a in (camera.topLeftViewportCorner, camera.horizontalViewportDelta, camera.verticalViewportDelta))
        shadowed = np.zeros((n, 3), F)
        ux, uy = xs.astype(np.uint32), ys.astype(np.uint32)
        for i in range(params.spp):
            with np.errstate(over="ignore"):
                jx, jy = pcg(ux + np.uint32(i)), pcg(ux + np.uint32(i) * np.uint32(322))     # raygen.rgen:83: both seeds use x only
            pw = TL + dH * (xs + jx - 0.5)[:, None] + dV * (ys + jy - 0.5)[:, None]
            rd = normalize(pw - cam)
            hit, k, t, bu, bv = self.closest(np.broadcast_to(cam, (n, 3)), rd, 10000.0)
            # miss.rmiss:15-27
            miss = ~hit
            if miss.any():
                sky = np.broadcast_to(self.sky, (miss.sum(), 3))
                if self.hdri is not None:
                    dd = rd[miss]
                    hu = np.arctan2(dd[:, 2], dd[:, 0]) / (2 * 3.14159265) + 0.5
                    hv = 1.0 - np.arccos(np.clip(dd[:, 1], -1, 1)) / 3.14159265
                    sky = np.power(self.sample(self.hdri, hu, hv)[:, :3], 2.2)
                shadowed[miss] += sky
            cu = self.cust[k]
            islight = hit & (cu < self.numLights)
            for li in np.unique(cu[islight]):
                shadowed[islight & (cu == li)] += np.array(self.lights[li].color[:3], F)
            surf = np.nonzero(hit & (cu >= self.numLights))[0]
            if len(surf) == 0:
                continue
            shadowed[surf] += self._shade(surf, xs, ys, cam, rd, k, bu, bv, info, params)
        shadowed /= params.spp
        return shadowed, self.pack(shadowed)

    def _shade(self, rows, xs, ys, cam, rd, k, bu, bv, info, params):
        m = len(rows)
        out = np.zeros((m, 3), F)
        kk, u, v = k[rows], bu[rows], bv[rows]
        b0 = 1 - u - v
        P = np.zeros((m, 3), F); N = np.zeros((m, 3), F)
        color = np.zeros((m, 3), F); metallic = np.zeros(m, F); rough = np.zeros(m, F)
        for ci in np.unique(self.cust[kk]):                                             # closesthit.rchit:53-106, object by object
            sel = np.nonzero(self.cust[kk] == ci)[0]
            oi = self.objects[ci - self.numLights]
            tri = self.idx[(oi.indexOffset + 3 * self.prim[kk[sel]])[:, None] + np.arange(3)].astype(np.int64) + oi.vertexOffset
            vv = self.verts[tri]                                                        # (s, 3, 12)
            w = np.stack([b0[sel], u[sel], v[sel]], 1)[:, :, None]
            lp = np.sum(vv[:, :, 0:3] * w, 1)
            M = self.xform[ci]
            P[sel] = lp @ M[:, :3].T + M[:, 3]
            ns = np.sum(vv[:, :, 4:7] * w, 1)
            zero = np.sum(ns * ns, 1) <= 0
            g = np.cross(vv[:, 1, 0:3] - vv[:, 0, 0:3], vv[:, 2, 0:3] - vv[:, 0, 0:3])     # D1: geometric normal, facing the ray
            ns = np.where(zero[:, None], g, ns)
            nn = normalize(normalize(ns) @ self.nmat[ci].T)
            flip = zero & (np.sum(nn * rd[rows][sel], 1) > 0)
            N[sel] = np.where(flip[:, None], -nn, nn)
            uvw = np.sum(vv[:, :, 8:10] * w, 1)
            col = np.broadcast_to(np.array(oi.color[:3], F), (len(sel), 3)).copy()
            r_, me_ = np.full(len(sel), oi.specular, F), np.full(len(sel), oi.metallic, F)
            if oi.usesColorMap:
                col = self.sample(self.textures[oi.colorIndex], uvw[:, 0], uvw[:, 1])[:, :3]
            if oi.usesSpecularMap:
                r_ = self.sample(self.textures[oi.specularIndex], uvw[:, 0], uvw[:, 1])[:, 0]
            if oi.usesMetallicMap:
                me_ = self.sample(self.textures[oi.metallicIndex], uvw[:, 0], uvw[:, 1])[:, 0]
            color[sel] = np.power(col, 2.2); metallic[sel] = me_; rough[sel] = 1.0 - r_
        V = normalize(cam - P)
        mspec = 0.04 * (1 - metallic)[:, None] + color * metallic[:, None]              # mix(vec3(0.04), color, metallic)
        so = P + N * 0.01
        px, py = xs[rows].astype(np.uint32), ys[rows].astype(np.uint32)
        ns_ = params.numShadowRays
        for li in range(info.numAreaLights):                                            # raygen.rgen:165-285
            L = self.lights[li]
            T = np.array(L.transform[:], F).reshape(4, 4).T                             # column-major mat4
            lcol = np.array(L.color[:3], F)
            for ti in range(L.numTriangles):
                j = self.idx[L.indexOffset + 3 * ti: L.indexOffset + 3 * ti + 3].astype(np.int64) + L.vertexOffset
                Pl = self.verts[j, 0:3] @ T[:3, :3].T + T[:3, 3]
                ln = np.cross(Pl[2] - Pl[1], Pl[0] - Pl[1])
                area = np.sqrt(ln @ ln) * 0.5
                pdf = 1.0 / (area * 0.7)
                ln = ln / np.sqrt(ln @ ln)
                front = np.ones(m, bool) if L.isTwoSided else ((P - Pl[0]) @ ln >= 0)
                acc = np.zeros((m, 3), F)
                for s in range(ns_):
                    with np.errstate(over="ignore"):
                        seed = np.uint32(s) + px * np.uint32(733) + py * np.uint32(1933) + np.uint32(info.frame)
                        r1, r2 = pcg(seed), pcg(seed + np.uint32(100))
                    # the fold is decided on the fp32 sum as the shader does it (r1, r2 are fp32 values)
                    fold = (r1.astype(np.float32) + r2.astype(np.float32)) > np.float32(1.0)
                    r1, r2 = np.where(fold, 1 - r1, r1), np.where(fold, 1 - r2, r2)
                    lp = Pl[0] + (Pl[1] - Pl[0]) * r1[:, None] + (Pl[2] - Pl[0]) * r2[:, None]
                    lv = lp - P
                    dist = np.sqrt(np.sum(lv * lv, 1))
                    ld = lv / dist[:, None]
                    vis = ~self.occluded(so, ld, dist - 0.5)
                    B, ndl = self.brdf(N, V, ld, color, metallic, rough, mspec, 0.1, 0.1)
                    Lr = lcol[None, :] * (L.intensity * ndl / (dist * dist) * 10.0)[:, None]
                    acc += vis[:, None] * B * Lr / pdf
                out += np.where(front[:, None], acc / ns_, 0.0)
        dl = np.array([-1.0, 1.0, -0.5], F); dl /= np.sqrt(dl @ dl)                    # raygen.rgen:289-338
        lit = N @ dl > 0
        if lit.any():
            r = np.nonzero(lit)[0]
            dd = np.broadcast_to(dl, (len(r), 3))
            vis = ~self.occluded(so[r], dd, np.full(len(r), 10000.0))
            B, ndl = self.brdf(N[r], V[r], dd, color[r], metallic[r], rough[r], mspec[r], 5.0, 0.0001)
            out[r] += vis[:, None] * B * (np.array([1.0, 1.0, 0.5], F) * 0.2)[None, :] * (ndl * 20.0)[:, None]
        return out

    @staticmethod
    def pack(c):
        """ACESFilm, pow(1/2.2), imageStore(vec4(b, g, r, 1)) into rgba8 (raygen.rgen:45-59,345-357)."""
        x = np.asarray(c, F)
        with np.errstate(invalid="ignore", divide="ignore"):
            a = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0, 1)
        a = np.nan_to_num(a, nan=0.0)
        q = np.rint(np.power(a, 1 / 2.2) * 255.0).astype(np.uint32)
        return q[:, 2] | (q[:, 1] << 8) | (q[:, 0] << 16) | np.uint32(0xff000000)
