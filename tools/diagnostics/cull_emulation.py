# CPU emulation (numpy, float64 geometry) of culling certificates for SECONDARY rays: which fraction of (ray, cell) pairs can be certified
# "the reference's edge test rejects every triangle of the cell", per ray and per packet, for cells of 40 (quad) or 10 (tile) triangles.
# Certificate A = DESIGN 3.3 (plane-intersection form, needs |d.n| guard).  Certificate B = moment form:
#   F_k = e_k.w - (d.N)/3,  w = d x (o - G)  (G centroid)   =>   min_k F_k <= -mu/2 + max(0,-d.N)/3,  mu = max_k |e_k.w| >= h_min |w_p|
#   (w_p = component of w in the triangle's plane; |d.n^| <= |d| |w_p| / |w|)
# usage: python tools/diagnostics/cull_emulation.py C2 1 [cell_tris]
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import raytracer_glsl_amd as rt

def morton_cells(v, cell):
    n = v.shape[0]
    cen = v.mean(axis=1); lo, hi = cen.min(0), cen.max(0); ext = (hi - lo).max()
    q = np.minimum(1023, ((cen - lo) / ext * 1023)).astype(np.uint64)
    def spread(x):
        x = x & 0x3ff; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249; return x
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = np.lexsort((np.arange(n), code))
    return v[order]

def kd_order(v, leaf):
    """recursive median split on the longest axis of the centroid box; split positions are multiples of `leaf` so every leaf but the last is full"""
    cen = v.mean(axis=1)
    out = []
    def rec(idx):
        n = len(idx)
        if n <= leaf: out.append(idx); return
        c = cen[idx]; ax = int(np.argmax(c.max(0) - c.min(0)))
        srt = idx[np.argsort(c[:, ax], kind="stable")]
        nl = ((n // 2 + leaf - 1) // leaf) * leaf
        rec(srt[:nl]); rec(srt[nl:])
    rec(np.arange(len(v)))
    return v[np.concatenate(out)]

class Cells:
    def __init__(self, vq, cell):
        n = vq.shape[0]; self.cell = cell; self.nq = nq = (n + cell - 1) // cell
        pad = nq * cell - n
        if pad: vq = np.concatenate([vq, np.repeat(vq[-1:], pad, 0)])
        t = vq.reshape(nq, cell, 3, 3)
        self.tri = t
        G = t.mean(2)                                   # centroids (nq, cell, 3)
        p = t.reshape(nq, -1, 3)
        self.C = 0.5 * (p.min(1) + p.max(1))
        self.R = np.linalg.norm(p - self.C[:, None], axis=2).max(1)             # vertex ball
        self.Rc = np.linalg.norm(G - self.C[:, None], axis=2).max(1)            # centroid ball
        e0, e1, e2 = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 0] - t[:, :, 2]
        N = np.cross(e0, -e2); nn = np.linalg.norm(N, axis=2); nh = N / nn[..., None]
        self.nh = nh
        l = np.stack([np.linalg.norm(e, axis=2) for e in (e0, e1, e2)], 2)
        self.lmax = l.max((1, 2)); self.hmin = (nn / l.max(2)).min(1); self.Nmax = nn.max(1); self.Nmin = nn.min(1)
        a = nh.mean(1); a /= np.linalg.norm(a, axis=1, keepdims=True)
        self.a = a; cosn = np.einsum('qtk,qk->qt', nh, a).min(1); self.cosn = np.clip(cosn, -1, 1); self.sinn = np.sqrt(1 - self.cosn ** 2)
        self.nlo = nh.min(1); self.nhi = nh.max(1)
        c0 = -(e0 * e2).sum(2) / (l[..., 0] * l[..., 2]); c1 = -(e1 * e0).sum(2) / (l[..., 1] * l[..., 0]); c2 = -(e2 * e1).sum(2) / (l[..., 2] * l[..., 1])
        s = np.sqrt(np.maximum(0, 0.5 * (1 - np.maximum(c0, np.maximum(c1, c2)))))
        self.shape = (s * l.min(2) / nn).min(1)
        an = np.linalg.norm(t, axis=3)
        self.E = l.max((1, 2)); self.Pw = np.maximum(an[..., 0] * an[..., 1], np.maximum(an[..., 1] * an[..., 2], an[..., 2] * an[..., 0])).max(1)

def noise(c, on, dn=1.0):          # the reference's own rounding: 2^-20 (E|o| + Pw)|d|
    return 9.5367431640625e-07 * (c.E * on + c.Pw) * dn * 1.01

def cert_A(c, o, d):               # per ray (sigma = 0, ro = 0); d unit
    wv = c.C - o; L = np.linalg.norm(wv, axis=1)
    delta = np.linalg.norm(np.cross(wv, d), axis=1) - c.R - 1e-5 * (L + c.R)
    plo = np.minimum(d * c.nlo, d * c.nhi).sum(1); phi = np.maximum(d * c.nlo, d * c.nhi).sum(1)
    cmin = np.where(plo > 0, plo, np.where(phi < 0, -phi, -1.0)) - 1e-5
    return (delta > 0) & (cmin > 0) & (c.Nmin * cmin * np.minimum(0.3333, delta * c.shape) * 0.99 >= noise(c, np.linalg.norm(o)))

def cert_B(c, o, d, ro=0.0, sig=0.0):   # moment form; packet: every origin within ro of o, every unit direction within sig of d
    g = o - c.C
    w = np.cross(d, g)                                            # moment of the axis line about the cell centre
    gl = np.linalg.norm(g, axis=1)
    slack = ro + sig * (gl + ro) + c.Rc                           # |w_T(line) - w| <= |d x dg| + |dd x g| ...
    wn = np.linalg.norm(w, axis=1)
    wa = np.abs((w * c.a).sum(1))
    wp = np.sqrt(np.maximum(0.0, wn * wn - wa * wa)) * c.cosn - wa * c.sinn - slack      # >= |w_T,p| for every triangle, every line
    dist = wn - slack                                            # >= distance of every line to every centroid
    ok = (wp > 0) & (dist > 1.3334 * c.lmax)
    with np.errstate(divide='ignore', invalid='ignore'):
        lhs = c.hmin * wp * (0.5 - c.lmax / (3.0 * dist))
    return ok & (lhs * 0.99 >= noise(c, np.linalg.norm(o) + ro))

def truth_needed(c, o, d):         # cells holding a triangle whose exact min_k F_k > -noise (what must never be skipped)
    t = c.tri
    cv = np.cross(d, o)
    need = np.zeros(c.nq, bool)
    mn = np.full(t.shape[:2], np.inf)
    for k in range(3):
        a = (k + 1) % 3
        F = ((t[:, :, a] - t[:, :, k]) * cv).sum(2) + (np.cross(t[:, :, a], t[:, :, k]) * d).sum(2)
        mn = np.minimum(mn, F)
    return (mn > -noise(c, np.linalg.norm(o))[:, None]).any(1)

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"; bounce = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cell = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    scene = rt.scenes.CONFIGS[name]["scene"]()
    v = scene.vertices.reshape(-1, 3, 4)[:, :, :3].astype(np.float64)
    order = sys.argv[4] if len(sys.argv) > 4 else "morton"
    c = Cells(morton_cells(v, cell) if order == "morton" else kd_order(v, 10), cell)
    print(name, "cells of", cell, ":", c.nq, "| normal cone half-angle deg: median", np.degrees(np.arccos(c.cosn)).round(1).__getitem__(c.nq // 2) if False else np.median(np.degrees(np.arccos(c.cosn))).round(1),
          "max", np.degrees(np.arccos(c.cosn)).max().round(1), "| R median", np.median(c.R).round(2), "hmin", c.hmin.min().round(3), "lmax", c.lmax.max().round(3), "noise", noise(c, 40.0).max())
    rays = np.load(f"/tmp/rays/{name}_b{bounce}.npy"); rays = rays[~np.isnan(rays[:, 0])].astype(np.float64)
    rng = np.random.default_rng(0)
    pick = rng.choice(len(rays), 1500, replace=False)
    tot = kA = kB = kAB = need = viol = 0
    kept = []
    for i in pick:
        o, d = rays[i, :3], rays[i, 3:]; dl = np.linalg.norm(d)
        if not (dl > 0): continue
        d = d / dl
        A = cert_A(c, o, d); B = cert_B(c, o, d); T = truth_needed(c, o, d)
        viol += int((T & (A | B)).sum())
        tot += c.nq; kA += A.sum(); kB += B.sum(); kAB += (A | B).sum(); need += T.sum(); kept.append(c.nq - (A | B).sum())
    print(f"per ray: certified A {kA / tot:.3f}  B {kB / tot:.3f}  A|B {kAB / tot:.3f}   truly needed {need / tot:.4f}   violations {viol}   cells kept per ray: mean {np.mean(kept):.1f} median {np.median(kept)} p90 {np.percentile(kept, 90)}")

# ---- packets: rays sorted by a key, consecutive 128 = one granule; bounding origin sphere + direction cone; certificate A|B per (granule, cell)
def oct_bin(d, nb):
    """octahedral map of unit directions to nb x nb bins"""
    s = np.abs(d).sum(1, keepdims=True); p = d / s
    u, v = p[:, 0].copy(), p[:, 1].copy()
    neg = p[:, 2] < 0
    uu = (1 - np.abs(v)) * np.sign(u + (u == 0)); vv = (1 - np.abs(u)) * np.sign(v + (v == 0))
    u[neg] = uu[neg]; v[neg] = vv[neg]
    iu = np.minimum(nb - 1, ((u * 0.5 + 0.5) * nb).astype(np.int64)); iv = np.minimum(nb - 1, ((v * 0.5 + 0.5) * nb).astype(np.int64))
    return iu * nb + iv

def morton2(ix, iy):
    def sp(x):
        x = x & 0xffff; x = (x | (x << 8)) & 0x00ff00ff; x = (x | (x << 4)) & 0x0f0f0f0f; x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555; return x
    return sp(ix) | (sp(iy) << 1)

def cert_A_packet(c, O, D, ro, sig, On):
    wv = c.C - O; L = np.linalg.norm(wv, axis=1)
    delta = (np.linalg.norm(np.cross(wv, D), axis=1) - L * sig) - (ro + c.R) - 1e-5 * (L + ro + c.R)
    plo = np.minimum(D * c.nlo, D * c.nhi).sum(1); phi = np.maximum(D * c.nlo, D * c.nhi).sum(1)
    cmin = np.where(plo > 0, plo, np.where(phi < 0, -phi, -1.0)) - sig - 1e-5
    return (delta > 0) & (cmin > 0) & (c.Nmin * cmin * np.minimum(0.3333, delta * c.shape) * 0.99 >= noise(c, On))

def packets(c, rays, key, n_sample=300, gran=128, rng=None):
    order = np.argsort(key, kind="stable"); r = rays[order]
    ng = len(r) // gran
    pick = rng.choice(ng, min(n_sample, ng), replace=False)
    tot = cert = 0; ros = []; sigs = []; union_need = 0
    for g in pick:
        rr = r[g * gran:(g + 1) * gran]; o = rr[:, :3]; d = rr[:, 3:] / np.linalg.norm(rr[:, 3:], axis=1, keepdims=True)
        O = 0.5 * (o.min(0) + o.max(0)); D = 0.5 * (d.min(0) + d.max(0)); Dl = np.linalg.norm(D)
        if Dl < 0.25: tot += c.nq; ros.append(np.nan); sigs.append(2.0); continue
        D = D / Dl
        ro = np.linalg.norm(o - O, axis=1).max(); sig = np.linalg.norm(d - D, axis=1).max(); On = np.linalg.norm(O) + ro
        ok = cert_A_packet(c, O, D, ro, sig, On) | cert_B(c, O, D, ro, sig)
        tot += c.nq; cert += ok.sum(); ros.append(ro); sigs.append(sig)
        # what a union of exact per-ray certificates would keep (upper bound of any packet scheme with this sorting)
        need = np.zeros(c.nq, bool)
        for i in range(0, gran, 8): need |= ~(cert_A(c, o[i], d[i]) | cert_B(c, o[i], d[i]))
        union_need += need.sum()
    return cert / tot, np.nanmedian(ros), np.median(sigs), 1 - union_need / tot

if __name__ == "__main__":
    dirs = rays[:, 3:] / np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    ok = np.isfinite(dirs).all(1); rays = rays[ok]; dirs = dirs[ok]
    mult = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # emulate a frame batch: the same rays, jittered, mult times
    if mult > 1:
        rr = [rays]
        for m in range(1, mult):
            j = rays.copy(); perm = rng.permutation(len(rays)); j[:, 3:] = rays[perm, 3:] * 1.0   # same origins, directions of other rays: NOT physical; only for density
            rr.append(j)
        # physical alternative: load more frames if they were dumped
    for nb in (4, 8, 16):
        for ocell in (0.5, 1.0, 2.0, 4.0):
            db = oct_bin(dirs, nb)
            ix = np.clip(((rays[:, 0] + 40) / ocell).astype(np.int64), 0, 65535); iy = np.clip(((rays[:, 1] + 40) / ocell).astype(np.int64), 0, 65535)
            iz = np.clip(((rays[:, 2] + 40) / (4 * ocell)).astype(np.int64), 0, 255)
            ocode = morton2(ix, iy) * 256 + iz
            # origin cell major, direction bin minor, then fine origin
            key = (ocode.astype(np.int64) << 20) | (db << 8)
            frac, ro, sig, ub = packets(c, rays, key, rng=rng)
            print(f"dir bins {nb}x{nb} origin cell {ocell}: certified {frac:.3f}  (median ro {ro:.2f} sigma {sig:.2f})  union of per-ray certificates {ub:.3f}")
