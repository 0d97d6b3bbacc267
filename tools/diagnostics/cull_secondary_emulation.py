# CPU emulation (numpy): which fraction of (ray, cell) pairs does the packet-culling certificate of DESIGN.md 3.3 certify PER RAY (sigma = 0)
# for cosine-distributed rays leaving the mesh surface?  usage: python tools/diagnostics/cull_secondary_emulation.py [C2|C4] [triangles per cell]
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import raytracer_glsl_amd as rt
from test_cull_certificate import cull_record, dot3, cross3, f32
sc = rt.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = sc.CONFIGS[name]; scene = cfg["scene"]()
v = scene.vertices.reshape(-1, 3, 4)[:, :, :3].astype(np.float64)
n = v.shape[0]
cen = v.mean(axis=1); lo, hi = cen.min(0), cen.max(0); ext = (hi - lo).max()
q = np.minimum(1023, ((cen - lo) / ext * 1023)).astype(np.uint64)
def spread(x):
    x = x & 0x3ff; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249; return x
code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
order = np.lexsort((np.arange(n), code)); vq = v[order]
QT = int(sys.argv[2]) if len(sys.argv) > 2 else 40          # triangles per cull cell
nq = (n + QT - 1) // QT
recs = [cull_record(vq[QT * i:QT * i + QT]) for i in range(nq)]
C = np.array([r["c"] for r in recs], np.float64); R = np.array([r["R"] for r in recs]); Nmin = np.array([r["Nmin"] for r in recs]); shape = np.array([r["shape"] for r in recs])
E = np.array([r["E"] for r in recs]); Pw = np.array([r["Pw"] for r in recs])
nlo = np.array([r["nlo"] for r in recs]); nhi = np.array([r["nhi"] for r in recs])
def nrm_of(t):
    N = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]); return N / np.linalg.norm(N, axis=1, keepdims=True)
N16 = [nrm_of(vq[QT * i:QT * i + QT]) for i in range(nq)]
rng = np.random.default_rng(0)
tot = box_ok = tri_ok = near = 0
keep_counts = []
for _ in range(400):
    t = vq[rng.integers(n)]
    w = rng.dirichlet([1, 1, 1]); o = (t * w[:, None]).sum(0)
    nrm = np.cross(t[1] - t[0], t[2] - t[0]); nrm /= np.linalg.norm(nrm)
    if nrm[2] > 0: nrm = -nrm                     # the side facing the camera (z = -35)
    # cosine-weighted direction around nrm
    z = rng.uniform(-1, 1); a = rng.uniform(0, 2 * np.pi); r = np.sqrt(1 - z * z)
    d = nrm + np.array([r * np.cos(a), r * np.sin(a), z]); d /= np.linalg.norm(d)
    wv = C - o; L = np.linalg.norm(wv, axis=1)
    crn = np.linalg.norm(np.cross(wv, d), axis=1)
    delta = crn - R - 1e-5 * (L + R)
    plo = np.minimum(d * nlo, d * nhi).sum(1); phi = np.maximum(d * nlo, d * nhi).sum(1)
    cbox = np.where(plo > 0, plo, np.where(phi < 0, -phi, -1.0)) - 1e-5
    ctri = np.array([np.abs(N @ d).min() for N in N16]) - 2e-3
    On = np.linalg.norm(o)
    rhs = 9.5367431640625e-07 * (E * On + Pw) * 1.01
    def ok(c): return (delta > 0) & (c > 0) & (Nmin * c * np.minimum(0.3333, delta * shape) * 0.99 >= rhs)
    tot += nq; box_ok += ok(cbox).sum(); tri_ok += ok(np.maximum(cbox, ctri)).sum(); near += (delta <= 0).sum()
    keep_counts.append(nq - ok(np.maximum(cbox, ctri)).sum())
print(name, "cells of", QT, ":", nq, "| certified with box guard", round(box_ok / tot, 3), "| with per-triangle guard", round(tri_ok / tot, 3), "| inside sphere reach", round(near / tot, 3),
      "| cells kept per ray: mean", np.mean(keep_counts).round(1), "median", np.median(keep_counts), "p90", np.percentile(keep_counts, 90))
