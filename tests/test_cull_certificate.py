"""The packet-culling certificate (raytracer.glsl_amd/csrc/rt_mfma.hpp `MfCull`) against the reference's own
triangle test, on the CPU.

The HIP path may skip a quad of 40 triangles for a wave of 128 rays only if the reference's edge test
(/root/reference/shaders/raytracer.glsl:226-245: all three `dot(e_k, cv) + dot(m_k, d) > 0`, evaluated in fp32) rejects every
(ray, triangle) pair.  This file restates the certificate in numpy (fp32, the device formulas) and the reference's test in fp32
with the pinned operation order, and checks on random and adversarial inputs that

  * whenever the certificate fires, the reference rejects every pair (including rays placed IN the plane of a far triangle of
    the quad, +- a few ulps: the case a purely geometric "misses the bounding sphere" cull gets wrong);
  * the identity the proof rests on, F_k = -(d.N) beta_k, holds;
  * the certificate is not vacuous (it fires for most far packets on a bumpy height field).

The device implementation itself is checked on the GPU by image parity with the cull on / off / on every bounce
(tests/test_gpu_fullsize.py, test_gpu_golden.py, test_gpu_parity.py).  Parity of the certificate's THEORY is what is tested here.
"""
import numpy as np
import pytest

f32 = np.float32


def dot3(a, b):          # the oracle's order: (z*z + y*y) + x*x, every operation rounded to fp32
    return (a[..., 2] * b[..., 2] + a[..., 1] * b[..., 1]) + a[..., 0] * b[..., 0]


def cross3(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def reference_accepts(tri, o, d):
    """tri (T,3,3), o/d (R,3) fp32 -> (R,T) bool: all three edge tests of :243-245 pass (cv = cross(d, o), m_k = cross(v_a, v_k))."""
    tri, o, d = tri.astype(f32), o.astype(f32), d.astype(f32)
    cv = cross3(d, o)[:, None, :]
    dd = d[:, None, :]
    ok = np.ones((o.shape[0], tri.shape[0]), bool)
    for k in range(3):
        a = (k + 1) % 3
        e = (tri[:, a] - tri[:, k])[None]
        m = cross3(tri[:, a], tri[:, k])[None]
        ok &= -dot3(np.broadcast_to(e, (o.shape[0],) + e.shape[1:]), np.broadcast_to(cv, (o.shape[0], tri.shape[0], 3))) \
            < dot3(np.broadcast_to(m, (o.shape[0],) + m.shape[1:]), np.broadcast_to(dd, (o.shape[0], tri.shape[0], 3)))
    return ok


def cull_record(tri):
    """prepare_cull_kernel: bounding sphere, box of the unit normals, N_min, shape, E, Pw (fp32 with the kernel's slack)."""
    tri = tri.astype(f32)
    p = tri.reshape(-1, 3)
    c = f32(0.5) * p.min(0) + f32(0.5) * p.max(0)
    R = np.sqrt(((p - c) ** 2).sum(1)).max() * f32(1.0001) + f32(1e-30)
    e0, e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 1], tri[:, 0] - tri[:, 2]
    N = cross3(e0, -e2)
    nn = np.sqrt(dot3(N, N))
    nh = N / nn[:, None]
    l0, l1, l2 = (np.sqrt(dot3(e, e)) for e in (e0, e1, e2))
    c0, c1, c2 = -dot3(e0, e2) / (l0 * l2), -dot3(e1, e0) / (l1 * l0), -dot3(e2, e1) / (l2 * l1)
    cmax = np.minimum(f32(1), np.maximum(c0, np.maximum(c1, c2)))
    s = np.sqrt(np.maximum(f32(0), f32(0.5) * (f32(1) - cmax)))
    shape = (f32(0.999) * s * np.minimum(l0, np.minimum(l1, l2)) / nn).min()
    an = [np.sqrt(dot3(tri[:, k], tri[:, k])) for k in range(3)]
    return dict(c=c, R=f32(R), nlo=nh.min(0) - f32(1e-6), nhi=nh.max(0) + f32(1e-6), Nmin=nn.min() * f32(0.999), shape=f32(shape),
                E=np.maximum(l0, np.maximum(l1, l2)).max() * f32(1.001),
                Pw=max((an[0] * an[1]).max(), (an[1] * an[2]).max(), (an[2] * an[0]).max()) * f32(1.001))


def packet_bounds(o, d):
    """packet_cull_kernel: origin sphere, direction cone, |o| bound."""
    o, d = o.astype(f32), d.astype(f32)
    dh = d / np.sqrt(dot3(d, d))[:, None]
    O = f32(0.5) * o.min(0) + f32(0.5) * o.max(0)
    D = f32(0.5) * dh.min(0) + f32(0.5) * dh.max(0)
    Dl = np.sqrt(dot3(D, D))
    D = D / Dl
    ro = np.sqrt(((o - O) ** 2).sum(1)).max() * f32(1.0001) + f32(1e-30)
    sigma = np.sqrt(((dh - D) ** 2).sum(1)).max() * f32(1.0001) + f32(2e-6)
    return dict(O=O, D=D, ro=f32(ro), sigma=f32(sigma), On=np.sqrt(dot3(O, O)) * f32(1.0001) + f32(ro), usable=bool(Dl > 0.25))


def certified(rec, pk):
    """the per-quad test of packet_cull_kernel"""
    if not pk["usable"]:
        return False
    w = rec["c"] - pk["O"]
    L = np.sqrt(dot3(w, w)) * f32(1.0001)
    crn = np.sqrt(dot3(cross3(w, pk["D"]), cross3(w, pk["D"])))
    delta = (crn * f32(0.9999) - L * pk["sigma"]) - (pk["ro"] + rec["R"]) - f32(1e-5) * (L + pk["ro"] + rec["R"])
    D = pk["D"]
    plo = sum(min(D[i] * rec["nlo"][i], D[i] * rec["nhi"][i]) for i in (2, 1, 0))
    phi = sum(max(D[i] * rec["nlo"][i], D[i] * rec["nhi"][i]) for i in (2, 1, 0))
    cmin = (plo if plo > 0 else (-phi if phi < 0 else f32(-1))) - pk["sigma"] - f32(1e-5)
    lhs = (rec["Nmin"] * cmin) * min(f32(0.3333), delta * rec["shape"]) * f32(0.99)
    rhs = f32(9.5367431640625e-07) * (rec["E"] * pk["On"] + rec["Pw"]) * f32(1.01)
    return bool(delta > 0 and cmin > 0 and rec["Nmin"] > 0 and lhs > 0 and lhs >= rhs)


def bumpy_quad(rng, amp):
    """40 triangles of a height field patch (4 x 5 cells), somewhere in a 40 x 20 scene, wound like the benchmark mesh"""
    x0, y0 = rng.uniform(-18, 14), rng.uniform(-12, 2)
    cell = rng.uniform(0.2, 0.8)
    ph = rng.uniform(0, 6.28, 2)

    def z(x, y):
        return 5.0 + amp * np.sin(0.75 * x + ph[0]) * np.cos(0.5 * y + ph[1])
    tris = []
    for i in range(4):
        for j in range(5):
            xs, ys = x0 + i * cell, y0 + j * cell
            a, b, c, d = [(xs, ys), (xs + cell, ys), (xs + cell, ys + cell), (xs, ys + cell)]
            P = [np.array([q[0], q[1], z(*q)]) for q in (a, b, c, d)]
            tris += [[P[0], P[2], P[1]], [P[0], P[3], P[2]]]
    return np.array(tris, np.float64)


def random_packet(rng, quad, n=32):
    """rays from a small origin region towards (or past) the scene, narrow cone"""
    O = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-40, -20)])
    target = np.array([rng.uniform(-22, 22), rng.uniform(-15, 8), 5.0])
    D = target - O
    o = O + rng.normal(size=(n, 3)) * rng.choice([1e-3, 0.05, 0.5])
    d = D + rng.normal(size=(n, 3)) * np.linalg.norm(D) * rng.choice([1e-3, 0.01, 0.05])
    d *= rng.uniform(0.5, 2.0, size=(n, 1))              # the reference does not normalise every direction
    return o, d


def coplanar_packet(rng, quad, n=32, tilt=1e-6):
    """adversarial: rays IN the plane of one triangle of the quad (nudged by a few fp32 ulps), starting far away, passing beside it"""
    t = quad[rng.integers(len(quad))]
    nrm = np.cross(t[1] - t[0], t[2] - t[0]); nrm /= np.linalg.norm(nrm)
    u = t[1] - t[0]; u /= np.linalg.norm(u)
    v = np.cross(nrm, u)
    cen = t.mean(0)
    side = rng.uniform(2.0, 15.0) * rng.choice([-1, 1])
    start = cen + v * side - u * rng.uniform(20, 40)
    o = start + (rng.normal(size=(n, 1)) * u + rng.normal(size=(n, 1)) * v) * 1e-3 + nrm * rng.normal(size=(n, 1)) * tilt
    d = u + v * rng.normal(size=(n, 1)) * 1e-3 + nrm * rng.normal(size=(n, 1)) * tilt * 0.1
    return o, d


def test_edge_functions_are_minus_d_dot_N_times_barycentrics():
    rng = np.random.default_rng(1)
    for _ in range(200):
        t = rng.normal(size=(3, 3)) * 3
        o, d = rng.normal(size=3) * 10, rng.normal(size=3)
        N = np.cross(t[1] - t[0], t[2] - t[0])
        tt = -np.dot(o - t[0], N) / np.dot(d, N)
        p = o + tt * d
        for k in range(3):
            a = (k + 1) % 3
            F = np.dot(t[a] - t[k], np.cross(d, o)) + np.dot(np.cross(t[a], t[k]), d)
            beta = np.dot(np.cross(t[a] - t[k], p - t[k]), N) / np.dot(N, N)          # barycentric coordinate of the opposite vertex
            assert abs(F + np.dot(d, N) * beta) <= 1e-9 * (1 + abs(F))


@pytest.mark.parametrize("amp", [0.0, 0.5, 2.0])
def test_certified_packets_are_rejected_by_the_reference_test(amp):
    rng = np.random.default_rng(int(amp * 10) + 7)
    fired = tried = 0
    for it in range(400):
        quad = bumpy_quad(rng, amp)
        rec = cull_record(quad)
        for make in (random_packet, random_packet, coplanar_packet):
            o, d = make(rng, quad)
            pk = packet_bounds(o, d)
            tried += 1
            if certified(rec, pk):
                fired += 1
                acc = reference_accepts(quad, o, d)
                assert not acc.any(), f"certificate fired but the reference accepts {int(acc.sum())} pairs (iteration {it}, {make.__name__})"
    assert fired > 0.3 * tried, f"certificate fired for {fired} of {tried} packets only"


def test_rays_in_the_plane_of_a_far_triangle_are_never_certified_but_can_be_accepted():
    """The reason the certificate lives in edge-function space: for rays within rounding noise of a far triangle's plane the
    reference's three edge values are noise, and it does accept some of them.  The certificate must not fire for such packets."""
    rng = np.random.default_rng(3)
    accepted = fired = 0
    for _ in range(300):
        quad = bumpy_quad(rng, 2.0)
        rec = cull_record(quad)
        o, d = coplanar_packet(rng, quad, n=64, tilt=0.0)      # exactly in the plane in float64: what is left after rounding to fp32 is noise
        pk = packet_bounds(o, d)
        fired += certified(rec, pk)
        accepted += int(reference_accepts(quad, o, d).any())
    assert fired == 0
    assert accepted > 0, "expected the fp32 reference test to accept at least one in-plane ray of a far triangle (it does on llvmpipe too)"
