"""The nested-shell rule of the render kernels (trav_begin in raytracer_amd/csrc/rt_device.hip), modelled in numpy f64.

Of a group of concentric, well separated spheres the kernels test only a run of three (or, away from the tangent, two)
that a segment can meet first.  This file restates the kernels' arithmetic operation by operation (the sphere test of
src/geometry.rs:104-121 as the device evaluates it: b = 2 oc.d, c = oc.oc - r*r, both roots, the 1e-6 rule) and checks
the rule against testing every shell, on rays chosen to hurt: origins on a shell, a hair off one, hit points of earlier
segments, directions grazing a shell at angles from 1e-12 to 1e-1, shell gaps from 1e-4 of the radius upwards.
It checks the rule, not the device code; the device code is held to the same result by the GPU parity tests
(render dump == per-path entry point, which tests every sphere)."""
import numpy as np



def unit_f64(d32):
    a = d32.astype(np.float64)
    e = (a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1] + a[:, 2] * a[:, 2]) - 1.0
    k = 1.0 - 0.5 * e + 0.375 * (e * e)
    return a * k[:, None]

def normalize32(v):
    v = v.astype(np.float32)
    inv = (np.float32(1.0) / np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])).astype(np.float32)
    return v * inv[:, None]

def centre_terms(o, d, c):
    r = o - c
    b = 2.0 * ((r[:, 0] * d[:, 0] + r[:, 1] * d[:, 1]) + r[:, 2] * d[:, 2])
    oc2 = (r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2]
    return b, b * b, oc2

def candidate(b, bb, oc2, radius):
    with np.errstate(invalid='ignore'):
        c = oc2 - radius * radius
        disc = bb - 4.0 * c
        sq = np.sqrt(disc)
        ts = -0.5 * (b + sq)
        tl = ts + sq
        near_ok = ts > 0.0
        hit = ~(disc < 0.0) & (near_ok | (tl > 1.0e-6))
        t = np.where(near_ok, ts, tl)
    return hit, t

def run(rng, n_rays, radii):
    m = len(radii)
    c = rng.uniform(-1, 1, 3) * rng.choice([0.0, 1.0, 100.0])
    rmin = radii[0]
    kind = rng.integers(0, 6, n_rays)
    u = rng.normal(size=(n_rays, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    j = rng.integers(0, m, n_rays)
    rr = radii[j]
    # origins: 0 on a shell, 1 a hair off a shell, 2 between shells, 3 inside the smallest, 4 outside all (near), 5 hit point of a previous segment
    dist = np.where(kind == 0, rr,
           np.where(kind == 1, rr * (1.0 + rng.choice([-1, 1], n_rays) * 10.0 ** rng.uniform(-16, -6, n_rays)),
           np.where(kind == 2, rr * (1.0 + rng.uniform(0, 1, n_rays) * ((np.append(radii, radii[-1] * 2)[j + 1] / rr) - 1.0)),
           np.where(kind == 3, rmin * rng.uniform(0, 1, n_rays),
                    radii[-1] * (1.0 + 10.0 ** rng.uniform(-3, 2, n_rays))))))
    o = c + u * dist[:, None]
    # kind 5: shoot a ray from far away, take its hit point on shell j
    sel = kind == 5
    if sel.any():
        k = sel.sum()
        o0 = c + u[sel] * (radii[-1] * 3.0)
        tgt = c + (rng.normal(size=(k, 3)) * 0.5) * rr[sel][:, None]
        d0 = unit_f64(normalize32(tgt - o0))
        b, bb, oc2 = centre_terms(o0, d0, c)
        hit, t = candidate(b, bb, oc2, rr[sel])
        hp = o0 + d0 * np.where(hit, t, 1.0)[:, None]
        o[sel] = np.where(hit[:, None], hp, o[sel])
    # directions: random, or nearly tangent, or straight in / out
    dk = rng.integers(0, 4, n_rays)
    w = rng.normal(size=(n_rays, 3))
    oc = o - c
    ocn = oc / np.maximum(np.linalg.norm(oc, axis=1, keepdims=True), 1e-300)
    tang = w - (w * ocn).sum(1, keepdims=True) * ocn
    tang /= np.maximum(np.linalg.norm(tang, axis=1, keepdims=True), 1e-300)
    eps = (rng.choice([-1, 1], n_rays) * 10.0 ** rng.uniform(-12, -1, n_rays))[:, None]
    d = np.where((dk == 0)[:, None], w, np.where((dk == 1)[:, None], tang + eps * ocn, np.where((dk == 2)[:, None], -ocn + 0.3 * w, ocn + 0.3 * w)))
    d = unit_f64(normalize32(d))
    b, bb, oc2 = centre_terms(o, d, c)
    far = oc2 > 1.0e6 * rmin * rmin
    # exhaustive
    best_t = np.full(n_rays, np.inf); best_k = np.full(n_rays, -1)
    for i in range(m):
        hit, t = candidate(b, bb, oc2, radii[i])
        better = hit & (t < best_t)
        best_t = np.where(better, t, best_t); best_k = np.where(better, i, best_k)
    # the rule
    r2 = radii * radii
    outside = np.zeros(n_rays, dtype=np.int64); r2_last = np.full(n_rays, -1.0e300)
    for i in range(m):
        out = oc2 > r2[i]
        outside += out
        r2_last = np.where(out, r2[i], r2_last)
    c_last = oc2 - r2_last
    first = outside - np.where(c_last < 1.0e-9 * oc2, 2, 1)
    return dict(b=b, bb=bb, oc2=oc2, far=far, best_t=best_t, best_k=best_k, first=first, radii=radii, kind=kind, dk=dk)

def rule(R, n_cand, by_direction, tau_abs=1.6e-11, tau_rel=1e-6):
    b, bb, oc2, first, radii = R['b'], R['bb'], R['oc2'], R['first'], R['radii']
    m = len(radii); n = len(b)
    start = first.copy(); cnt = np.full(n, 3)
    if by_direction:
        safe = (bb > tau_abs) & (bb > tau_rel * oc2)
        start = np.where(safe & (b > 0.0), first + 1, first)
        cnt = np.where(safe, 2, 3)
    best_t = np.full(n, np.inf); best_k = np.full(n, -1)
    for cidx in range(3):
        i = start + cidx
        valid = (i >= 0) & (i < m) & (cidx < cnt)
        hit, t = candidate(b, bb, oc2, radii[np.clip(i, 0, m - 1)])
        better = valid & hit & (t < best_t)
        best_t = np.where(better, t, best_t); best_k = np.where(better, np.clip(i, 0, m - 1), best_k)
    return best_t, best_k


def test_shell_rule_agrees_with_testing_every_shell():
    rng = np.random.default_rng(7)
    total = 0
    for trial in range(24):
        m = int(rng.integers(4, 10))
        scale = 10.0 ** rng.uniform(-2, 3)
        gaps = 1.0 + 10.0 ** rng.uniform(-4 + 1e-3, 0, m - 1)
        radii = scale * np.concatenate([[1.0], np.cumprod(gaps)])
        if not np.all(np.diff(radii) >= 4.0e-6):
            continue  # rt_scene_create would not mark such a group
        R = run(rng, 12000, radii)
        near = ~R["far"]  # the kernels test every shell when an origin of the wave is farther out than this
        for by_direction in (False, True):
            t, k = rule(R, 3, by_direction)
            same = ((t == R["best_t"]) & (k == R["best_k"])) | ~near
            assert same.all(), (trial, by_direction, int((~same).sum()))
        total += int(near.sum())
    assert total > 200000
