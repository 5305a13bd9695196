"""TEST INFRASTRUCTURE: fit the camera and the time of the reference's own screenshots.

The reference holds rendered frames of this very path: /root/reference/Images/*.png, 1200 x 800 screenshots of its
window (README.md:30-104; the window is 1200 x 800 and the camera fovy 60 deg, aspect 1.5, roll 0, FPS mode:
Engine/Application.cpp:39-40, 214-224).  They are post-tone-map LDR images of an unknown camera and time, so they cannot be
compared bit for bit -- but everything else about them is known, and a camera has six numbers plus the time: this tool
finds them by least squares on the oracle's frame pushed through the oracle's restatement of HDR::process
(oracle/postprocess.h), and then reports how far apart the two images are.  The fitted parameters are committed in
tests/golden/reference_images.json; tests/test_reference_images_cpu.py re-renders at them and asserts the statistics.

The screenshots are read where they lie and never copied.  Nothing here is product code; the oracle is the renderer.

  python tools/fit_reference_images.py sphere --init ex ey ez yaw pitch stime [--spread ...] [--save /tmp/prefix]
(global_search / refine / scan_time are the pieces the five committed fits were made with: a quasi-random search over the camera box
at every 16th pixel, Nelder-Mead on finer and finer samples, a scan of the time against the sky -- and, where a scene repeats,
an enumeration of the equivalent cameras against what does not repeat: the gems on their checker floor, the sky in the cubes' tops.)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402

IMAGES = "/root/reference/Images"
W, H = 1200, 800  # Engine/Application.cpp:39-40
FOVY = np.float32(60.0) * np.float32(3.14159265358979) / np.float32(180.0)  # Application.cpp:216 ToRadian(60)
ASPECT = np.float32(W) / np.float32(H)

# image name -> (file, scene, initial guess [ex, ey, ez, yaw, pitch, stime], scene variables moved from their defaults)
# yaw / pitch in radians: dir = (cos(pitch) sin(yaw), sin(pitch), cos(pitch) cos(yaw)); the start-up camera is yaw 0, pitch -0.32.
TARGETS = {
    "sphere": ("sphere.png", "fast_sphere", None, {}),
    "cube-sea": ("cube-sea.png", "cube_sea", None, {}),
    "labyrinth": ("labyrinth.png", "labyrinth", None, {}),
    "gems": ("gems.png", "gems", None, {}),
    "table": ("table.png", "table", None, {}),
    "transparency": ("transparency.png", "basic_transparency", None, {}),
    "shell": ("shell.png", "shell", None, {}),
    "distortion": ("distortion.png", "distortion", None, {}),
    "gyroid": ("gyroid.png", "gyroid", None, {}),
    "fractal": ("fractal.png", "fractal", None, {}),
    # two sliders moved: the cutting box pulled in (boxoffset 2 -> 1), the line threshold lowered (thres 0.4 -> 0.2); least squares: 0.99999697, 0.2023
    "coordinate-material": ("coordinate material.png", "coordinate_material", None, {"boxoffset": 1.0, "thres": 0.2}),
    # taken with three sliders moved (VariableManager): found by least squares, they sit on the sliders' 0.05 grid to three decimals
    "neon": ("neon.png", "neon", None, {"red": 0.2, "green": 1.5, "blue": 2.05}),
    "sierpinski": ("sierpinski.png", "sierpinski", None, {}),
    # taken with an older version of the scene file, whose lights' phases step the other way (oracle/scenes.h: SceneLightShadowsT)
    "multi-lights": ("multi-lights.png", "light_shadows_backwards", None, {}),
    "multi-lights-todays-file": ("multi-lights.png", "light_shadows", None, {}),
    # the scene of BASELINE configuration 5 at its default sliders: two lattices of reflecting blobs, a refracting lens, an emissive ball, a
    # half-mirror in a wooden frame turning with the time
    "lense1": ("lense1.png", "lense", None, {}),
    # a spring that hops along a parabola while the floor scrolls under it (period 8 / 3 s of stime); the sky: see ROWS_FROM
    "spiral": ("spiral.png", "spiral", None, {}),
}


def load_reference(name):
    from PIL import Image

    path = os.path.join(IMAGES, TARGETS[name][0] if name in TARGETS else name)
    img = np.asarray(Image.open(path).convert("RGB"))
    assert img.shape == (H, W, 3), img.shape
    return img


def direction(yaw, pitch):
    return (math.cos(pitch) * math.sin(yaw), math.sin(pitch), math.cos(pitch) * math.cos(yaw))


def make_frame(scene, p, variables=None):
    eye = (float(p[0]), float(p[1]), float(p[2]))
    basis = po.camera_direction(eye, direction(p[3], p[4]), FOVY, ASPECT)
    f = po.default_frame(scene, W, H, basis=basis, stime=float(p[5]))
    if variables:
        slots = {n: s for n, _a, _b, _c, _d, _e, s in po.var_table(scene)}
        for k, v in variables.items():
            if slots[k] >= 0:
                f.scene_var[slots[k]] = v
            else:
                setattr(f, k, v)
    return f


def tonemap(rgba):
    """pshader_hdr.hlsl:20-25 without the bloom term: lerp(scene, 1 - exp(-scene), scene.a), to [0, 1]."""
    rgb = rgba[..., :3].astype(np.float64)
    a = rgba[..., 3:4].astype(np.float64)
    return np.clip(rgb + a * ((1.0 - np.exp(-rgb)) - rgb), 0.0, 1.0)


def render_sample(scene, p, s, variables=None, off=None):
    """The oracle's frame at every s-th pixel of the 1200 x 800 frame (true pixel footprints), tone-mapped, no bloom."""
    f = make_frame(scene, p, variables)
    o = s // 2 if off is None else off
    out, _, _ = po.render(scene, f, region=(o, o, W, H), step=(s, s))
    return tonemap(out[o::s, o::s])


def render_full(scene, p, variables=None):
    """Full frame through the oracle's HDR::process (fp16 target, bloom, tone map, unorm8)."""
    f = make_frame(scene, p, variables)
    out, _, totals = po.render(scene, f)
    h = po.half_to_float(po.float_to_half(out)).astype(np.float16)
    _b1, _b2, ldr = po.postprocess(h)
    return ldr[..., :3], out, totals


def blur(img, sigma):
    if sigma <= 0:
        return img
    from scipy.ndimage import gaussian_filter

    return gaussian_filter(img, sigma=(sigma, sigma, 0), mode="nearest")


class Objective:
    def __init__(self, name, scene, ref8, s, sigma, variables=None, mask_bright=True, fit_vars=()):
        self.scene, self.s, self.sigma, self.variables = scene, s, sigma, dict(variables or {})
        self.fit_vars = tuple(fit_vars)
        o = s // 2
        ref = ref8[o::s, o::s].astype(np.float64) / 255.0
        # bloom is not in the fast objective: leave out what the screenshot shows saturated (highlights and their halo)
        self.weight = np.ones(ref.shape[:2])
        if mask_bright:
            bright = (ref.max(axis=2) > 0.92).astype(np.float64)
            self.weight = 1.0 - np.clip(blur(bright[..., None], 24.0 / s)[..., 0] * 8.0, 0, 1)
        self.ref = blur(ref, sigma)
        self.evals = 0

    def image(self, p):
        v = dict(self.variables)
        for i, k in enumerate(self.fit_vars):
            v[k] = float(p[6 + i])
        return blur(render_sample(self.scene, p, self.s, v), self.sigma)

    def __call__(self, p):
        self.evals += 1
        if abs(p[4]) > 1.39:  # Application.cpp:218-219: the camera's pitch is held within 80 degrees
            return 1.0
        d = (self.image(p) - self.ref) ** 2
        return float((d.sum(axis=2) * self.weight).sum() / (3.0 * self.weight.sum()))


def nelder_mead(obj, p0, scale, iters):
    from scipy.optimize import minimize

    n = len(p0)
    simplex = np.vstack([p0] + [np.asarray(p0) + np.eye(n)[i] * scale[i] for i in range(n)])
    r = minimize(obj, p0, method="Nelder-Mead", options={"initial_simplex": simplex, "maxfev": iters, "xatol": 1e-6, "fatol": 1e-10})
    return r.x, r.fun


def fit(name, p0, spread, variables=None, fit_vars=(), var0=(), var_spread=(), quick=False, seed=1, verbose=True, schedule=None):
    """Coarse random search about p0 within +-spread, then Nelder-Mead on finer and finer samples."""
    _file, scene, _g, _v = TARGETS[name]
    ref8 = load_reference(name)
    rng = np.random.default_rng(seed)
    p0 = np.concatenate([np.asarray(p0, float), np.asarray(var0, float)])
    spread = np.concatenate([np.asarray(spread, float), np.asarray(var_spread, float)])
    t0 = time.time()
    schedule = schedule or ([(8, 2.0, 600 if not quick else 150, 400), (4, 1.5, 0, 500), (2, 1.0, 0, 400), (2, 0.0, 0, 300)])
    best = [(None, p0)]
    for s, sigma, nrandom, iters in schedule:
        obj = Objective(name, scene, ref8, s, sigma, variables, fit_vars=fit_vars)
        cands = [(obj(p), p) for _f, p in best]
        for _ in range(nrandom):
            p = p0 + (rng.random(len(p0)) * 2 - 1) * spread
            cands.append((obj(p), p))
        cands.sort(key=lambda c: c[0])
        keep = cands[: (4 if nrandom else 2)]
        best = []
        for f0, p in keep:
            x, fx = nelder_mead(obj, p, spread * (0.15 if nrandom else 0.03), iters)
            best.append((fx, x))
        best.sort(key=lambda c: c[0])
        spread = spread * 0.5
        if verbose:
            print("  step %d sigma %.1f: loss %.6f  p = %s  (%d evals, %.0f s)" % (s, sigma, best[0][0], np.array2string(best[0][1], precision=5), obj.evals, time.time() - t0), flush=True)
    return best[0][1], best[0][0]


def global_search(name, lo, hi, n, s=16, sigma=2.0, keep=24, seed=7, variables=None, verbose=True, recheck=0):
    """n quasi-random cameras in the box [lo, hi], the `keep` best refined by Nelder-Mead at (s, sigma); returns them sorted.
    recheck: that many of the best are looked at again at every 8th pixel with half the blur before the `keep` are chosen -- from a
    high camera a far checker floor is uniformly grey at the coarse level, and every camera over it looks equally good."""
    from scipy.stats import qmc

    _file, scene, _g, _v = TARGETS[name]
    ref8 = load_reference(name)
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    obj = Objective(name, scene, ref8, s, sigma, variables)
    pts = qmc.scale(qmc.Sobol(d=len(lo), seed=seed).random(n), lo, np.maximum(hi, lo + 1e-9))
    t0 = time.time()
    cands = sorted(((obj(p), tuple(p)) for p in pts), key=lambda c: c[0])
    if verbose:
        print("  global: %d samples in %.0f s, best %.5f, %d-th %.5f" % (n, time.time() - t0, cands[0][0], keep, cands[keep - 1][0]), flush=True)
    if recheck:
        fine = Objective(name, scene, ref8, 8, 1.0, variables)
        cands = sorted(((fine(np.asarray(p)), p) for _f, p in cands[:recheck]), key=lambda c: c[0])
        obj = fine
        if verbose:
            print("  recheck of %d at every 8th pixel: best %.5f, %d-th %.5f (%.0f s)" % (recheck, cands[0][0], keep, cands[min(keep, len(cands)) - 1][0], time.time() - t0), flush=True)
    # keep candidates that are not neighbours of a better one
    chosen = []
    span = hi - lo + 1e-9
    for f, p in cands:
        if all(np.abs((np.asarray(p) - np.asarray(q)) / span).max() > 0.02 for _g, q in chosen):
            chosen.append((f, p))
        if len(chosen) == keep:
            break
    out = []
    for f, p in chosen:
        x, fx = nelder_mead(obj, np.asarray(p), span * 0.02, 300)
        out.append((fx, x))
    out.sort(key=lambda c: c[0])
    if verbose:
        for fx, x in out[:6]:
            print("    %.5f %s" % (fx, np.array2string(x, precision=4)), flush=True)
    return out


def refine(name, p, levels=((8, 2.0, 400), (4, 1.0, 400), (2, 0.5, 300)), scale=(0.05, 0.05, 0.05, 0.02, 0.01, 0.5), variables=None, verbose=True):
    _file, scene, _g, _v = TARGETS[name]
    ref8 = load_reference(name)
    scale = np.asarray(scale, float)
    f = None
    for s, sigma, iters in levels:
        obj = Objective(name, scene, ref8, s, sigma, variables)
        p, f = nelder_mead(obj, np.asarray(p, float), scale, iters)
        scale = scale * 0.5
        if verbose:
            print("  refine step %d sigma %.1f: %.6f %s" % (s, sigma, f, np.array2string(np.asarray(p), precision=5)), flush=True)
    return p, f


def scan_time(name, p, times, s=4, sigma=1.0, variables=None):
    _file, scene, _g, _v = TARGETS[name]
    obj = Objective(name, scene, load_reference(name), s, sigma, variables)
    res = []
    for t in times:
        q = np.array(p, float)
        q[5] = t
        res.append((obj(q), float(t)))
    res.sort()
    return res


def floor_equivalents(name, p, reach=6, s=8, sigma=1.0, times=None, keep=6):
    """The checker floor (sdf_common.hlsl:24-59) looks the same from eye + (a, b) with a + b even, and after a quarter turn about a tile
    centre: a fit that locked onto the floor -- most of most screenshots -- may have the scene's objects in the wrong place.  Tries the
    equivalent cameras (and `times`, if what distinguishes them moves) and returns the best `keep` as (loss, params)."""
    _file, scene, _g, _v = TARGETS[name]
    obj = Objective(name, scene, load_reference(name), s, sigma)
    p = np.asarray(p, float)
    c = np.array([0.5, 0.5])
    out = []
    for k in range(4):
        ang = k * math.pi / 2
        ca, sa = math.cos(-ang), math.sin(-ang)
        e = np.array([p[0], p[2]]) - c
        e2 = c + np.array([ca * e[0] - sa * e[1], sa * e[0] + ca * e[1]])
        for a in range(-reach, reach + 1):
            for b in range(-reach, reach + 1):
                if (a + b) % 2:
                    continue
                for t in (times if times is not None else [p[5]]):
                    q = p.copy()
                    q[0], q[2], q[3], q[5] = e2[0] + a, e2[1] + b, p[3] + ang, t
                    out.append((obj(q), q))
    out.sort(key=lambda x: x[0])
    return out[:keep]


class FullObjective:
    """Sum of squares over the whole frame through the oracle's HDR::process (bloom included): the final polish."""

    def __init__(self, name, variables=None):
        self.scene = TARGETS[name][1]
        self.ref = load_reference(name).astype(np.float64)
        self.variables = variables
        self.evals = 0

    def __call__(self, p):
        self.evals += 1
        ldr, _hdr, _t = render_full(self.scene, p, self.variables)
        return float(((ldr.astype(np.float64) - self.ref) ** 2).mean()) / 65025.0


# spiral.png shows a sky that today's sky_color() paints at no time (an older sky; floor, spring and shadow agree to the pixel): that
# screenshot is compared from the horizon down
ROWS_FROM = {"spiral": 256}


def compare(name, p, variables=None, save=None):
    """Statistics of |oracle through HDR::process - screenshot| at the fitted parameters, on the full 1200 x 800 frame (from row
    ROWS_FROM[name] down, for a screenshot listed there)."""
    _file, scene, _g, _v = TARGETS[name]
    if variables is None:
        variables = _v
    r0 = ROWS_FROM.get(name, 0)
    ref8 = load_reference(name).astype(np.int32)[r0:]
    ldr, hdr, totals = render_full(scene, p, variables)
    ldr, hdr = ldr[r0:], hdr[r0:]
    d = np.abs(ldr.astype(np.int32) - ref8).max(axis=2)  # per pixel: largest channel difference, in 1/255
    sky = hdr[..., 3] == 0  # misses write alpha 0 only when use_hdr is off; keep a geometric notion instead
    # an edge pixel is one whose 3 x 3 neighbourhood in the SCREENSHOT spans more than 24/255: a sub-pixel shift of a
    # silhouette or a checker line changes such a pixel by the whole contrast, whatever the renderer
    from scipy.ndimage import maximum_filter, minimum_filter

    g = ref8.max(axis=2)
    edge = (maximum_filter(g, 3) - minimum_filter(g, 3)) > 24
    flat = ~edge
    stats = {
        "mean_abs_err": float(np.abs(ldr.astype(np.int32) - ref8).mean()),
        "within_3": float((d <= 3).mean()),
        "within_8": float((d <= 8).mean()),
        "flat_fraction": float(flat.mean()),
        "flat_within_3": float((d[flat] <= 3).mean()),
        "flat_within_8": float((d[flat] <= 8).mean()),
        "flat_mean_abs_err": float(np.abs(ldr.astype(np.int32) - ref8)[flat].mean()),
        "p99_flat": float(np.percentile(d[flat], 99)),
        "rays": int(totals[1]),
    }
    if save:
        from PIL import Image

        Image.fromarray(ldr).save(save + "_oracle.png")
        Image.fromarray(np.clip(d * 8, 0, 255).astype(np.uint8)).save(save + "_diff_x8.png")
    return stats, ldr, d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--init", type=float, nargs=6)
    ap.add_argument("--spread", type=float, nargs=6, default=[0.5, 0.3, 0.5, 0.3, 0.1, 5.0])
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--compare-only", action="store_true")
    ap.add_argument("--save", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    p = np.asarray(a.init, float)
    if not a.compare_only:
        p, loss = fit(a.name, p, a.spread, quick=a.quick)
    stats, _, _ = compare(a.name, p, save=a.save)
    print(json.dumps({"name": a.name, "params": [float(x) for x in p], "stats": stats}, indent=1))


if __name__ == "__main__":
    main()
