"""Seeded synthetic multi-view-stereo scenes (inputs for tests and bench.py).

The reference ships no dense datasets (Gerrard-Hall /
South-Building are download-only, doc/datasets.rst:9-19) and its own tests
synthesise inputs (scene/synthetic.h:40-117), so the PatchMatch workloads here
are ray-cast stand-ins with the same shapes: N pinhole cameras on a ring of
radius `ring_radius` looking at the origin (scene/synthetic.cc:458-464 places
cameras on a radius-5 sphere looking at the origin), a textured unit cube on a
ground plane inside a textured background sphere, rendered with a band-limited
procedural solid texture whose wavelengths are tied to the pixel footprint so
that an 11x11 NCC window always sees texture.

Pure torch so the same code renders 96x72 test images on the CPU and
2560x1920 bench images on the GPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class View:
    K: np.ndarray       # (3,3) float32
    R: np.ndarray       # (3,3) float32, x_cam = R x_world + T
    T: np.ndarray       # (3,)  float32
    gray: np.ndarray    # (H,W) uint8
    depth: np.ndarray   # (H,W) float32 ground-truth z-depth
    normal: np.ndarray  # (3,H,W) float32 ground-truth camera-frame normal


def _look_at(center: np.ndarray, target: np.ndarray) -> np.ndarray:
    """World->camera rotation with +z forward, +y down (COLMAP convention)."""
    z = target - center
    z = z / np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)  # x = down x forward (right-handed, y down)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], 0)


def ring_cameras(num: int, width: int, height: int, focal: float, ring_radius: float = 5.0,
                 cam_height: float = -1.2, arc_deg: float = 360.0, start_deg: float = 0.0):
    """`num` pinhole cameras on a ring (or arc) around the y axis, looking at the origin."""
    cams = []
    for i in range(num):
        denom = num if arc_deg >= 360.0 else max(num - 1, 1)
        a = math.radians(start_deg + arc_deg * i / denom)
        C = np.array([ring_radius * math.sin(a), cam_height, -ring_radius * math.cos(a)])
        R = _look_at(C, np.zeros(3))
        T = -R @ C
        K = np.array([[focal, 0, (width - 1) / 2.0], [0, focal, (height - 1) / 2.0], [0, 0, 1]])
        cams.append((K.astype(np.float32), R.astype(np.float32), T.astype(np.float32)))
    return cams


def _texture_params(seed: int, n_comp: int, fmin: float, fmax: float, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = torch.randn(n_comp, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=1, keepdim=True)
    logf = torch.rand(n_comp, generator=g, dtype=torch.float64) * (math.log(fmax) - math.log(fmin)) + math.log(fmin)
    f = torch.exp(logf)
    phase = torch.rand(n_comp, generator=g, dtype=torch.float64) * 2 * math.pi
    amp = 1.0 / torch.sqrt(f / fmin)
    return (d * f[:, None]).to(device), phase.to(device), amp.to(device)


def render_view(K, R, T, width, height, seed=0, device="cpu", n_comp=24, chunk_rows=256):
    """Ray-cast the scene for one camera. Returns (gray u8, depth f32, normal f32[3]) as torch tensors."""
    dev = torch.device(device)
    dt = torch.float64
    Kt = torch.as_tensor(np.asarray(K, np.float64), device=dev)
    Rt = torch.as_tensor(np.asarray(R, np.float64), device=dev)
    Tt = torch.as_tensor(np.asarray(T, np.float64), device=dev)
    Cw = -(Rt.T @ Tt)
    fx = float(Kt[0, 0])
    # pixel footprint at the scene centre sets the texture band: wavelengths 5..40 px
    dist = float(Cw.norm())
    px = dist / fx
    wave, phase, amp = _texture_params(seed, n_comp, 2 * math.pi / (40 * px), 2 * math.pi / (5 * px), dev)

    gray = torch.empty(height, width, dtype=torch.uint8, device=dev)
    depth = torch.empty(height, width, dtype=torch.float32, device=dev)
    normal = torch.empty(3, height, width, dtype=torch.float32, device=dev)
    xs = torch.arange(width, device=dev, dtype=dt)
    for r0 in range(0, height, chunk_rows):
        r1 = min(height, r0 + chunk_rows)
        ys = torch.arange(r0, r1, device=dev, dtype=dt)
        v, u = torch.meshgrid(ys, xs, indexing="ij")
        dc = torch.stack([(u - Kt[0, 2]) / Kt[0, 0], (v - Kt[1, 2]) / Kt[1, 1], torch.ones_like(u)], -1)
        dw = dc @ Rt  # R^T applied to row vectors
        o = Cw
        t_best = torch.full(u.shape, float("inf"), device=dev, dtype=dt)
        n_best = torch.zeros(u.shape + (3,), device=dev, dtype=dt)
        # background sphere radius 20 (seen from inside)
        b = (dw * o).sum(-1)
        a = (dw * dw).sum(-1)
        c = (o * o).sum() - 400.0
        t_s = (-b + torch.sqrt(b * b - a * c)) / a
        t_best = t_s
        p_s = o + t_s[..., None] * dw
        n_best = -p_s / 20.0
        # ground plane y = 1
        t_g = (1.0 - o[1]) / dw[..., 1]
        hit = (t_g > 0) & (t_g < t_best)
        t_best = torch.where(hit, t_g, t_best)
        n_g = torch.tensor([0.0, -1.0, 0.0], device=dev, dtype=dt)
        n_best = torch.where(hit[..., None], n_g.expand_as(n_best), n_best)
        # unit cube |x|,|y|,|z| <= 1 (slab method), rotated 25 deg about y for obliqueness
        ca, sa = math.cos(math.radians(25.0)), math.sin(math.radians(25.0))
        Rb = torch.tensor([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], device=dev, dtype=dt)
        ob = Rb @ o
        db = dw @ Rb.T
        inv = 1.0 / db
        t0 = (-1.0 - ob) * inv
        t1 = (1.0 - ob) * inv
        tmin = torch.minimum(t0, t1)
        tmax = torch.maximum(t0, t1)
        tn, axis = tmin.max(-1)
        tf = tmax.min(-1).values
        hit = (tn < tf) & (tn > 0) & (tn < t_best)
        t_best = torch.where(hit, tn, t_best)
        nb = torch.zeros_like(n_best)
        sign = -torch.sign(torch.gather(db, -1, axis[..., None])).squeeze(-1)
        nb.scatter_(-1, axis[..., None], sign[..., None])
        nb = nb @ Rb  # back to world
        n_best = torch.where(hit[..., None], nb, n_best)

        P = o + t_best[..., None] * dw
        arg = P @ wave.T + phase
        tex = (torch.sin(arg) * amp).sum(-1) / float(amp.norm())
        g = torch.clamp(128.0 + 70.0 * tex, 0, 255)
        gray[r0:r1] = torch.round(g).to(torch.uint8)
        depth[r0:r1] = t_best.to(torch.float32)  # dc has z = 1, so t is the z-depth
        ncam = n_best @ Rt.T
        normal[:, r0:r1] = ncam.permute(2, 0, 1).to(torch.float32)
    return gray, depth, normal


def make_scene(num_views: int, width: int, height: int, focal: float | None = None, seed: int = 0,
               device: str = "cpu", arc_deg: float = 360.0, ring_radius: float = 5.0,
               start_deg: float = 0.0) -> list[View]:
    if focal is None:
        focal = 2400.0 * width / 2560.0
    views = []
    for (K, R, T) in ring_cameras(num_views, width, height, focal, ring_radius, arc_deg=arc_deg,
                                  start_deg=start_deg):
        g, d, n = render_view(K, R, T, width, height, seed=seed, device=device)
        views.append(View(K, R, T, g.cpu().numpy(), d.cpu().numpy(), n.cpu().numpy()))
    return views


def depth_range(views: list[View], ref_idx: int):
    d = views[ref_idx].depth
    return float(d.min() * 0.9), float(d.max() * 1.1)


def as_image_dicts(views: list[View], with_maps: bool = False):
    out = []
    for v in views:
        d = dict(K=v.K, R=v.R, T=v.T, gray=v.gray)
        if with_maps:
            d["depth"] = v.depth
            d["normal"] = v.normal
        out.append(d)
    return out
