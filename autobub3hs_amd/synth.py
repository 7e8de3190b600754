"""Seeded, integer-only synthetic PICO-like events (SURVEY.md section 8d generator).

Every value is produced with int64 +,*,>>,&,^,% only, so numpy (CPU tests, golden fixtures) and
torch (bench: frames generated straight into HBM) give bit-identical frames.

Scene: smooth background + fixed-pattern noise per camera, per-frame noise = sum of four U{-1,0,1}
draws (sigma ~1.6 ADU), optional whole-frame LED flicker (+3 ADU on one frame), and bubbles:
discs of radius 2+1.5k px at frame t0+k with +-40 ADU contrast.
"""
import numpy as np

BASE_SEED = 0xAB0B3


def _hash32(x):
    """32-bit integer mix on int64 arrays (values stay < 2**59: no overflow, numpy == torch)."""
    x = x & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    x = ((x ^ (x >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return x


def _hash32_scalar(v):
    v &= 0xFFFFFFFF
    v = ((v ^ (v >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    v = ((v ^ (v >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    return v ^ (v >> 16)


class _NP:
    int64 = np.int64

    @staticmethod
    def arange(n, device=None):
        return np.arange(n, dtype=np.int64)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def clip(a, lo, hi):
        return np.clip(a, lo, hi)

    @staticmethod
    def to_u8(a):
        return a.astype(np.uint8)

    @staticmethod
    def empty_u8(shape, device=None):
        return np.empty(shape, np.uint8)


class _TORCH:
    def __init__(self):
        import torch

        self.t = torch
        self.int64 = torch.int64

    def arange(self, n, device=None):
        return self.t.arange(n, dtype=self.t.int64, device=device)

    def where(self, c, a, b):
        return self.t.where(c, a, b)

    def clip(self, a, lo, hi):
        return self.t.clamp(a, lo, hi)

    def to_u8(self, a):
        return a.to(self.t.uint8)

    def empty_u8(self, shape, device=None):
        return self.t.empty(shape, dtype=self.t.uint8, device=device)


def _backend(xp):
    return _TORCH() if xp == "torch" else _NP()


def background(W, H, cam_seed, xp="numpy", device=None):
    """B(y,x) as int64 [H,W] (not yet clipped)."""
    be = _backend(xp)
    x = be.arange(W, device)[None, :]
    y = be.arange(H, device)[:, None]
    t = x % 74
    tri = be.where(t < 37, t, 74 - t)
    fixed = _hash32((((y * W + x) * 2654435761) & 0xFFFFFFFF) ^ _hash32_scalar(cam_seed * 7919 + 1)) % 5 - 2
    return 40 + (x * 60) // W + (y * 30) // H + (12 * tri) // 37 + fixed


def frame_noise(W, H, frame_seed, xp="numpy", device=None):
    """Sum of four U{-1,0,1} draws per pixel, int64 [H,W]."""
    be = _backend(xp)
    x = be.arange(W, device)[None, :]
    y = be.arange(H, device)[:, None]
    s = _hash32_scalar(frame_seed)
    h = _hash32((((y * W + x) * 2654435761) & 0xFFFFFFFF) ^ s)
    n = (h & 0xFF) % 3 + ((h >> 8) & 0xFF) % 3 + ((h >> 16) & 0xFF) % 3 + ((h >> 24) & 0xFF) % 3
    return n - 4


class EventSpec:
    """What happens in one (event, camera) stack."""

    def __init__(self, F, t0=None, bubbles=(), flicker=None, flicker_adu=3, dense_from=None, hot_per_400=0):
        self.F = F
        self.t0 = t0            # genesis frame index (None: no bubble -> status -3)
        self.bubbles = list(bubbles)  # [(cx, cy, contrast)]
        self.flicker = flicker  # frame index with +flicker_adu on the whole frame, or None
        self.flicker_adu = flicker_adu
        # "decompression": from this frame on 35 % of the image (a band of columns) carries the offsets +8, +8, -8, -8, +8, ..
        # ADU, so that every such frame differs from the frame two before it by 16 ADU over that band.  The band is also
        # livelier in the training frames (-2 in frame 0, +2 in frame 1: sigma = 2 there, 6 sigma = 12), so 16 ADU is more
        # than 6 sigma -- dense work for the trigger search, if it ever looks there -- while |f - mu| = 8 stays below it:
        # the tracking frames keep their one bubble (None: never)
        self.dense_from = dense_from
        # hot pixels: this many of every 400 pixels (fixed per camera) are exactly constant in frames 0 and 1 (the training
        # frames: sigma = 0 there) and jitter by U{-3..3} in every later frame
        self.hot_per_400 = hot_per_400


def random_spec(W, H, F, event, cam, p_second=0.2, p_none=0.0, p_flicker=0.0, margin=40, accept=None, regime="default"):
    """Deterministic spec from (event, cam) -- host-side, tiny.  `accept(cx, cy)` (optional) rejects bubble positions
    (e.g. outside the fiducial mask); `regime`: "default", "post_trigger_dense" (EventSpec.dense_from = t0 + 3) or
    "noisy" (hot pixels at ten times the default's density of supra-threshold pixels)."""
    rs = np.random.RandomState((BASE_SEED + event * 1000 + cam) & 0x7FFFFFFF)
    if rs.rand() < p_none:
        return EventSpec(F)
    lo, hi = 10, max(11, F - 15)
    t0 = int(rs.randint(lo, hi))
    nb = 2 if rs.rand() < p_second else 1
    bubbles = []
    for _ in range(nb):
        for _try in range(64):
            cx = int(rs.randint(margin, W - margin))
            cy = int(rs.randint(margin, H - margin))
            if accept is None or accept(cx, cy):
                break
        contrast = -40 if rs.rand() < 0.5 else 40
        bubbles.append((cx, cy, contrast))
    flicker = None
    if rs.rand() < p_flicker:
        flicker = int(rs.randint(3, max(4, t0 - 3))) if t0 > 6 else None
    return EventSpec(F, t0, bubbles, flicker, dense_from=t0 + 3 if regime == "post_trigger_dense" else None,
                     hot_per_400=11 if regime == "noisy" else 0)


def camera_masks(W, H, cam):
    """Synthetic masks in the shape of cam_masks/40l-19: a fiducial mask (non-zero inside an ellipse that covers about
    80 % of the frame) for every camera and, for odd cameras (the ones that see the bellows), a bellows mask: a strip
    over the lowest part of the frame, 12 % of its area.  -> (fiducial uint8 [H,W] of 0 / 255, bellows or None)."""
    y, x = np.ogrid[:H, :W]
    fid = (((2 * x - W) * (2 * x - W)) * (H * H) + ((2 * y - H) * (2 * y - H)) * (W * W) <= (W * W) * (H * H)) .astype(np.uint8) * 255
    bel = None
    if cam % 2 == 1:
        bel = np.zeros((H, W), np.uint8)
        bel[(H * 80) // 100:(H * 95) // 100, W // 10:(W * 9) // 10] = 255
    return fid, bel


def write_masks(maskdir, W, H, ncams):
    """cam<N>_mask.bmp / cam<N>_bellows_mask.bmp as 1-bit BMP files (like the reference's cam0_mask.bmp) into `maskdir`
    (with a trailing separator, as L3Localizer concatenates it); returns an accept(cam) -> (cx, cy) -> bool factory for
    random_spec: bubbles only inside the fiducial ellipse and outside the bellows strip."""
    import os

    from PIL import Image

    os.makedirs(maskdir, exist_ok=True)
    masks = {}
    for c in range(ncams):
        fid, bel = camera_masks(W, H, c)
        Image.fromarray(fid).convert("1").save(os.path.join(maskdir, f"cam{c}_mask.bmp"))
        if bel is not None:
            Image.fromarray(bel).convert("1").save(os.path.join(maskdir, f"cam{c}_bellows_mask.bmp"))
        masks[c] = (fid, bel)

    def accept_for(cam, margin=60):
        fid, bel = masks[cam]

        def accept(cx, cy):
            # the whole tracked disc (radius <= ~20 px within the tracking window, plus the box centre) stays inside
            for dx, dy in ((0, 0), (margin, 0), (-margin, 0), (0, margin), (0, -margin)):
                xx, yy = min(max(cx + dx, 0), W - 1), min(max(cy + dy, 0), H - 1)
                if not fid[yy, xx] or (bel is not None and bel[yy, xx]):
                    return False
            return True

        return accept

    return accept_for


def render_event(W, H, spec, event, cam, xp="numpy", device=None, out=None, bg=None):
    """-> uint8 [F,H,W] stack (numpy array or torch tensor on `device`)."""
    be = _backend(xp)
    if bg is None:
        bg = background(W, H, BASE_SEED + cam, xp, device)
    if out is None:
        out = be.empty_u8((spec.F, H, W), device)
    x = be.arange(W, device)[None, :]
    y = be.arange(H, device)[:, None]
    ev_seed = BASE_SEED + event * 1000 + cam
    hot = None
    if spec.hot_per_400:
        hot = _hash32((((y * W + x) * 40503) & 0xFFFFFFFF) ^ _hash32_scalar(cam * 104729 + 17)) % 400 < spec.hot_per_400
    for f in range(spec.F):
        noise = frame_noise(W, H, ev_seed * 131 + f, xp, device)
        if hot is not None:  # constant in the training frames, U{-3..3} afterwards
            jit = _hash32((((y * W + x) * 2246822519) & 0xFFFFFFFF) ^ _hash32_scalar(ev_seed * 977 + f)) % 7 - 3
            noise = be.where(hot, jit if f >= 2 else jit * 0, noise)
        v = bg + noise
        if spec.dense_from is not None and (f >= spec.dense_from or f < 2):
            band = (x >= (W * 30) // 100) & (x < (W * 65) // 100) & (y >= 0)
            off = (-2 if f == 0 else 2) if f < 2 else (8 if (f - spec.dense_from) % 4 < 2 else -8)
            v = be.where(band, v + off, v)
        if spec.flicker is not None and f == spec.flicker:
            v = v + spec.flicker_adu
        if spec.t0 is not None and f >= spec.t0:
            k = f - spec.t0
            r2 = (4 + 3 * k) * (4 + 3 * k)  # (2*(2+1.5k))^2, compare in doubled coordinates
            for (cx, cy, contrast) in spec.bubbles:
                dx = 2 * (x - cx)
                dy = 2 * (y - cy)
                inside = (dx * dx + dy * dy) <= r2
                v = be.where(inside, v + contrast, v)
        out[f] = be.to_u8(be.clip(v, 0, 255))
    return out


def training_pairs(W, H, n_events, cam, F, xp="numpy", device=None, **kw):
    """Frames 0 and 1 of events 0..n_events-1 (quiet by construction) -> uint8 [2*n,H,W]."""
    be = _backend(xp)
    out = be.empty_u8((2 * n_events, H, W), device)
    bg = background(W, H, BASE_SEED + cam, xp, device)
    for e in range(n_events):
        ev_seed = BASE_SEED + e * 1000 + cam
        for f in (0, 1):
            v = bg + frame_noise(W, H, ev_seed * 131 + f, xp, device)
            out[2 * e + f] = be.to_u8(be.clip(v, 0, 255))
    return out


def long_stack(N, W, H, xp="numpy", device=None, out=None, first=0, seed=BASE_SEED):
    """One long contiguous stack for the kernel microbench (BASELINE configs[2], SURVEY 8d "Config 3"): background +
    per-frame noise as above, and a bright disc (+40 ADU, radius 2..33) that grows over the second half of every
    64-frame block at a position that changes from block to block.  Frames `first .. first+N-1` -> uint8 [N,H,W]."""
    be = _backend(xp)
    bg = background(W, H, seed, xp, device)
    if out is None:
        out = be.empty_u8((N, H, W), device)
    x = be.arange(W, device)[None, :]
    y = be.arange(H, device)[:, None]
    for k in range(N):
        f = first + k
        v = bg + frame_noise(W, H, seed * 977 + f, xp, device)
        ph = f % 64
        if ph >= 32:
            blk = f // 64
            cx, cy, r = 200 + (blk * 37) % max(1, W - 400), 150 + (blk * 53) % max(1, H - 300), 2 + ph - 32
            x0, x1, y0, y1 = max(cx - r, 0), min(cx + r + 1, W), max(cy - r, 0), min(cy + r + 1, H)
            dx = x[:, x0:x1] - cx
            dy = y[y0:y1] - cy
            sub = v[y0:y1, x0:x1]
            v[y0:y1, x0:x1] = be.where(dx * dx + dy * dy <= r * r, sub + 40, sub)
        out[k] = be.to_u8(be.clip(v, 0, 255))
    return out
