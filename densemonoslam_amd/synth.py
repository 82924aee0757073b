"""Deterministic synthetic RGB-D streams (SURVEY.md §8(d) "Synthetic inputs").

Scene: an axis-aligned box room (6 × 3 × 4 m, camera inside) with three inner boxes, textured
with I(p) = 128 + 64·sin(7x)·sin(5y) + 32·sin(11z) per channel (phase offsets per channel), so
every pixel has intensity > 0 and image gradients above the tracker's gate.
Camera k: t_k = (0.8·sin(0.01k), 0.1·sin(0.023k), 0.6·cos(0.01k)), yaw 0.3·sin(0.008k).
Depth: exact ray-cast z in mm (u16), optional Gaussian noise σ = 1.5 mm·z² and 3 % dropped
pixels, both from PCG32 (seed 20260929 + camera id).  numpy only: this is host-side input
synthesis, not part of the measured path.
"""
import numpy as np

SEED = 20260929

ROOM_MIN = np.array([-3.0, -1.5, -2.0])
ROOM_MAX = np.array([3.0, 1.5, 2.0])
BOXES = [
    (np.array([-1.4, 0.3, 1.3]), np.array([-0.6, 1.5, 1.9])),
    (np.array([0.55, 0.7, 1.15]), np.array([1.25, 1.5, 1.85])),
    (np.array([-0.15, -0.05, 1.65]), np.array([0.35, 0.45, 1.95])),
]



class Scene:
    """A variant of the synthetic world: room, inner boxes, checker cell frequency, and a camera path.  `None` everywhere below means
    the default scene of SURVEY 8(d) (module constants, `pose`), whose outputs every committed fixture depends on."""

    def __init__(self, room_min, room_max, boxes, checker=25.0, pose_fn=None):
        self.room_min, self.room_max = np.asarray(room_min, np.float64), np.asarray(room_max, np.float64)
        self.boxes = [(np.asarray(a, np.float64), np.asarray(b, np.float64)) for a, b in boxes]
        self.checker = float(checker)
        self.pose_fn = pose_fn


def look_at(eye, target, down=(0.0, 1.0, 0.0)):
    """Camera-to-world pose (float64) at `eye` looking at `target`; camera axes: x right, y down, z forward."""
    eye, target = np.asarray(eye, np.float64), np.asarray(target, np.float64)
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(np.asarray(down, np.float64), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, eye
    return T


def _corner_pose(k, cam_id=0):
    """A camera that looks into the room corner (+x, +y, +z) - floor and two walls, three orthogonal planes, fill the view at
    1 - 2 m - from an eye point on a small loop (<= 1.3 cm and 0.4 degrees per frame); camera `cam_id` is phase-shifted like `pose`."""
    k = k + 37.0 * cam_id
    w, amp = 0.05, 0.25
    eye = np.array([1.55 + amp * np.sin(w * k), 0.25 + 0.1 * np.sin(1.7 * w * k), 0.65 + 0.8 * amp * np.cos(w * k)])
    target = np.array([2.6 + 0.15 * np.sin(0.8 * w * k), 1.15, 1.65])
    return look_at(eye, target)


# "cluttered corner": the default room seen towards one of its corners, with boxes on the floor and against both walls, coarse
# checker cells (31 cm).  Point-to-plane ICP is constrained in all six degrees of freedom down to W/32 x H/32 pixels, which is what
# the reference's inter-map fern verification (Ferns.cpp:344-383: 3 x 50 iterations on W/8 x H/8 thumbnails) needs to converge.
CORNER_SCENE = Scene([-3.0, -1.5, -2.0], [3.0, 1.5, 2.0],
                     [([2.2, 0.7, 1.2], [3.0, 1.5, 2.0]), ([1.6, 1.0, 0.9], [2.2, 1.5, 1.5]), ([2.5, 0.2, 0.4], [3.0, 0.7, 1.0]),
                      ([1.2, 1.2, 1.6], [1.9, 1.5, 2.0])], checker=10.0, pose_fn=_corner_pose)


K_640 = (528.0, 528.0, 320.0, 240.0)  # fx, fy, cx, cy (GPUTest.cpp:150-152)
K_KITTI = (718.856, 718.856, 607.19, 185.22)


# ---- PCG32 (XSH-RR 64/32), vectorised with LCG jump-ahead ----------------------------------
_PCG_MULT = np.uint64(6364136223846793005)


class PCG32:
    def __init__(self, seed, stream=0):
        self.inc = np.uint64((int(stream) << 1) | 1)
        self.state = np.uint64(0)
        self._step()
        self.state = np.uint64((int(self.state) + int(seed)) & 0xFFFFFFFFFFFFFFFF)
        self._step()

    def _step(self):
        with np.errstate(over="ignore"):
            self.state = self.state * _PCG_MULT + self.inc

    def uint32(self, n):
        """Next n outputs, bit-identical to n sequential pcg32_random_r calls."""
        n = int(n)
        with np.errstate(over="ignore"):
            # A[i], C[i] with state_i = A[i]*s0 + C[i]  (i = 0..n-1)
            A = np.ones(1, np.uint64)
            Cc = np.zeros(1, np.uint64)
            am, cm = _PCG_MULT, self.inc  # multiplier / increment of one step, then doubled
            while A.size < n:
                A = np.concatenate([A, A * am])
                Cc = np.concatenate([Cc, Cc * am + cm])
                cm = cm * (am + np.uint64(1))
                am = am * am
            A, Cc = A[:n], Cc[:n]
            old = A * self.state + Cc
            # advance the generator by n steps
            last = old[-1] if n else self.state
            if n:
                self.state = last * _PCG_MULT + self.inc
            xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
            rot = (old >> np.uint64(59)).astype(np.uint32)
            return (xorshifted >> rot) | (xorshifted << ((np.uint32(32) - rot) & np.uint32(31)))

    def uniform(self, n):
        return (self.uint32(n).astype(np.float64) + 0.5) / 4294967296.0

    def normal(self, n):
        m = (n + 1) // 2
        u = self.uniform(2 * m)
        r = np.sqrt(-2.0 * np.log(u[:m]))
        th = 2.0 * np.pi * u[m:]
        return np.concatenate([r * np.cos(th), r * np.sin(th)])[:n]


# ---- trajectory ---------------------------------------------------------------------------
def pose(k, cam_id=0):
    """Camera-to-world 4×4 (float64) of frame k; camera `cam_id` is phase-shifted."""
    k = k + 37.0 * cam_id
    t = np.array([0.8 * np.sin(0.01 * k), 0.1 * np.sin(0.023 * k), 0.6 * np.cos(0.01 * k)])
    yaw = 0.3 * np.sin(0.008 * k)
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    T[:3, 3] = t
    return T


# ---- ray casting --------------------------------------------------------------------------
def _slab(o, d, bmin, bmax):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (bmin - o) * inv
        t1 = (bmax - o) * inv
    tn = np.minimum(t0, t1).max(axis=-1)
    tf = np.maximum(t0, t1).min(axis=-1)
    return tn, tf


def texture(p, checker_freq=25.0):
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    out = np.empty(p.shape[:-1] + (3,), np.float64)
    # The smooth SURVEY texture alone has image gradients of ~1 level/pixel at 1.4 m, far below the
    # photometric gate of the tracker (|Sobel| >= 40, RGBDOdometry.cpp:445), so a 3-D checker of
    # 12.6 cm cells is added: its edges give the >= 11 levels/pixel the gate asks for.
    cf = checker_freq
    checker = np.sign(np.sin(cf * x + 0.3) * np.sin(cf * y + 1.1) * np.sin(cf * z + 2.0))
    for ch, ph in enumerate((0.0, 0.9, 1.7)):
        out[..., ch] = (128 + 40 * np.sin(7 * x + ph) * np.sin(5 * y + 0.5 * ph) + 20 * np.sin(11 * z + 2 * ph)
                        + (45 - 6 * ch) * checker)
    return out


def render(T, width=640, height=480, K=K_640, scene=None):
    """Exact depth (float64 metres, camera z) and RGB (u8) for camera-to-world pose T."""
    room_min, room_max, boxes, cf = (ROOM_MIN, ROOM_MAX, BOXES, 25.0) if scene is None else (scene.room_min, scene.room_max, scene.boxes, scene.checker)
    fx, fy, cx, cy = K
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dirs_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)
    R, o = T[:3, :3], T[:3, 3]
    d = dirs_c @ R.T
    _, tf = _slab(o, d, room_min, room_max)  # inside the room: exit distance
    tbest = tf
    for bmin, bmax in boxes:
        tn, tfb = _slab(o, d, bmin, bmax)
        hit = (tn <= tfb) & (tn > 1e-6) & (tn < tbest)
        tbest = np.where(hit, tn, tbest)
    p = o + d * tbest[..., None]
    rgb = np.clip(np.rint(texture(p, cf)), 1, 255).astype(np.uint8)
    return tbest, rgb  # depth along camera z equals t because dirs_c.z == 1


def frame(k, cam_id=0, width=640, height=480, K=K_640, noise=True, rng=None, max_depth_mm=65535, scene=None):
    """(depth u16 mm, rgb u8 H×W×3, pose 4×4 float64) for frame k of camera cam_id."""
    T = pose(k, cam_id) if scene is None or scene.pose_fn is None else scene.pose_fn(k, cam_id)
    z, rgb = render(T, width, height, K, scene)
    n = width * height
    zmm = z * 1000.0
    if noise:
        if rng is None:
            rng = PCG32(SEED + cam_id, stream=k)
        zmm = zmm + (1.5 * z * z) * rng.normal(n).reshape(height, width)
        drop = rng.uniform(n).reshape(height, width) < 0.03
    else:
        drop = np.zeros((height, width), bool)
    d = np.clip(np.rint(zmm), 0, max_depth_mm).astype(np.uint16)
    d[drop] = 0
    return d, rgb, T


def rgba(rgb):
    return np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
