"""Image-branch augmentation on the device (reference: the albumentations pipeline of datasets/cityscapes_loader.py:39-74).

`DeviceAugmentation(height, width, ...)` draws, per sample, the random DECISIONS of the pipeline on the host -- which ops fire and
with what magnitude, exactly the probabilities / ranges the reference configures -- and applies them to the whole batch with one
kernel launch (`hip.augment_image_label`): HorizontalFlip(p=0.5), ShiftScaleRotate(scale_limit=(0, 0.5), rotate_limit=0,
shift_limit, p=0.5, border 0), PadIfNeeded(height, width, border 0), RandomCrop(height, width), GaussNoise(p=0.2, var 10..50),
RandomBrightnessContrast(p=0.5, limits 0.2), uint8 quantisation, ToTensor, label id -> trainId table; and -- second stage,
`hip.augment_perspective_filter`, two more launches -- Perspective(p=0.2, scale (0.05, 0.1), keep_size) between the noise and the
brightness / contrast step, and OneOf([Sharpen, Blur(3), MotionBlur(3)], p=0.5) at the end, in the reference's order
(datasets/cityscapes_loader.py:39-58; parameter draws restated from albumentations 1.1.0, the version the reference pins).
albumentations / cv2 are absent here: the interpolation arithmetic is restated (oracle.augment_image_label,
oracle.augment_perspective_filter), parity against the libraries themselves is unpinned.  Known deviations inside that caveat:
(i) `perspective_matrix` floors a degenerate quadrilateral's target size to max(2, int(.)) where albumentations 1.1.0 widens the
quadrilateral in a loop until min_width / min_height >= 2 and recomputes the corner points -- draws that hit the floor differ;
(ii) the brightness / contrast step quantises with round-half-up where albumentations' uint8 LUT truncates (`astype(uint8)`): values
may differ by one grey level."""
import math

import torch

from .. import hip

PARAM_NAMES = ('flip', 'scale', 'dx', 'dy', 'pad_top', 'pad_left', 'crop_y', 'crop_x', 'alpha', 'beta', 'sigma', 'seed')


def draw_params(n, src_hw, out_hw, shift_limit=0.1, generator=None, augment=True):
    """[n, 12] parameter rows (PARAM_NAMES).  augment=False: centred pad / crop only (the reference's CenterCrop path)."""
    Hs, Ws = src_hw
    H, W = out_hw
    g = generator
    u = lambda lo=0.0, hi=1.0: lo + (hi - lo) * torch.rand(n, generator=g)  # noqa: E731
    pad_h, pad_w = max(H - Hs, 0), max(W - Ws, 0)
    p = torch.zeros(n, 12)
    p[:, 1], p[:, 8] = 1.0, 1.0
    p[:, 4], p[:, 5] = pad_h // 2, pad_w // 2                      # PadIfNeeded: centred
    Hp, Wp = Hs + pad_h, Ws + pad_w
    if not augment:
        p[:, 6], p[:, 7] = (Hp - H) // 2, (Wp - W) // 2           # CenterCrop
        return p
    p[:, 0] = (u() < 0.5).float()                                   # HorizontalFlip
    ssr = (u() < 0.5).float()                                       # ShiftScaleRotate
    p[:, 1] = 1.0 + ssr * u(0.0, 0.5)
    p[:, 2] = ssr * u(-shift_limit, shift_limit) * Ws
    p[:, 3] = ssr * u(-shift_limit, shift_limit) * Hs
    p[:, 6] = torch.floor(u() * (Hp - H + 1)).clamp(max=Hp - H)     # RandomCrop
    p[:, 7] = torch.floor(u() * (Wp - W + 1)).clamp(max=Wp - W)
    noise = (u() < 0.2).float()                                     # GaussNoise: var in (10, 50) on the uint8 scale
    p[:, 10] = noise * torch.sqrt(u(10.0, 50.0))
    p[:, 11] = torch.floor(u() * 16777216.0)
    bc = (u() < 0.5).float()                                        # RandomBrightnessContrast (limits 0.2, brightness_by_max)
    p[:, 8] = 1.0 + bc * u(-0.2, 0.2)
    p[:, 9] = bc * u(-0.2, 0.2) * 255.0
    return p


PARAM2_NAMES = ('persp',) + tuple(f'minv{i}' for i in range(9)) + ('max_w', 'max_h', 'alpha', 'beta', 'stencil') + \
    tuple(f'k{i}' for i in range(9))


def _order_points(pts):
    """albumentations Perspective._order_points: the two left-most points are (top-left, bottom-left), the others (tr, br)."""
    pts = sorted(pts, key=lambda q: q[0])
    (tl, bl) = (pts[0], pts[1]) if pts[0][1] < pts[1][1] else (pts[1], pts[0])
    (tr, br) = (pts[2], pts[3]) if pts[2][1] < pts[3][1] else (pts[3], pts[2])
    return [tl, tr, br, bl]


def perspective_matrix(points, H, W):
    """(inverse homography as 9 floats, max_w, max_h) of albumentations' Perspective (keep_size, fit_output=False) for the four
    jittered corner points (tl, tr, br, bl order after _order_points; pixels): the forward matrix maps the quadrilateral onto the
    rectangle [0, max_w] x [0, max_h] (cv2.getPerspectiveTransform), the warp needs its inverse."""
    tl, tr, br, bl = _order_points([list(map(float, q)) for q in points])
    dist = lambda a, b: math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2)  # noqa: E731
    max_w = max(2, int(max(dist(tr, tl), dist(br, bl))))
    max_h = max(2, int(max(dist(tr, br), dist(tl, bl))))
    src = [tl, tr, br, bl]
    dst = [[0.0, 0.0], [float(max_w), 0.0], [float(max_w), float(max_h)], [0.0, float(max_h)]]
    A = torch.zeros(8, 8, dtype=torch.float64)
    b = torch.zeros(8, dtype=torch.float64)
    for i, ((x, y), (u, v)) in enumerate(zip(src, dst)):
        A[2 * i] = torch.tensor([x, y, 1, 0, 0, 0, -u * x, -u * y], dtype=torch.float64)
        A[2 * i + 1] = torch.tensor([0, 0, 0, x, y, 1, -v * x, -v * y], dtype=torch.float64)
        b[2 * i], b[2 * i + 1] = u, v
    m = torch.cat([torch.linalg.solve(A, b), torch.ones(1, dtype=torch.float64)]).view(3, 3)
    minv = torch.linalg.inv(m)
    minv = minv / minv[2, 2]
    return minv.reshape(9).float(), max_w, max_h


def _line3(xs, ys, xe, ye):
    """cv2.line(kernel, (xs, ys), (xe, ye), 1, thickness=1) on a 3 x 3 grid: 8-connected Bresenham, iterated left to right."""
    k = torch.zeros(3, 3)
    if xe < xs:
        xs, ys, xe, ye = xe, ye, xs, ys
    dx, dy = xe - xs, ye - ys
    sy = 1 if dy >= 0 else -1
    dy = abs(dy)
    vert = dy > dx
    major, minor = (dy, dx) if vert else (dx, dy)
    err, x, y = major - 2 * minor, xs, ys
    k[y, x] = 1
    for _ in range(major):
        step_minor = err < 0
        err += -2 * minor + (2 * major if step_minor else 0)
        if vert:
            y += sy
            x += 1 if step_minor else 0
        else:
            x += 1
            y += sy if step_minor else 0
        k[y, x] = 1
    return k


def draw_params2(n, out_hw, generator=None, alpha_beta=None):
    """[n, 24] parameter rows of the second stage (PARAM2_NAMES): Perspective(p=0.2), the brightness / contrast values handed over
    from the first stage's draw (alpha_beta: [n, 2], or identity), OneOf([Sharpen, Blur(3), MotionBlur(3)], p=0.5)."""
    H, W = out_hw
    g = generator
    u = lambda lo=0.0, hi=1.0: float(lo + (hi - lo) * torch.rand((), generator=g))  # noqa: E731
    ri = lambda k: int(torch.randint(0, k, (), generator=g))  # noqa: E731
    p = torch.zeros(n, 24)
    p[:, 1], p[:, 5], p[:, 9] = 1.0, 1.0, 1.0
    p[:, 10], p[:, 11] = W, H
    p[:, 12] = 1.0
    p[:, 19] = 1.0
    if alpha_beta is not None:
        p[:, 12:14] = alpha_beta
    for i in range(n):
        if u() < 0.2:                                               # Perspective
            scale = u(0.05, 0.1)
            pts = torch.randn(4, 2, generator=g) * scale
            pts = torch.remainder(pts.abs(), 1.0)
            pts[1, 0] = 1.0 - pts[1, 0]                             # top right
            pts[2] = 1.0 - pts[2]                                   # bottom right
            pts[3, 1] = 1.0 - pts[3, 1]                             # bottom left
            pts[:, 0] *= W
            pts[:, 1] *= H
            minv, mw, mh = perspective_matrix(pts.tolist(), H, W)
            p[i, 0], p[i, 1:10], p[i, 10], p[i, 11] = 1.0, minv, mw, mh
        if u() < 0.5:                                               # OneOf(Sharpen, Blur(3), MotionBlur(3)), equal weights
            which = ri(3)
            if which == 0:
                a, light = u(0.2, 0.5), u(0.5, 1.0)
                k = torch.full((3, 3), -a)
                k[1, 1] = (1.0 - a) + a * (8.0 + light)
            elif which == 1:
                k = torch.full((3, 3), 1.0 / 9.0)
            else:
                xs, xe = ri(3), ri(3)
                if xs == xe:
                    ys = ri(3)
                    ye = (ys + 1 + ri(2)) % 3                       # random.sample(range(3), 2): two distinct rows
                else:
                    ys, ye = ri(3), ri(3)
                k = _line3(xs, ys, xe, ye)
                k = k / k.sum()
            p[i, 14], p[i, 15:24] = 1.0, k.reshape(9)
    return p


class DeviceAugmentation:
    def __init__(self, height, width, shift_limit=0.1, id_lut=None, seed=0, augment=True):
        self.height, self.width, self.shift_limit, self.augment = height, width, shift_limit, augment
        self.generator = torch.Generator().manual_seed(seed)
        self.id_lut = id_lut

    def __call__(self, img, label=None):
        """img: [N, Hs, Ws] (uint8 or float, 0..255) on the device; label: [N, Hs, Ws] integer ids or None."""
        params = draw_params(img.shape[0], img.shape[1:], (self.height, self.width), self.shift_limit, self.generator, self.augment)
        lut = None if self.id_lut is None else self.id_lut.to(img.device, torch.int64).contiguous()
        lab = None if label is None else label.long().contiguous()
        if not self.augment:
            return hip.augment_image_label(img.float().contiguous(), lab, params.to(img.device), self.height, self.width, lut)
        # two stages: geometry + noise, then Perspective -> brightness / contrast -> stencil (the reference's order); the
        # brightness / contrast values move to the second stage, and so does the id -> trainId table
        params2 = draw_params2(img.shape[0], (self.height, self.width), self.generator, alpha_beta=params[:, 8:10].clone())
        params[:, 8], params[:, 9] = 1.0, 0.0
        mid, mid_l = hip.augment_image_label(img.float().contiguous(), lab, params.to(img.device), self.height, self.width, None)
        return hip.augment_perspective_filter(mid, mid_l, params2.to(img.device), lut)
