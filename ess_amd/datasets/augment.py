"""Image-branch augmentation on the device (reference: the albumentations pipeline of datasets/cityscapes_loader.py:39-74).

`DeviceAugmentation(height, width, ...)` draws, per sample, the random DECISIONS of the pipeline on the host -- which ops fire and
with what magnitude, exactly the probabilities / ranges the reference configures -- and applies them to the whole batch with one
kernel launch (`hip.augment_image_label`): HorizontalFlip(p=0.5), ShiftScaleRotate(scale_limit=(0, 0.5), rotate_limit=0,
shift_limit, p=0.5, border 0), PadIfNeeded(height, width, border 0), RandomCrop(height, width), GaussNoise(p=0.2, var 10..50),
RandomBrightnessContrast(p=0.5, limits 0.2), uint8 quantisation, ToTensor, label id -> trainId table.  Perspective(p=0.2) and the
Sharpen / Blur / MotionBlur group are not provided.  albumentations / cv2 are absent here: the interpolation arithmetic is
restated (oracle.augment_image_label), parity against the libraries is unpinned."""
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


class DeviceAugmentation:
    def __init__(self, height, width, shift_limit=0.1, id_lut=None, seed=0, augment=True):
        self.height, self.width, self.shift_limit, self.augment = height, width, shift_limit, augment
        self.generator = torch.Generator().manual_seed(seed)
        self.id_lut = id_lut

    def __call__(self, img, label=None):
        """img: [N, Hs, Ws] (uint8 or float, 0..255) on the device; label: [N, Hs, Ws] integer ids or None."""
        params = draw_params(img.shape[0], img.shape[1:], (self.height, self.width), self.shift_limit, self.generator, self.augment)
        lut = None if self.id_lut is None else self.id_lut.to(img.device, torch.int64).contiguous()
        return hip.augment_image_label(img.float().contiguous(), None if label is None else label.long().contiguous(),
                                       params.to(img.device), self.height, self.width, lut)
