"""`utils.transforms` (un-vendored in the reference; call sites configs/dataset/cub200.yaml:10-47) plus the four
torchvision eval transforms those configs name -- torchvision is not installed in the target image, so
`concepthash_amd.config.locate` maps `torchvision.transforms.{Resize, CenterCrop, ToTensor, Compose}` here.

`normalize_transform(norm)`: the reference's constants live in the missing module; `norm=3` is what the ConceptHash
config selects (configs/model/concept_hash_final_v1_nosa_apt.yaml:72-73) and is taken to be the CLIP statistics
(unpinned); `norm=2` ImageNet, `norm=1` 0.5/0.5, `norm=0` identity."""
from __future__ import annotations

import numpy as np
import torch
from PIL import Image

_NORMS = {
    0: ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)),
    1: ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),
    2: ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
    3: ((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)),
}


def interpolation(name: str):
    return {"nearest": Image.NEAREST, "bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC, "lanczos": Image.LANCZOS}[name]


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean).view(3, 1, 1)
        self.std = torch.tensor(std).view(3, 1, 1)

    def __call__(self, x):
        return (x - self.mean) / self.std


def normalize_transform(norm: int):
    return Normalize(*_NORMS[int(norm)])


class Resize:
    def __init__(self, size, interpolation=Image.BILINEAR):
        self.size, self.interp = size, interpolation

    def __call__(self, img):
        if isinstance(self.size, int):          # shorter side -> size, aspect kept; the long side TRUNCATES, as
            w, h = img.size                     # torchvision's _compute_resized_output_size does: int(size * long / short)
            if w <= h:
                nw, nh = self.size, max(1, int(self.size * h / w))
            else:
                nw, nh = max(1, int(self.size * w / h)), self.size
            return img.resize((nw, nh), self.interp)
        return img.resize((self.size[1], self.size[0]), self.interp)


class CenterCrop:
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        w, h = img.size
        th, tw = self.size
        left, top = int(round((w - tw) / 2.0)), int(round((h - th) / 2.0))
        return img.crop((left, top, left + tw, top + th))


class ToTensor:
    def __call__(self, img):
        a = np.array(img.convert("RGB"), dtype=np.uint8)      # a writable copy: torch.from_numpy warns about PIL's read-only buffer
        return torch.from_numpy(a).permute(2, 0, 1).float().div_(255.0)


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class RandomResizedCrop:
    """torchvision.transforms.RandomResizedCrop as the reference's train_dataset uses it (configs/dataset/cub200.yaml:13-19):
    area fraction U(0.08, 1), log-uniform aspect ratio in (3/4, 4/3), ten tries, then the centre-crop fallback."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation=Image.BILINEAR):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.scale, self.ratio, self.interp = tuple(scale), tuple(ratio), interpolation

    def get_params(self, w, h):
        import math
        area = w * h
        log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * float(torch.empty(1).uniform_(self.scale[0], self.scale[1]))
            ar = math.exp(float(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])))
            cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                top = int(torch.randint(0, h - ch + 1, (1,)))
                left = int(torch.randint(0, w - cw + 1, (1,)))
                return top, left, ch, cw
        in_ratio = w / h
        if in_ratio < self.ratio[0]:
            cw, ch = w, int(round(w / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            ch, cw = h, int(round(h * self.ratio[1]))
        else:
            cw, ch = w, h
        return (h - ch) // 2, (w - cw) // 2, ch, cw

    def __call__(self, img):
        w, h = img.size
        top, left, ch, cw = self.get_params(w, h)
        return img.crop((left, top, left + cw, top + ch)).resize((self.size[1], self.size[0]), self.interp)


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        return img.transpose(Image.FLIP_LEFT_RIGHT) if float(torch.rand(1)) < self.p else img
