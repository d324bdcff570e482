"""`utils.datasets` (un-vendored in the reference; call sites configs/dataset/cub200.yaml:10-70): list-file image
datasets with one-hot targets, plus a synthetic variant for machines without the images (there are none in the
reference snapshot)."""
from __future__ import annotations

import os

import torch
from torch.utils.data import Dataset


class OneHot:
    def __init__(self, nclass: int):
        self.nclass = int(nclass)

    def __call__(self, label):
        t = torch.zeros(self.nclass)
        t[int(label)] = 1.0
        return t


def read_list(path: str):
    """`<relative/path.jpg> <label>` per line."""
    items = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                p, _, lab = line.rpartition(" ")
                items.append((p, int(lab)))
    return items


def jpeg_size(buf):
    """(height, width) from a JPEG file's frame header (the SOFn marker), without decoding; None when `buf` is not a JPEG this
    parser follows.  A `gpu_decode` worker needs the size to draw a training crop box and must not pay for a decode."""
    n = len(buf)
    if n < 4 or buf[0] != 0xFF or buf[1] != 0xD8:
        return None
    i = 2
    while i + 3 < n:
        if buf[i] != 0xFF:
            return None
        m = buf[i + 1]
        if m == 0xFF:                      # fill byte
            i += 1
            continue
        if m == 0xD8 or m == 0x01 or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        seg = (buf[i + 2] << 8) | buf[i + 3]
        if 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            if i + 8 < n:
                return (buf[i + 5] << 8) | buf[i + 6], (buf[i + 7] << 8) | buf[i + 8]
            return None
        if m == 0xDA or m == 0xD9:         # scan data before any frame header
            return None
        i += 2 + seg
    return None


def gpu_augmentation(transform):
    """The random part of a TRAINING transform list that the GPU pre-processing can take over: (RandomResizedCrop, RandomHorizontalFlip
    or None) when the list is RandomResizedCrop(bicubic) [-> RandomHorizontalFlip] -> ToTensor -> normalize (configs/dataset/
    cub200.yaml:13-23), None for the evaluation chain (Resize -> CenterCrop, whose geometry the trainer takes from the dataset
    config).  Anything else cannot be reproduced there and raises."""
    from utils import transforms as T
    from PIL import Image
    items = list(getattr(transform, "transforms", transform) or [])
    rrc = [t for t in items if isinstance(t, T.RandomResizedCrop)]
    if not rrc:
        return None
    flips = [t for t in items if isinstance(t, T.RandomHorizontalFlip)]
    other = [t for t in items if not isinstance(t, (T.RandomResizedCrop, T.RandomHorizontalFlip, T.ToTensor, T.Normalize))]
    if len(rrc) != 1 or len(flips) > 1 or other or items.index(rrc[0]) != 0 or rrc[0].interp != Image.BICUBIC \
            or rrc[0].size[0] != rrc[0].size[1]:
        raise ValueError("gpu_preprocess / gpu_decode: the training transform list must be RandomResizedCrop(square, bicubic) "
                         "[-> RandomHorizontalFlip] -> ToTensor -> normalize; set dataset.gpu_decode=false dataset.gpu_preprocess=false to run "
                         "any other list on CPU workers, as the reference does")
    return rrc[0], (flips[0] if flips else None)


class RawImageBatch:
    """Decoded, untransformed images of one batch for the GPU pre-processing path: `pixels` = the uint8 RGB bytes of all images
    back to back (image i is [h_i, w_i, 3]), `sizes` = [(h, w)].  Quacks enough like a tensor for the trainer's plumbing.
    `boxes` [B, 4] (top, left, height, width) / `flips` [B]: the draws of the training transforms, made in the loader worker with the
    CPU chain's own random calls; None for the evaluation chain."""

    def __init__(self, pixels: torch.Tensor, sizes, boxes=None, flips=None):
        self.pixels, self.sizes = pixels, list(sizes)
        self.boxes, self.flips = boxes, flips

    def size(self, dim=0):
        if dim != 0:
            raise IndexError("RawImageBatch only has a batch dimension")
        return len(self.sizes)

    def to(self, device, non_blocking=False):
        return RawImageBatch(self.pixels.to(device, non_blocking=non_blocking), self.sizes, self.boxes, self.flips)

    def pin_memory(self):
        return RawImageBatch(self.pixels.pin_memory(), self.sizes, self.boxes, self.flips)


def raw_collate(batch):
    """collate_fn of a `gpu_preprocess` dataset: images stay decoded uint8 of their own sizes (no CPU resize), targets and
    indices are stacked as usual."""
    imgs, targets, idxs = zip(*batch)
    boxes = flips = None
    if isinstance(imgs[0], tuple):         # (image, box, flip): a training dataset
        boxes = torch.as_tensor([im[1] for im in imgs], dtype=torch.int32)
        flips = torch.as_tensor([im[2] for im in imgs], dtype=torch.bool)
        imgs = [im[0] for im in imgs]
    pixels = torch.cat([im.reshape(-1) for im in imgs])
    targets = torch.stack([t if torch.is_tensor(t) else torch.as_tensor(t) for t in targets])
    return RawImageBatch(pixels, [tuple(im.shape[:2]) for im in imgs], boxes, flips), targets, torch.as_tensor(idxs)


class RawJpegBatch:
    """UNDECODED image files of one batch for the GPU decode path (`gpu_decode: true`): `data` = the files' bytes back to back in ONE
    uint8 host tensor (one shared-memory segment per batch between worker and trainer, not one per image), `lengths` = bytes per file.
    Stays on the host -- the entropy decode runs on host threads -- until the trainer hands `files` to
    `concepthash_amd.jpeg.GpuJpegDecoder`, which returns the RawImageBatch the GPU pre-processing takes."""

    def __init__(self, data: torch.Tensor, lengths, boxes=None, flips=None):
        self.data = data
        self.lengths = [int(n) for n in lengths]
        self.boxes, self.flips = boxes, flips      # training transforms' draws (see RawImageBatch), or None

    @property
    def files(self):
        out, o = [], 0
        for n in self.lengths:
            out.append(self.data[o:o + n])
            o += n
        return out

    def size(self, dim=0):
        if dim != 0:
            raise IndexError("RawJpegBatch only has a batch dimension")
        return len(self.lengths)

    def to(self, device, non_blocking=False):
        return self

    def pin_memory(self):
        return self


def jpeg_collate(batch):
    """collate_fn of a `gpu_decode` dataset: the items are file bytes; nothing is decoded or resized on the CPU."""
    files, targets, idxs = zip(*batch)
    boxes = flips = None
    if isinstance(files[0], tuple):
        boxes = torch.as_tensor([f[1] for f in files], dtype=torch.int32)
        flips = torch.as_tensor([f[2] for f in files], dtype=torch.bool)
        files = [f[0] for f in files]
    targets = torch.stack([t if torch.is_tensor(t) else torch.as_tensor(t) for t in targets])
    return RawJpegBatch(torch.cat(files), [f.numel() for f in files], boxes, flips), targets, torch.as_tensor(idxs)


class HashingDataset(Dataset):
    """Returns (image, target, index), as the trainers unpack it (trainers/coop.py:62).

    `gpu_decode=True` (dataset config key of the same name; implies GPU pre-processing): the worker only READS the file; the item is its
    bytes, the loader collates with `jpeg_collate`, and the trainer decodes the batch with `concepthash_amd.jpeg.GpuJpegDecoder` (host
    threads: Huffman entropy decode; GPU: inverse DCT, upsampling, colour conversion -- bytes bit-equal to PIL's) before the GPU
    pre-processing.

    `gpu_preprocess=True` (dataset config key of the same name): `transform` is NOT applied on the CPU worker; the item is the
    decoded uint8 [H, W, 3] image and the loader collates batches with `raw_collate`; the trainer then runs Resize -> CenterCrop ->
    ToTensor -> normalize on the GPU (`concepthash_amd.preprocess.GpuPreprocess`, bit-equal to the CPU chain)."""

    def __init__(self, root, filename="train.txt", transform=None, target_transform=None, num_classes=None, num_shots=0,
                 separate_multiclass=False, gpu_preprocess=False, gpu_decode=False, **kwargs):
        from utils.transforms import Compose
        self.gpu_decode = bool(gpu_decode)
        self.gpu_preprocess = bool(gpu_preprocess) or self.gpu_decode
        if self.gpu_decode:
            self.collate_fn = jpeg_collate
        elif self.gpu_preprocess:
            self.collate_fn = raw_collate
        self.root = root
        self.items = read_list(os.path.join(root, filename))
        if num_shots:
            per, keep = {}, []
            for it in self.items:
                if per.setdefault(it[1], 0) < num_shots:
                    per[it[1]] += 1
                    keep.append(it)
            self.items = keep
        self.transform = Compose(transform) if isinstance(transform, (list, tuple)) else transform
        self.target_transform = target_transform
        # a TRAINING transform list on a GPU path: the worker keeps the random draws (crop box, flip), the GPU does the arithmetic
        self.augment = gpu_augmentation(self.transform) if self.gpu_preprocess else None
        self.read_threads = int(kwargs.get("read_threads", 8))
        self.file_workers = bool(kwargs.get("file_workers", False))   # gpu_decode: DataLoader worker processes instead of in-process reads     # gpu_decode: threads of one batch's file reads (ch_io_read_files)
        self._paths = {}          # index -> resolved path (a worker resolves each file once)

    def __len__(self):
        return len(self.items)

    def _resolve(self, rel):
        for cand in (rel, os.path.join(self.root, rel), os.path.join(os.path.dirname(os.path.dirname(self.root)), rel)):
            if os.path.exists(cand):
                return cand
        raise FileNotFoundError(f"image '{rel}' not found (list root {self.root}); use dataset=synthetic_* without images")

    def _draw(self, h, w):
        """One image's training draws, with the random calls and in the order of the CPU chain (RandomResizedCrop.get_params, then
        RandomHorizontalFlip's torch.rand): the same worker seed gives the same boxes and flips as the reference's loader."""
        rrc, flip = self.augment
        box = rrc.get_params(w, h)
        return box, (bool(float(torch.rand(1)) < flip.p) if flip is not None else False)

    def _file_size(self, buf):
        """(h, w) of an undecoded file: the JPEG frame header, or PIL's lazy open for anything else"""
        hw = jpeg_size(buf)
        if hw is None:
            import io
            from PIL import Image
            w, h = Image.open(io.BytesIO(bytes(buf))).size
            hw = (h, w)
        return hw

    def _read_batch(self, indices):
        """`gpu_decode`: the files of a whole batch read back to back into ONE uint8 buffer (no decode, no per-item tensors, no collate)."""
        paths = []
        for i in indices:
            p = self._paths.get(i)
            if p is None:
                p = self._paths[i] = self._resolve(self.items[i][0])
            paths.append(p)
        # sizes and reads by the library's host helpers (ch_io_file_sizes / ch_io_read_files: POSIX reads on a few threads, the GIL
        # released for the whole batch) -- 256 opens + reads in Python hold the interpreter for milliseconds that the thread launching
        # GPU work needs
        import ctypes
        import numpy as np
        from concepthash_amd import _lib
        lib = _lib.load()
        n = len(paths)
        enc = [os.fsencode(p) for p in paths]
        cpaths = (ctypes.c_char_p * n)(*enc)
        sizes = np.zeros(n, dtype=np.int64)
        _lib.check(lib.ch_io_file_sizes(cpaths, n, sizes.ctypes.data), "ch_io_file_sizes")
        offsets = np.zeros(n, dtype=np.int64)
        np.cumsum(sizes[:-1], out=offsets[1:])
        lengths = sizes.tolist()
        data = torch.empty(int(sizes.sum()), dtype=torch.uint8)
        _lib.check(lib.ch_io_read_files(cpaths, n, offsets.ctypes.data, sizes.ctypes.data, data.data_ptr(), self.read_threads),
                   "ch_io_read_files")
        view = memoryview(data.numpy())
        targets = [self.target_transform(self.items[i][1]) if self.target_transform is not None else self.items[i][1] for i in indices]
        targets = torch.stack([t if torch.is_tensor(t) else torch.as_tensor(t) for t in targets])
        boxes = flips = None
        if self.augment is not None:
            draws, o = [], 0
            for n in lengths:
                draws.append(self._draw(*self._file_size(view[o:o + n])))
                o += n
            boxes = torch.as_tensor([d[0] for d in draws], dtype=torch.int32)
            flips = torch.as_tensor([d[1] for d in draws], dtype=torch.bool)
        return RawJpegBatch(data, lengths, boxes, flips), targets, torch.as_tensor(list(indices))

    def __getitem__(self, index):
        from PIL import Image
        if self.gpu_decode and isinstance(index, (list, tuple)):       # engine.dataloader hands a gpu_decode dataset index LISTS
            return self._read_batch(index)
        rel, lab = self.items[index]
        target = self.target_transform(lab) if self.target_transform is not None else lab
        if self.gpu_decode:
            import numpy as np
            buf = np.fromfile(self._resolve(rel), dtype=np.uint8)      # read only: no decode on the CPU
            item = torch.from_numpy(buf)
            if self.augment is not None:
                item = (item,) + self._draw(*self._file_size(memoryview(buf)))
            return item, target, index
        img = Image.open(self._resolve(rel)).convert("RGB")
        if self.gpu_preprocess:
            import numpy as np
            img = torch.from_numpy(np.array(img, dtype=np.uint8))          # decode only; resize / crop / normalise on the GPU
            if self.augment is not None:
                img = (img,) + self._draw(img.shape[0], img.shape[1])
        elif self.transform is not None:
            img = self.transform(img)
        return img, target, index


class SyntheticHashingDataset(Dataset):
    """Seeded N(0,1) 'post-normalisation' images with the label vector of a real list file (or uniform labels):
    exercises the whole encode-and-retrieve path where the images themselves are unavailable."""
    in_memory = True

    def __init__(self, nclass, size=0, root=None, filename=None, image_size=224, seed=0, dtype="float32", limit=0, **kwargs):
        self.nclass = int(nclass)
        if root is not None and filename is not None and os.path.exists(os.path.join(root, filename)):
            labels = torch.tensor([lab for _, lab in read_list(os.path.join(root, filename))], dtype=torch.int64)
        else:
            g = torch.Generator().manual_seed(seed + 17)
            labels = torch.randint(0, self.nclass, (int(size),), generator=g)
        if limit:
            labels = labels[:: max(1, len(labels) // int(limit))][: int(limit)]
        self.labels = labels
        self.image_size, self.seed = int(image_size), int(seed)
        self.dtype = getattr(torch, dtype)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + index)
        # class-dependent mean so that the codes carry some label signal even with random weights
        lab = int(self.labels[index])
        img = torch.randn(3, self.image_size, self.image_size, generator=g)
        img += 0.5 * torch.sin(torch.arange(3).view(3, 1, 1) * 1.7 + lab * 0.37)
        target = torch.zeros(self.nclass)
        target[lab] = 1.0
        return img.to(self.dtype), target, index


class DeviceRawLoader:
    """A loader-shaped iterable over decoded uint8 images that already sit in GPU memory (all of one size): yields
    `(RawImageBatch, one-hot targets, indices)` per batch, exactly what `raw_collate` hands the trainer for a `gpu_preprocess`
    dataset, without a DataLoader, worker processes or host copies.  For measuring the evaluator loop itself
    (`bench.py`'s `evaluator_inclusive`) and for tests; `pixels` is [N, H, W, 3] uint8, `labels` [N] int64, both on the device."""

    def __init__(self, pixels: torch.Tensor, labels: torch.Tensor, nclass: int, batch_size: int):
        assert pixels.dim() == 4 and pixels.shape[-1] == 3 and pixels.dtype == torch.uint8
        self.pixels, self.labels, self.nclass, self.batch_size = pixels, labels, int(nclass), int(batch_size)
        self.targets = torch.nn.functional.one_hot(labels.long(), self.nclass).float()
        self.index = torch.arange(pixels.shape[0], device=pixels.device)

    def __len__(self):
        return -(-self.pixels.shape[0] // self.batch_size)

    def __iter__(self):
        n, (h, w) = self.pixels.shape[0], self.pixels.shape[1:3]
        for b0 in range(0, n, self.batch_size):
            b1 = min(n, b0 + self.batch_size)
            yield RawImageBatch(self.pixels[b0:b1].reshape(-1), [(h, w)] * (b1 - b0)), self.targets[b0:b1], self.index[b0:b1]
