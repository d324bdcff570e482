"""Image pre-processing: Resize(256, bicubic) -> CenterCrop(224) -> ToTensor -> normalize
(reference configs/dataset/cub200.yaml:31-47, torchvision transforms on PIL images).

CPU: oracle/preprocess_oracle.py (numpy restatement of the Pillow resampler) is PINNED against Pillow itself, and
utils.transforms' Resize follows torchvision's truncating size rule.
GPU: csrc/preprocess.hip through the C-ABI equals the PIL chain of utils.transforms BIT FOR BIT (fp32 output), and its bf16
output is the RNE rounding of that -- tolerance 0."""
import numpy as np
import pytest
import torch
from PIL import Image

SIZES = [(375, 500), (500, 375), (333, 500), (64, 48), (100, 731), (256, 256), (257, 300), (1200, 900), (229, 1000)]


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 90 * np.sin(yy[..., None] / (7.0 + seed) + np.arange(3)) * np.cos(xx[..., None] / (11.0 + seed))
    return np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)


def _pil_chain(img, resize, crop, norm):
    from utils import transforms as T
    chain = T.Compose([T.Resize(resize, T.interpolation("bicubic")), T.CenterCrop(crop), T.ToTensor(), T.normalize_transform(norm)])
    return chain(Image.fromarray(img))


@pytest.mark.parametrize("h,w", SIZES)
def test_oracle_resampler_is_bit_equal_to_pillow(h, w):
    from oracle import preprocess_oracle as po
    img = _image(h, w, h % 7)
    nw, nh = po.resized_size(w, h, 256)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
    assert np.array_equal(po.resize_bicubic(img, nw, nh), ref)
    out = po.preprocess(img, 256, 224, (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))
    assert np.array_equal(out, _pil_chain(img, 256, 224, 3).numpy())


def test_resize_truncates_the_long_side_like_torchvision():
    from utils import transforms as T
    # 500 x 333: 256 * 500 / 333 = 384.38 -> 384; 731 x 100: 256 * 731 / 100 = 1871.36 -> 1871;
    # 453 x 300: 256 * 453 / 300 = 386.56 -> 386 (rounding would give 387)
    for (w, h), want in (((500, 333), (384, 256)), ((100, 731), (256, 1871)), ((453, 300), (386, 256))):
        img = Image.fromarray(np.zeros((h, w, 3), np.uint8))
        assert T.Resize(256, Image.BICUBIC)(img).size == want


# ---- the training chain: RandomResizedCrop(224, bicubic) -> RandomHorizontalFlip -> ToTensor -> normalize (configs/dataset/cub200.yaml:13-23)

def _train_chain(norm=3):
    from utils import transforms as T
    return [T.RandomResizedCrop(224, interpolation=T.interpolation("bicubic")), T.RandomHorizontalFlip(), T.ToTensor(), T.normalize_transform(norm)]


def _write_dataset(root, sizes, quality=90, progressive_every=0):
    import os
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    lines = []
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(_image(h, w, i)).save(os.path.join(root, "img", f"{i}.jpg"), "JPEG", quality=quality,
                                               progressive=bool(progressive_every and i % progressive_every == 0))
        lines.append(f"img/{i}.jpg {i % 5}\n")
    open(os.path.join(root, "train.txt"), "w").write("".join(lines))


def test_tap_bound_covers_every_output_index():
    """`ch_preprocess(max_taps=...)` picks kernels that hold their coefficients in 8 or 16 registers: the bound the planner passes
    (Pillow's ksize) must cover the taps of every output index of every image, for the evaluation geometry and for crop boxes."""
    from concepthash_amd.preprocess import _resized_size, _row_bounds, _taps
    rng = np.random.default_rng(0)
    for _ in range(300):
        h, w = int(rng.integers(8, 3000)), int(rng.integers(8, 3000))
        nw, _ = _resized_size(w, h, 256)
        bound = _taps(w, nw)
        assert bound % 2 == 1 and bound >= 5
        assert max(_row_bounds(w, nw, xx)[1] for xx in range(nw)) <= bound
        bw = int(rng.integers(1, w + 1))                       # a crop box resized to 224 columns
        assert max(_row_bounds(bw, 224, xx)[1] for xx in range(224)) <= _taps(bw, 224)
    assert _taps(500, 341) == 7 and _taps(375, 256) == 7 and _taps(300, 600) == 5 and _taps(1024, 256) == 17


def test_jpeg_size_reads_the_frame_header(tmp_path):
    import io
    from utils.datasets import jpeg_size
    for i, (h, w) in enumerate(SIZES):
        for kw in ({}, {"progressive": True}, {"optimize": True}, {"subsampling": 0}):
            bio = io.BytesIO()
            Image.fromarray(_image(h, w, i)).save(bio, "JPEG", quality=80, **kw)
            assert jpeg_size(memoryview(bio.getvalue())) == (h, w)
    bio = io.BytesIO()
    Image.fromarray(_image(40, 56, 0)[..., 0]).save(bio, "JPEG")                      # greyscale
    assert jpeg_size(bio.getvalue()) == (40, 56)
    bio = io.BytesIO()
    Image.fromarray(_image(40, 56, 0)).save(bio, "PNG")
    assert jpeg_size(bio.getvalue()) is None and jpeg_size(b"") is None and jpeg_size(b"\xff\xd8\xff") is None


def test_training_draws_are_the_cpu_chains_and_the_oracle_reproduces_its_images(tmp_path):
    """The loader worker of a GPU-path TRAINING dataset draws the crop box and the flip with the CPU chain's random calls: with the same
    seed the boxes are the ones RandomResizedCrop / RandomHorizontalFlip would have used, and crop-box resize + flip of the decoded
    image (the arithmetic the GPU kernel restates, here through the oracle) gives the CPU chain's tensors bit for bit."""
    from oracle import preprocess_oracle as po
    from utils.datasets import HashingDataset, OneHot
    from utils.transforms import _NORMS
    root = str(tmp_path)
    sizes = SIZES[:6]
    _write_dataset(root, sizes)
    cpu = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5))
    raw = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5), gpu_preprocess=True)
    jpg = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5), gpu_decode=True)
    assert cpu.augment is None and raw.augment is not None and jpg.augment is not None
    torch.manual_seed(123)
    want = [cpu[i][0] for i in range(len(sizes))]
    torch.manual_seed(123)
    items = [raw[i][0] for i in range(len(sizes))]
    torch.manual_seed(123)
    batch, targets, index = jpg[list(range(len(sizes)))]            # the batch-level read of a gpu_decode dataset
    torch.manual_seed(123)
    single = [jpg[i][0] for i in range(len(sizes))]
    flips = []
    for i, (img, box, flip) in enumerate(items):
        assert tuple(batch.boxes[i].tolist()) == tuple(box) == tuple(single[i][1]) and bool(batch.flips[i]) == flip == single[i][2]
        top, left, bh, bw = box
        assert 0 <= top and 0 <= left and top + bh <= sizes[i][0] and left + bw <= sizes[i][1]
        got = po.preprocess_train(img.numpy(), box, flip, 224, *_NORMS[3])
        assert np.array_equal(got, want[i].numpy()), i
        flips.append(flip)
    assert targets.shape == (len(sizes), 5) and index.tolist() == list(range(len(sizes)))
    # a list the GPU cannot reproduce is refused when the dataset is built, not silently replaced by the evaluation chain
    from utils import transforms as T
    with pytest.raises(ValueError):
        HashingDataset(root, "train.txt", transform=[T.RandomResizedCrop(224, interpolation=T.interpolation("bilinear")), T.ToTensor()],
                       gpu_preprocess=True)
    with pytest.raises(ValueError):
        HashingDataset(root, "train.txt", transform=[T.Resize(256), T.RandomResizedCrop(224, interpolation=T.interpolation("bicubic"))],
                       gpu_decode=True)


def test_file_batch_loader_reads_what_a_dataloader_would_hand_over(tmp_path):
    """A `gpu_decode` dataset is loaded by `engine.FileBatchLoader` -- in-process reads through `ch_io_file_sizes` / `ch_io_read_files`,
    no worker processes -- and must hand over the batches `DataLoader(d, bs, shuffle, drop_last, num_workers=0)` would: same index
    order under the same seed, same targets, the files' exact bytes; a missing file is named in the error."""
    import ctypes
    import os
    import engine
    from concepthash_amd import _lib
    from torch.utils.data import DataLoader
    from utils.datasets import HashingDataset, OneHot
    root = str(tmp_path)
    sizes = [(40 + 3 * i, 64 + 5 * i) for i in range(11)]
    _write_dataset(root, sizes)
    cpu = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5))
    jpg = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5), gpu_decode=True, read_threads=3)
    fl = engine.dataloader(jpg, 4, shuffle=True, drop_last=True)
    assert isinstance(fl, engine.FileBatchLoader) and fl.num_workers == 0 and len(fl) == 2
    for epoch in range(2):                                           # the same loader object, iterated again
        torch.manual_seed(31 + epoch)
        want = [(idx.tolist(), tgt) for _, tgt, idx in DataLoader(cpu, 4, shuffle=True, drop_last=True, num_workers=0)]
        torch.manual_seed(31 + epoch)
        got = list(fl)
        assert [b[2].tolist() for b in got] == [w[0] for w in want]
        for (raw, tgt, idx), (_, wt) in zip(got, want):
            assert torch.equal(tgt, wt) and raw.boxes.shape == (4, 4) and raw.flips.shape == (4,)
            for f, i in zip(raw.files, idx.tolist()):
                assert bytes(f.numpy()) == open(os.path.join(root, "img", f"{i}.jpg"), "rb").read()
    seq = engine.dataloader(jpg, 4, shuffle=False, drop_last=False)
    assert [b[2].tolist() for b in seq] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]
    # the host helpers themselves: sizes, a short destination offset table, a missing file
    lib = _lib.load()
    paths = [os.path.join(root, "img", f"{i}.jpg").encode() for i in range(3)] + [os.path.join(root, "img", "nope.jpg").encode()]
    arr = (ctypes.c_char_p * 4)(*paths)
    out = (ctypes.c_int64 * 4)()
    assert lib.ch_io_file_sizes(arr, 3, out) == 0 and list(out)[:3] == [os.path.getsize(p) for p in paths[:3]]
    assert lib.ch_io_file_sizes(arr, 4, out) != 0 and b"nope.jpg" in lib.ch_last_error()
    os.remove(os.path.join(root, "img", "5.jpg"))
    with pytest.raises(RuntimeError, match="5.jpg"):
        list(engine.dataloader(jpg, 4, shuffle=False, drop_last=False))


@pytest.mark.gpu
def test_gpu_training_chain_equals_the_pil_chain_bit_for_bit(tmp_path):
    """RandomResizedCrop -> RandomHorizontalFlip -> ToTensor -> normalize on the GPU (crop box + flip per image) against the CPU chain
    run with the same seed, through both GPU data paths: decoded images (gpu_preprocess) and undecoded files (gpu_decode).  fp32
    outputs are EQUAL; bf16 is the RNE rounding."""
    from concepthash_amd.jpeg import GpuJpegDecoder
    from concepthash_amd.preprocess import GpuPreprocess
    from utils.datasets import HashingDataset, OneHot, raw_collate
    from utils.transforms import _NORMS
    dev = torch.device("cuda:0")
    root = str(tmp_path)
    sizes = SIZES + [(224, 224), (30, 40)]               # a box the size of the output, and up-scaling of a tiny image
    _write_dataset(root, sizes, progressive_every=4)     # baseline and progressive files
    n = len(sizes)
    cpu = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5))
    raw = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5), gpu_preprocess=True)
    jpg = HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5), gpu_decode=True)
    pre32 = GpuPreprocess(256, 224, *_NORMS[3], out_dtype=torch.float32, device=dev)
    pre16 = GpuPreprocess(256, 224, *_NORMS[3], out_dtype=torch.bfloat16, device=dev)
    dec = GpuJpegDecoder(device=dev)
    for seed in (5, 77):
        torch.manual_seed(seed)
        want = torch.stack([cpu[i][0] for i in range(n)])
        torch.manual_seed(seed)
        b, _, _ = raw_collate([raw[i] for i in range(n)])
        got = pre32(b.pixels.to(dev), b.sizes, boxes=b.boxes, flips=b.flips).cpu()
        assert torch.equal(got, want), float((got - want).abs().max())
        assert torch.equal(pre16(b.pixels.to(dev), b.sizes, boxes=b.boxes, flips=b.flips).cpu(), want.to(torch.bfloat16))
        assert bool(b.flips.any()) and not bool(b.flips.all())
        torch.manual_seed(seed)
        jb, _, _ = jpg[list(range(n))]
        staged = dec.host_stage(jb)
        assert torch.equal(staged.boxes, b.boxes) and torch.equal(staged.flips, b.flips)
        pixels, psizes = staged.finish()
        got = pre32(pixels, psizes, boxes=staged.boxes, flips=staged.flips).cpu()
        assert torch.equal(got, want)
    with pytest.raises(ValueError):
        pre32(b.pixels.to(dev), b.sizes, boxes=b.boxes + 400, flips=b.flips)     # a box outside its image


@pytest.mark.gpu
def test_gpu_preprocess_equals_the_pil_chain_bit_for_bit():
    from concepthash_amd.preprocess import GpuPreprocess
    dev = torch.device("cuda:0")
    imgs = [_image(h, w, i) for i, (h, w) in enumerate(SIZES)]
    pixels = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).to(dev)
    sizes = [im.shape[:2] for im in imgs]
    for norm in (3, 2):
        from utils.transforms import _NORMS
        mean, std = _NORMS[norm]
        want = torch.stack([_pil_chain(im, 256, 224, norm) for im in imgs])
        got32 = GpuPreprocess(256, 224, mean, std, out_dtype=torch.float32, device=dev)(pixels, sizes).cpu()
        assert got32.shape == want.shape == (len(imgs), 3, 224, 224)
        assert torch.equal(got32, want), float((got32 - want).abs().max())
        got16 = GpuPreprocess(256, 224, mean, std, out_dtype=torch.bfloat16, device=dev)(pixels, sizes).cpu()
        assert torch.equal(got16, want.to(torch.bfloat16))
    # other geometry: Resize(160) -> CenterCrop(128), up-scaling of the small image included
    want = torch.stack([_pil_chain(im, 160, 128, 1) for im in imgs])
    got = GpuPreprocess(160, 128, *_NORMS[1], out_dtype=torch.float32, device=dev)(pixels, sizes).cpu()
    assert torch.equal(got, want)
    # down-scaling beyond the kernels' tap limit (a 17-megapixel image to 256: 16x): that image goes through Pillow on the host, the
    # others through the kernels, in the same call -- the batch still equals the PIL chain
    big = _image(4200, 4100, 5)
    mixed = [imgs[0], big, imgs[1]]
    pre = GpuPreprocess(256, 224, *_NORMS[3], out_dtype=torch.float32, device=dev)
    got = pre(torch.from_numpy(np.concatenate([im.reshape(-1) for im in mixed])).to(dev), [im.shape[:2] for im in mixed]).cpu()
    assert torch.equal(got, torch.stack([_pil_chain(im, 256, 224, 3) for im in mixed])) and pre.host_routed == 1
    # the same for a training crop box that large (RandomResizedCrop of nearly the whole image), flipped
    from utils import transforms as T
    box = (100, 50, 3900, 4000)                                         # top, left, height, width
    ref = Image.fromarray(big).crop((box[1], box[0], box[1] + box[3], box[0] + box[2])).resize((224, 224), Image.BICUBIC).transpose(Image.FLIP_LEFT_RIGHT)
    want = T.normalize_transform(3)(T.ToTensor()(ref))
    got = pre(torch.from_numpy(big.reshape(-1)).to(dev), [big.shape[:2]], boxes=[box], flips=[True]).cpu()
    assert torch.equal(got[0], want) and pre.host_routed == 2
    with pytest.raises(TypeError):
        GpuPreprocess(256, 224, device=dev)(pixels.float(), sizes)


@pytest.mark.gpu
def test_gpu_preprocess_kernel_forms_agree_with_pillow():
    """`ch_preprocess` picks its kernels by the batch: the dword forms (crop % 4 == 0; coefficients in registers for up to 8 or up to 16
    horizontal taps) or the byte forms (any crop, up to 64 taps).  Every combination against the PIL chain, evaluation and training
    geometry (crop boxes inside wider images, flipped or not), fp32 and bf16."""
    from concepthash_amd.preprocess import GpuPreprocess, _taps
    from utils import transforms as T
    from utils.transforms import _NORMS
    dev = torch.device("cuda:0")
    batches = {8: [(375, 500), (500, 375), (333, 500), (64, 48), (257, 300), (256, 256), (100, 731)],     # <= 8 taps
               16: [(600, 800), (820, 640), (375, 500), (880, 881)],                                     # 9 .. 16 taps
               64: [(1200, 900), (375, 500), (2000, 1500)]}                                              # more: byte-form horizontal pass
    for bound, sizes in batches.items():
        imgs = [_image(h, w, 3 + i) for i, (h, w) in enumerate(sizes)]
        pixels = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).to(dev)
        for crop in (224, 222, 256):                                                  # 222: byte forms of both passes
            resize = 256 if crop < 256 else 292
            pre = GpuPreprocess(resize, crop, *_NORMS[3], out_dtype=torch.float32, device=dev)
            taps = pre.plan(sizes)[4]
            assert {8: 0, 16: 8, 64: 16}[bound] < taps <= bound, (bound, taps)
            want = torch.stack([_pil_chain(im, resize, crop, 3) for im in imgs])
            got = pre(pixels, sizes).cpu()
            assert torch.equal(got, want), (bound, crop, float((got - want).abs().max()))
            got16 = GpuPreprocess(resize, crop, *_NORMS[3], out_dtype=torch.bfloat16, device=dev)(pixels, sizes).cpu()
            assert torch.equal(got16, want.to(torch.bfloat16)), (bound, crop)
        # training geometry: a box per image, every second one flipped
        rng = np.random.default_rng(bound)
        boxes, flips, want = [], [], []
        for i, (im, (h, w)) in enumerate(zip(imgs, sizes)):
            bh, bw = int(rng.integers(h // 3, h + 1)), int(rng.integers(w // 3, w + 1))
            top, left = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
            boxes.append((top, left, bh, bw))
            flips.append(i % 2 == 1)
            ref = Image.fromarray(im).crop((left, top, left + bw, top + bh)).resize((224, 224), Image.BICUBIC)
            if flips[-1]:
                ref = ref.transpose(Image.FLIP_LEFT_RIGHT)
            want.append(T.normalize_transform(3)(T.ToTensor()(ref)))
        pre = GpuPreprocess(256, 224, *_NORMS[3], out_dtype=torch.float32, device=dev)
        assert pre.plan_boxes(sizes, boxes, flips)[4] == max(_taps(b[3], 224) for b in boxes)
        got = pre(pixels, sizes, boxes=boxes, flips=flips).cpu()
        assert torch.equal(got, torch.stack(want)), bound
    # an image whose rows start at every alignment (odd width, odd offset in the batch): the 16-byte staging of the horizontal pass
    sizes = [(61, 77), (59, 81), (300, 403), (301, 405), (64, 67)]
    imgs = [_image(h, w, 11 + i) for i, (h, w) in enumerate(sizes)]
    pixels = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).to(dev)
    for crop in (32, 224, 8):
        resize = max(crop, 40)
        got = GpuPreprocess(resize, crop, *_NORMS[3], out_dtype=torch.float32, device=dev)(pixels, sizes).cpu()
        assert torch.equal(got, torch.stack([_pil_chain(im, resize, crop, 3) for im in imgs])), crop


@pytest.mark.gpu
def test_gpu_preprocess_feeds_the_encoder():
    """decoded bytes -> GPU pre-processing -> ch_encode: same codes as the CPU chain's tensors given to the encoder."""
    from concepthash_amd.encoder import ConceptHashEncoder
    from concepthash_amd.preprocess import GpuPreprocess
    from concepthash_amd import synthetic as syn
    dev = torch.device("cuda:0")
    cfg = dict(syn.CONFIGS["vit_s16"])
    cfg["L"] = 2
    sd = syn.synthetic_state_dict(cfg, nbit=32, nclass=10)
    imgs = [_image(h, w, i) for i, (h, w) in enumerate(SIZES[:4])]
    pixels = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).to(dev)
    batch = GpuPreprocess(256, 224, device=dev)(pixels, [im.shape[:2] for im in imgs])
    enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=4, device=dev)
    a = enc.encode(batch)["codes"]
    b = enc.encode(torch.stack([_pil_chain(im, 256, 224, 3) for im in imgs]).to(dev).to(torch.bfloat16))["codes"]
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_trainer_with_gpu_preprocess_gives_the_codes_of_the_cpu_chain(tmp_path):
    """`dataset.gpu_preprocess: true` end to end: list-file dataset of JPEGs of different sizes -> DataLoader workers decode only
    -> RawImageBatch -> COOPTrainer runs ch_preprocess + ch_encode.  Codes are bit-equal to the run whose CPU workers apply the
    torchvision-style transform chain (the reference's loader, configs/dataset/cub200.yaml:31-47)."""
    from concepthash_amd import config as cfglib
    from concepthash_amd import synthetic as syn
    from trainers.coop import COOPTrainer
    from utils import transforms as T
    from utils.datasets import HashingDataset, OneHot
    root = tmp_path / "d"
    (root / "img").mkdir(parents=True)
    lines = []
    for i, (h, w) in enumerate(SIZES[:7]):
        Image.fromarray(_image(h, w, i)).save(root / "img" / f"{i}.png")        # lossless: both runs decode the same pixels
        lines.append(f"img/{i}.png {i % 3}")
    (root / "test.txt").write_text("\n".join(lines) + "\n")
    cfg = dict(syn.CONFIGS["vit_s16"])
    cfg["L"] = 2
    sd = syn.synthetic_state_dict(cfg, nbit=32, nclass=3)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            from concepthash_amd.encoder import ConceptHashEncoder
            self.enc = ConceptHashEncoder(sd, heads=cfg["heads"], max_batch=4, device=torch.device("cuda:0"))

        def forward(self, x):
            out = self.enc.encode(x, want=("codes", "logits_cont", "logits_bin"))
            return None, out

    class Crit(torch.nn.Module):
        losses = {}

        def forward(self, out, y):
            return out["codes"].sum() * 0

    chain = [T.Resize(256, T.interpolation("bicubic")), T.CenterCrop(224), T.ToTensor(), T.normalize_transform(3)]
    codes = {}
    for mode in (False, True):
        conf = cfglib.DictConfig(device="cuda", batch_size=4, model=cfglib.DictConfig(),
                                 dataset=cfglib.DictConfig(multiclass=False, resize=256, crop=224, norm=3, gpu_preprocess=mode))
        tr = COOPTrainer(conf)
        tr.dataset = {"train": [], "db": [], "test": HashingDataset(str(root), "test.txt", transform=chain, target_transform=OneHot(3),
                                                                     gpu_preprocess=mode)}
        tr.load_dataloader()
        tr.model, tr.criterion = Model(), Crit()
        meters, out = tr.inference_one_epoch("test", True)
        codes[mode] = out["codes"]
        assert out["codes"].shape == (7, 32) and out["labels"].shape == (7, 3)
    assert torch.equal(codes[True], codes[False])


@pytest.mark.gpu
def test_trainer_train_loader_on_the_gpu_paths_sees_the_cpu_loaders_batches(tmp_path, monkeypatch):
    """`dataset.gpu_decode / gpu_preprocess: true` on the TRAINING split, through COOPTrainer's own plumbing (shuffling loader ->
    iterate_loader -> compute_features_one_batch): with the same seed every batch holds the same images, crops and flips as the CPU
    loader's (in-process loading, so that one random stream feeds the sampler and the transforms in both runs); the GPU paths hand the
    model the bf16 rounding of the CPU chain's fp32 tensors."""
    import engine
    from concepthash_amd import config as cfglib
    from trainers.coop import COOPTrainer
    from utils.datasets import HashingDataset, OneHot
    monkeypatch.setattr(engine, "default_workers", 0)
    root = str(tmp_path)
    sizes = SIZES + [(300, 300)]
    _write_dataset(root, sizes)
    seen = {}

    class Model(torch.nn.Module):
        def forward(self, x):
            return None, {"codes": x.float().mean(dim=(1, 2, 3)).view(-1, 1)}

    for mode in ("cpu", "gpu_preprocess", "gpu_decode"):
        conf = cfglib.DictConfig(device="cuda", batch_size=4, model=cfglib.DictConfig(),
                                 dataset=cfglib.DictConfig(multiclass=False, resize=256, crop=224, norm=3))
        tr = COOPTrainer(conf)
        tr.dataset = {"test": [], "db": [], "train": HashingDataset(root, "train.txt", transform=_train_chain(), target_transform=OneHot(5),
                                                                     gpu_preprocess=mode == "gpu_preprocess", gpu_decode=mode == "gpu_decode")}
        tr.load_dataloader()
        tr.model = Model()
        torch.manual_seed(2024)
        batches = []
        for data in tr.iterate_loader(tr.dataloader["train"]):
            (image, labels, index), _ = tr.compute_features_one_batch(data)
            batches.append((image.detach().to(torch.bfloat16).cpu(), labels.cpu(), index.cpu()))
        assert len(batches) == len(sizes) // 4                      # drop_last
        seen[mode] = batches
    for mode in ("gpu_preprocess", "gpu_decode"):
        for (ia, la, xa), (ib, lb, xb) in zip(seen["cpu"], seen[mode]):
            assert torch.equal(xa, xb) and torch.equal(la, lb)
            assert torch.equal(ia, ib), mode
