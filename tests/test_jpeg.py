"""JPEG decode split (reference: the loader workers' `PIL.Image.open(path).convert("RGB")`, engine.py:41-54).

CPU: the HOST half of the product (`ch_jpeg_plan` / `ch_jpeg_entropy_decode`: marker parsing + Huffman decode, plain C++ -- it runs
without a GPU) feeds oracle/jpeg_oracle.py (numpy restatement of libjpeg-turbo's islow IDCT, fancy upsampling and YCbCr -> RGB); the
result must equal PIL's bytes -- this pins both the entropy decoder and the restatement against Pillow itself.
GPU: `GpuJpegDecoder` (host entropy decode + `ch_jpeg_reconstruct`) equals PIL bit for bit on the same files, mixed batches with
unsupported files (CMYK, tiny -> PIL on the host, counted) included; progressive files (SOF2) are decoded by the same split, and `dataset.gpu_decode: true` gives the codes of the
reference-style CPU loader through `COOPTrainer.inference_one_epoch`."""
import ctypes
import io

import numpy as np
import pytest
import torch
from PIL import Image


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        for _ in range(5):
            fx, fy = rng.uniform(0.005, 0.15, 2)
            img[:, :, c] += rng.uniform(10, 60) * np.sin(fx * xx + fy * yy + rng.uniform(0, 6.28))
    img += 128 + rng.normal(0, 10, img.shape)
    img[h // 4:h // 2, w // 3:w // 2] = rng.uniform(0, 255, 3)      # a flat, possibly saturated patch with hard edges
    img[:8, :8] = 255
    img[-5:, -7:] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


def _jpeg(img, mode="RGB", **kw):
    im = Image.fromarray(img)
    if mode == "L":
        im = im.convert("L")
    bio = io.BytesIO()
    im.save(bio, "JPEG", **kw)
    return bio.getvalue()


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


# (h, w), subsampling (0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0), quality, mode, extra save options
CASES = [((375, 500), 2, 75, "RGB", {}), ((500, 375), 2, 90, "RGB", {}), ((333, 500), 1, 85, "RGB", {}), ((241, 255), 0, 95, "RGB", {}),
         ((17, 16), 2, 75, "RGB", {}), ((16, 33), 1, 50, "RGB", {}), ((64, 64), 0, 30, "RGB", {}), ((480, 640), 2, 20, "RGB", {}),
         ((375, 500), 2, 75, "L", {}), ((129, 67), 0, 90, "L", {}), ((375, 500), 2, 75, "RGB", dict(optimize=True)),
         ((375, 500), 2, 75, "RGB", dict(restart_marker_blocks=5)), ((300, 401), 1, 85, "RGB", dict(restart_marker_rows=1)),
         ((99, 1001), 2, 60, "RGB", {}), ((1001, 99), 2, 60, "RGB", {}), ((257, 259), 2, 100, "RGB", {}),
         # progressive (SOF2): spectral selection + successive approximation, every sampling, grey, optimised tables, restart intervals
         ((375, 500), 2, 75, "RGB", dict(progressive=True)), ((333, 500), 1, 85, "RGB", dict(progressive=True)),
         ((241, 255), 0, 95, "RGB", dict(progressive=True)), ((129, 67), 0, 90, "L", dict(progressive=True)),
         ((17, 16), 2, 30, "RGB", dict(progressive=True)), ((480, 640), 2, 100, "RGB", dict(progressive=True, optimize=True)),
         ((300, 401), 2, 80, "RGB", dict(progressive=True, restart_marker_rows=1)),
         ((203, 310), 1, 60, "RGB", dict(progressive=True, restart_marker_blocks=7))]


def _files():
    out = []
    for i, ((h, w), sub, q, mode, kw) in enumerate(CASES):
        kw = dict(kw, quality=q)
        if mode == "RGB":
            kw["subsampling"] = sub
        out.append(_jpeg(_image(h, w, i), mode, **kw))
    return out


def _host_decode(files, threads=2):
    from concepthash_amd import _lib
    from concepthash_amd.jpeg import DESC_DTYPE, GpuJpegDecoder
    lib = _lib.load()
    n = len(files)
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (ctypes.c_int64 * n)(*[len(f) for f in files])
    desc = np.zeros(n, dtype=DESC_DTYPE)
    tc, tp, tl = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.ch_jpeg_plan(ptrs, lens, n, desc.ctypes.data, ctypes.byref(tc), ctypes.byref(tp), ctypes.byref(tl)) == 0
    assert (tc.value, tp.value, tl.value) == GpuJpegDecoder.layout(desc.copy())     # the library's default layout == the wrapper's
    coef = np.full(max(tc.value, 1), 12345, np.int16)                                 # poisoned: every block must be fully written
    assert lib.ch_jpeg_entropy_decode(ptrs, lens, n, desc.ctypes.data, coef.ctypes.data, threads) == 0
    return desc, coef


def test_host_entropy_decoder_and_oracle_reconstruction_equal_pillow():
    from oracle import jpeg_oracle as jo
    files = _files()
    desc, coef = _host_decode(files)
    for i, f in enumerate(files):
        d = desc[i]
        assert d["status"] == 0, (CASES[i], int(d["status"]))
        ref = _pil(f)
        assert (d["height"], d["width"]) == ref.shape[:2]
        got = jo.reconstruct(coef[d["coef_offset"]:d["coef_offset"] + int(d["nblocks"]) * 64], d)
        assert np.array_equal(got, ref), (CASES[i], int((got != ref).sum()))
    # one thread and many threads write the same coefficients
    d1, c1 = _host_decode(files, threads=1)
    d8, c8 = _host_decode(files, threads=8)
    assert np.array_equal(c1, c8) and np.array_equal(d1, d8)


def test_files_outside_the_subset_are_flagged_not_guessed():
    img = _image(120, 160, 3)
    prog = _jpeg(img, quality=80, progressive=True)
    cmyk = io.BytesIO()
    Image.fromarray(img).convert("CMYK").save(cmyk, "JPEG")
    tiny = _jpeg(_image(12, 40, 1), quality=80)
    png = io.BytesIO()
    Image.fromarray(img).save(png, "PNG")
    good = _jpeg(img, quality=80)
    trunc = good[: len(good) // 2]
    desc, _ = _host_decode([prog, cmyk.getvalue(), tiny, png.getvalue(), good, good[:100], trunc])
    assert list(desc["status"][:5]) == [0, 5, 10, 1, 0]           # (progressive files are decoded since round 4)
    assert desc["status"][5] != 0                                 # cut inside the headers
    assert desc["status"][6] == 2                                 # cut inside the entropy-coded data: found by the decode, left to PIL (which raises)
    # a damaged frame header that announces a gigantic image: flagged from the header alone (more blocks than the file has bits / more
    # pixels than Pillow accepts), never allocated
    sof = good.index(b"\xff\xc0")
    huge = bytearray(good)
    huge[sof + 5:sof + 9] = bytes([0xFF, 0xF0, 0xFF, 0xF0])       # 65520 x 65520
    d3, _ = _host_decode([bytes(huge), good])
    assert d3["status"][0] in (2, 12) and d3["status"][1] == 0 and int(d3["nblocks"][0]) == 0
    prog_trunc = prog[: len(prog) * 2 // 3]
    d2, _ = _host_decode([prog_trunc, good])
    assert d2["status"][0] != 0 and d2["status"][1] == 0
    assert (desc["height"][0], desc["width"][0]) == (120, 160)    # the size of a file the caller decodes itself is still reported


def test_damaged_files_never_crash_the_host_decoder_or_overrun_its_buffer():
    """The host half parses bytes from disk: whatever is in a file -- flipped bytes, a cut, a stray marker, a damaged header -- the calls
    return (with some status) and never write past the coefficient buffer the plan sized.  400 mutations of five valid files (baseline,
    progressive, 4:4:4, 4:2:2 with restart intervals, progressive with restart rows)."""
    from concepthash_amd import _lib
    from concepthash_amd.jpeg import DESC_DTYPE
    lib = _lib.load()
    rng = np.random.default_rng(7)
    seeds = [_jpeg(_image(64, 80, 1), quality=80), _jpeg(_image(48, 64, 2), quality=60, progressive=True),
             _jpeg(_image(40, 56, 3), quality=90, subsampling=0), _jpeg(_image(33, 47, 4), quality=70, subsampling=1, restart_marker_blocks=3),
             _jpeg(_image(64, 64, 5), quality=50, progressive=True, restart_marker_rows=1)]
    seen = set()
    for it in range(80):
        batch = []
        for sd in seeds:
            b = bytearray(sd)
            mode = int(rng.integers(0, 4))
            if mode == 0:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif mode == 1:
                b = b[: int(rng.integers(2, len(b)))]
            elif mode == 2:
                i = int(rng.integers(2, len(b) - 4))
                b[i:i + 2] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
            else:
                b[int(rng.integers(0, min(len(b), 600)))] = int(rng.integers(0, 256))       # the header region
            batch.append(bytes(b))
        n = len(batch)
        bufs = [np.frombuffer(f, dtype=np.uint8) for f in batch]
        ptrs = (ctypes.c_void_p * n)(*[x.ctypes.data for x in bufs])
        lens = (ctypes.c_int64 * n)(*[len(f) for f in batch])
        desc = np.zeros(n, dtype=DESC_DTYPE)
        tc, tp, tl = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        assert lib.ch_jpeg_plan(ptrs, lens, n, desc.ctypes.data, ctypes.byref(tc), ctypes.byref(tp), ctypes.byref(tl)) == 0
        assert tc.value <= 64 * 8 * sum(len(f) for f in batch)            # a plan never asks for more blocks than the files have bits
        coef = np.zeros(tc.value + 64, np.int16)
        coef[tc.value:] = 777
        assert lib.ch_jpeg_entropy_decode(ptrs, lens, n, desc.ctypes.data, coef.ctypes.data, 2) == 0
        assert (coef[tc.value:] == 777).all()
        seen.update(int(v) for v in desc["status"])
    assert 0 in seen and len(seen) >= 4                                     # intact-enough files decode, the others are told apart


_SAN_SCRIPT = r"""
import ctypes, io, sys
import numpy as np
from PIL import Image
lib = ctypes.CDLL(sys.argv[1])
lib.ch_jpeg_plan.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.ch_jpeg_entropy_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
rng = np.random.default_rng(11)
def jpeg(h, w, **kw):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(127 + 100 * np.sin(xx / 9.0 + c) * np.cos(yy / 7.0)) for c in range(3)], -1) + rng.normal(0, 12, (h, w, 3))
    bio = io.BytesIO()
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(bio, "JPEG", **kw)
    return bio.getvalue()
seeds = [jpeg(64, 80, quality=80), jpeg(48, 64, quality=60, progressive=True), jpeg(40, 56, quality=90, subsampling=0),
         jpeg(33, 47, quality=70, subsampling=1, restart_marker_blocks=3), jpeg(64, 64, quality=50, progressive=True, restart_marker_rows=1),
         jpeg(200, 120, quality=95, progressive=True, optimize=True)]
DESC = 448
files = 0
for it in range(int(sys.argv[2])):
    batch = []
    for sd in seeds:
        b = bytearray(sd)
        mode = int(rng.integers(0, 5))
        if mode == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif mode == 1:
            b = b[: int(rng.integers(2, len(b)))]
        elif mode == 2:
            i = int(rng.integers(2, len(b) - 4)); b[i:i + 2] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
        elif mode == 3:
            b[int(rng.integers(0, min(len(b), 600)))] = int(rng.integers(0, 256))
        batch.append(bytes(b))                       # mode 4: the intact file
    n = len(batch)
    bufs = [np.frombuffer(f, dtype=np.uint8).copy() for f in batch]      # exact-size heap copies: an over-read is an ASan report
    ptrs = (ctypes.c_void_p * n)(*[x.ctypes.data for x in bufs])
    lens = (ctypes.c_int64 * n)(*[len(f) for f in batch])
    desc = np.zeros(n * DESC, dtype=np.uint8)
    tc = ctypes.c_int64()
    assert lib.ch_jpeg_plan(ptrs, lens, n, desc.ctypes.data, ctypes.byref(tc), None, None) == 0
    coef = np.zeros(max(tc.value, 1), np.int16)      # exact size: an overrun is an ASan report
    assert lib.ch_jpeg_entropy_decode(ptrs, lens, n, desc.ctypes.data, coef.ctypes.data, 2) == 0
    files += n
print("sanitizer run over", files, "files: clean")
"""


def test_host_decoder_under_address_and_ub_sanitizers(tmp_path):
    """The host half of the decode split is plain C++ (csrc/jpeg_host.cpp + errors.cpp): built here with g++ -fsanitize=address,undefined
    and driven with intact and damaged files -- flipped bytes, cuts, stray markers, damaged headers -- on exact-size heap buffers, in a
    subprocess with the sanitizer runtime preloaded.  Any out-of-bounds read or write, use after free or undefined shift fails the test."""
    import os
    import shutil
    import subprocess
    import sys
    gxx = shutil.which("g++")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip() if gxx else ""
    if not gxx or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("g++ / libasan not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "concepthash_amd", "csrc")
    so = str(tmp_path / "libjpeg_host_san.so")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-shared", "-fPIC", "-pthread", "-I", csrc, os.path.join(csrc, "jpeg_host.cpp"), os.path.join(csrc, "errors.cpp"), "-o", so],
                   check=True)
    script = tmp_path / "san_fuzz.py"
    script.write_text(_SAN_SCRIPT)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, str(script), so, "60"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_gpu_jpeg_decoder_equals_pillow_bit_for_bit():
    from concepthash_amd.jpeg import GpuJpegDecoder, decode_to_list
    dev = torch.device("cuda:0")
    files = _files()
    cmyk = io.BytesIO()
    Image.fromarray(_image(200, 300, 77)).convert("CMYK").save(cmyk, "JPEG", quality=85)
    files.insert(3, cmyk.getvalue())                                                # outside the subset: PIL on the host, same bytes
    dec = GpuJpegDecoder(device=dev, threads=4)
    for rep in range(3):                                                             # the pinned ring is reused
        outs = decode_to_list(dec, files)
        torch.cuda.synchronize()
        for f, o in zip(files, outs):
            ref = _pil(f)
            assert tuple(o.shape) == ref.shape
            assert np.array_equal(o.cpu().numpy(), ref)
    assert dec.stats["pil_fallback"] == 3 and dec.stats["gpu"] == 3 * (len(files) - 1)
    assert dec.stats["fallback_reasons"] == {"component count": 3}
    with pytest.raises(ValueError, match="strict"):
        GpuJpegDecoder(device=dev, strict=True).decode(files)
    pixels, sizes = dec.decode([])
    assert pixels.numel() == 0 and sizes == []
    # a larger batch of one size (the loader's common case), several host threads
    batch = [_jpeg(_image(375, 500, 100 + i), quality=90) for i in range(24)]
    outs = decode_to_list(GpuJpegDecoder(device=dev), batch)
    for f, o in zip(batch, outs):
        assert np.array_equal(o.cpu().numpy(), _pil(f))


@pytest.mark.gpu
def test_trainer_with_gpu_decode_gives_the_codes_of_the_cpu_loader(tmp_path):
    """`dataset.gpu_decode: true` end to end: list-file dataset of JPEG files of different sizes / samplings -> DataLoader workers only
    READ -> RawJpegBatch -> COOPTrainer: host entropy decode + ch_jpeg_reconstruct + ch_preprocess + ch_encode.  Codes are bit-equal to
    the run whose CPU workers decode with PIL and apply the torchvision-style transform chain (the reference's loader, engine.py:41-54,
    configs/dataset/cub200.yaml:31-47)."""
    from concepthash_amd import config as cfglib
    from concepthash_amd import synthetic as syn
    from trainers.coop import COOPTrainer
    from utils import transforms as T
    from utils.datasets import HashingDataset, OneHot
    root = tmp_path / "d"
    (root / "img").mkdir(parents=True)
    lines = []
    # (file 4 is smaller than 16 x 16 blocks allow on the GPU path -> PIL on the host; file 5 is progressive -> GPU)
    cases = [((375, 500), 2, 75), ((500, 375), 2, 90), ((333, 500), 1, 85), ((300, 300), 0, 95), ((12, 48), 2, 80), ((400, 731), 2, 70),
             ((257, 300), 2, 90)]
    for i, ((h, w), sub, q) in enumerate(cases):
        (root / "img" / f"{i}.jpg").write_bytes(_jpeg(_image(h, w, i), quality=q, subsampling=sub, progressive=(i == 5)))
        lines.append(f"img/{i}.jpg {i % 3}")
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
            return None, self.enc.encode(x, want=("codes", "logits_cont", "logits_bin"))

    class Crit(torch.nn.Module):
        losses = {}

        def forward(self, out, y):
            return out["codes"].sum() * 0

    chain = [T.Resize(256, T.interpolation("bicubic")), T.CenterCrop(224), T.ToTensor(), T.normalize_transform(3)]
    codes = {}
    for mode in (False, True, "coalesced"):
        # `eval_batch_min` (default 256) reads a GPU-decode evaluation split in larger batches than configured: 0 keeps the two batches of 4 + 3
        conf = cfglib.DictConfig(device="cuda", batch_size=4, model=cfglib.DictConfig(), eval_batch_min=0 if mode is True else 256,
                                 dataset=cfglib.DictConfig(multiclass=False, resize=256, crop=224, norm=3, gpu_preprocess=bool(mode), gpu_decode=bool(mode)))
        tr = COOPTrainer(conf)
        tr.dataset = {"train": [], "db": [], "test": HashingDataset(str(root), "test.txt", transform=chain, target_transform=OneHot(3),
                                                                     gpu_decode=bool(mode))}
        tr.load_dataloader()
        assert len(tr.dataloader["test"]) == (1 if mode == "coalesced" else 2)
        tr.model, tr.criterion = Model(), Crit()
        meters, out = tr.inference_one_epoch("test", True)
        codes[mode] = out["codes"]
        assert out["codes"].shape == (7, 32) and out["labels"].shape == (7, 3)
        if mode:
            assert tr._gpu_jpeg.stats["pil_fallback"] == 1 and tr._gpu_jpeg.stats["gpu"] == 6
    # the CPU loader hands fp32 tensors to the model, the GPU path bf16: compare through the same rounding
    assert torch.equal(codes[True], codes[False]) and torch.equal(codes["coalesced"], codes[False])


def test_packed_batch_calls_equal_the_per_file_calls():
    """`ch_jpeg_plan_packed` / `ch_jpeg_entropy_decode_packed` (files back to back in one buffer, as a `gpu_decode` loader worker hands
    them over) give the descriptors and coefficients of the per-file calls."""
    from concepthash_amd import _lib
    from concepthash_amd.jpeg import DESC_DTYPE
    lib = _lib.load()
    files = _files()[:6]
    d1, c1 = _host_decode(files)
    data = np.concatenate([np.frombuffer(f, dtype=np.uint8) for f in files])
    off = np.zeros(len(files) + 1, np.int64)
    off[1:] = np.cumsum([len(f) for f in files])
    d2 = np.zeros(len(files), dtype=DESC_DTYPE)
    tc, tp, tl = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.ch_jpeg_plan_packed(data.ctypes.data, off.ctypes.data, len(files), d2.ctypes.data, ctypes.byref(tc), ctypes.byref(tp), ctypes.byref(tl)) == 0
    c2 = np.full(tc.value, 777, np.int16)
    assert lib.ch_jpeg_entropy_decode_packed(data.ctypes.data, off.ctypes.data, len(files), d2.ctypes.data, c2.ctypes.data, 3) == 0
    assert np.array_equal(d1, d2) and np.array_equal(c1, c2)
    bad = off.copy()
    bad[2] = bad[1] - 1
    assert lib.ch_jpeg_plan_packed(data.ctypes.data, bad.ctypes.data, len(files), d2.ctypes.data, None, None, None) != 0
