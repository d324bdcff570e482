"""CPU: the drop-in Python surface (same dotted names as the reference) -- config composition, instantiate, the model's
state_dict key layout against the key list captured from the reference's own model, checkpoint tolerance, datasets."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_fixture
from concepthash_amd import config as cfglib

CONFIGS = os.path.join(ROOT, "configs")


def _model_like_fixture(z, sd, image_size=64):
    from models.arch.coop import LGHWithFixedPrompt
    from models.backbone.clip import CLIP
    D = sd["backbone.vision_model.pre_layrnorm.weight"].shape[0]
    dims = dict(hidden_size=D, num_hidden_layers=2, num_attention_heads=int(z["meta/heads"]),
                intermediate_size=sd["backbone.vision_model.encoder.layers.0.mlp.fc1.weight"].shape[0],
                patch_size=sd["backbone.vision_model.embeddings.patch_embedding.weight"].shape[-1], image_size=image_size,
                projection_dim=sd["hash_queries"].shape[2], hidden_act=str(z["meta/act"]))
    nbit = sd["hash_fc.weight"].shape[0] * 4
    C, cd = sd["center"].shape
    upt = cfglib.DictConfig(multi=True, num_heads=8, dropout=0.1, ensemble_method="concat", single_hash_fc=True, hash_pe=True)
    tp = torch.nn.Sequential(torch.nn.Linear(cd, cd), torch.nn.ReLU(), torch.nn.Linear(cd, nbit))
    return LGHWithFixedPrompt(CLIP(dims, allow_random_init=True), nbit, C, 4, add_bn=True, upt_config=upt,
                              fixed_center=torch.zeros(C, cd), text_projection=tp, has_adapter=True,
                              adapter_bottleneck_dim=sd["backbone.vision_model.encoder.layers.0.adapt_mlp_1.down_proj.weight"].shape[0],
                              concept_reg=True)


def test_head_dim_guard():
    from models.backbone.clip import CLIP
    with pytest.raises(ValueError, match="head_dim"):
        CLIP(dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128, patch_size=16,
                  image_size=64, projection_dim=32), allow_random_init=True)
    with pytest.raises(FileNotFoundError):
        CLIP("openai/clip-vit-base-patch16")                 # known id, but no weights and no allow_random_init
    with pytest.raises(FileNotFoundError):
        CLIP("someone/unknown-model", allow_random_init=True)


def test_state_dict_keys_equal_the_reference_models():
    """tests/golden/encode_hd64.npz carries the full key list of the reference LGHWithFixedPrompt.state_dict()."""
    sd, z = load_fixture("encode_hd64")
    model = _model_like_fixture(z, sd)
    ours = model.state_dict()
    ref_keys = set(str(k) for k in z["meta/all_state_dict_keys"])
    assert set(ours.keys()) == ref_keys
    for k, v in sd.items():
        assert tuple(ours[k].shape) == tuple(v.shape), k
    # a reference-layout checkpoint (with or without the alias entries) loads strictly
    full = {k: (sd[k] if k in sd else ours[k]) for k in ours}
    model.load_state_dict(full)
    model.load_state_dict(sd)                                # aliases + text leftovers absent: tolerated
    assert torch.equal(model.state_dict()["adapter_params.adapter_1_adapt_mlp_2_up_proj_weight"],
                       sd["backbone.vision_model.encoder.layers.1.adapt_mlp_2.up_proj.weight"])   # alias shares storage
    bad = dict(sd)
    del bad["hash_fc.weight"]
    with pytest.raises(RuntimeError, match="hash_fc.weight"):
        model.load_state_dict(bad)
    with pytest.raises(RuntimeError):                        # CPU tensor: no silent fallback
        model(torch.zeros(1, 3, 64, 64))
    model.train()
    with pytest.raises(RuntimeError):                        # training mode as well: the HIP trainer or nothing
        model(torch.zeros(1, 3, 64, 64))


def test_compose_val_and_train_configs(tmp_path):
    cfg = cfglib.compose(CONFIGS, "val.yaml", ["logdir=/x/run", "dataset=synthetic_cub200", "R=[100,-1]", "batch_size=16"],
                         cwd=str(tmp_path))
    assert cfg.R == [100, -1] and cfg.PRs == [1, 5, 10] and cfg.batch_size == 16 and cfg.dataset_name == "synthetic_cub200"
    assert cfg.eval_logdir.startswith("/x/run/evaluations/42_") and cfg.dataset.nclass == 200
    assert cfg.work_dir == str(tmp_path) and cfg.dataset.test_dataset.root == f"{tmp_path}/data/cub200_2011"
    tr = cfglib.compose(CONFIGS, "train.yaml", ["dataset=nabirds", "model.nbit=128", "optim_lr=1"], cwd=str(tmp_path))
    assert tr.model.nclass == 555 and tr.model.text_projection._args_[2].out_features == 128
    assert tr.trainer._target_ == "trainers.coop.COOPTrainer" and tr.criterion.ncontext == 4
    assert tr.batch_size == 32 and tr.dataset.norm == 3                      # @package _global_ overrides of the model file
    assert tr.method_name == "concept_hash_final_v1_nosa_apt"
    with pytest.raises(ValueError):
        cfglib.compose(CONFIGS, "val.yaml", ["oops"])
    ev = cfglib.compose(CONFIGS, "val.yaml", ["logdir=/x", 'sub_code_eval_setting.end_bit=${eval:"int(64 - 1)"}'])
    assert ev.sub_code_eval_setting.end_bit == 63


def test_instantiate_and_torchvision_mapping():
    node = cfglib._wrap({"_target_": "torch.nn.Sequential", "_args_": [
        {"_target_": "torch.nn.Linear", "in_features": 4, "out_features": 3}, {"_target_": "torch.nn.ReLU"}]})
    m = cfglib.instantiate(node)
    assert isinstance(m[0], torch.nn.Linear) and m[0].out_features == 3
    t = cfglib.instantiate(cfglib._wrap({"_target_": "torchvision.transforms.CenterCrop", "size": 8}))
    from PIL import Image
    assert t(Image.new("RGB", (20, 10))).size == (8, 8)
    from trainers.coop import COOPTrainer
    tr = cfglib.instantiate(cfglib._wrap({"_target_": "trainers.coop.COOPTrainer"}), cfglib.DictConfig(device="cpu"))
    assert isinstance(tr, COOPTrainer)
    with pytest.raises(AssertionError):                      # nothing loaded yet (reference trainers/base.py:346)
        tr.train_one_epoch()


def test_datasets_and_transforms(tmp_path):
    from PIL import Image
    from utils.datasets import HashingDataset, OneHot, SyntheticHashingDataset
    from utils import transforms as T
    root = tmp_path / "d"
    (root / "img").mkdir(parents=True)
    rng = np.random.default_rng(0)
    lines = []
    for i in range(5):
        Image.fromarray(rng.integers(0, 255, (40 + i, 60, 3), dtype=np.uint8)).save(root / "img" / f"{i}.jpg")
        lines.append(f"img/{i}.jpg {i % 3}")
    (root / "test.txt").write_text("\n".join(lines) + "\n")
    ds = HashingDataset(str(root), "test.txt", transform=[T.Resize(32, T.interpolation("bicubic")), T.CenterCrop(24),
                                                         T.ToTensor(), T.normalize_transform(3)],
                        target_transform=OneHot(3))
    img, target, idx = ds[4]
    assert img.shape == (3, 24, 24) and target.tolist() == [0.0, 1.0, 0.0] and idx == 4 and len(ds) == 5
    # gpu_preprocess: the CPU side only decodes; batches are RawImageBatch (bytes back to back + sizes)
    import engine
    raw = HashingDataset(str(root), "test.txt", transform=[T.Resize(32, T.interpolation("bicubic"))], target_transform=OneHot(3),
                         gpu_preprocess=True)
    im, tg, ix = raw[4]
    assert im.dtype == torch.uint8 and im.shape == (44, 60, 3) and tg.tolist() == [0.0, 1.0, 0.0]
    batch = next(iter(engine.dataloader(raw, bs=3, workers=0)))
    rb, targets, idxs = batch
    assert rb.size(0) == 3 and rb.sizes == [(40, 60), (41, 60), (42, 60)] and rb.pixels.numel() == (40 + 41 + 42) * 60 * 3
    assert targets.shape == (3, 3) and idxs.tolist() == [0, 1, 2]
    assert torch.equal(rb.pixels[:40 * 60 * 3].reshape(40, 60, 3), raw[0][0])
    syn = SyntheticHashingDataset(7, size=50, image_size=32, seed=3)
    a, b = syn[10], syn[10]
    assert torch.equal(a[0], b[0]) and a[1].sum() == 1 and len(syn) == 50          # deterministic per index
    real = SyntheticHashingDataset(3, root=str(root), filename="test.txt", image_size=16)
    assert [int(real[i][1].argmax()) for i in range(5)] == [0, 1, 2, 0, 1]          # label vector from the list file


def test_loss_meters_match_a_direct_formula():
    from models.loss.coop import LGHLoss
    torch.manual_seed(0)
    out = {"codes": torch.randn(6, 16), "logits_cont": torch.rand(6, 5) * 2 - 1, "logits_bin": torch.rand(6, 5) * 2 - 1,
           "logits_concept": torch.rand(4, 6, 5) * 2 - 1}
    y = torch.randint(0, 5, (6,))
    crit = LGHLoss(scale=8, margin=0.2, loss_scales={"bin_logits": 1, "cont_logits": 1, "concept_logits": 1}, ncontext=4)
    total = crit(out, y)
    oh = torch.nn.functional.one_hot(y, 5).float()
    ce = lambda l, m=0.2: torch.nn.functional.cross_entropy(8 * (l - m * oh), y)
    # (Q, B, C) logits with index labels: the reference's scatter gives the margin to concept 0 only (models/loss/coop.py:55-57)
    concept = torch.stack([ce(out["logits_concept"][q], 0.2 if q == 0 else 0.0) for q in range(4)]).mean()
    assert torch.allclose(total, ce(out["logits_cont"]) + ce(out["logits_bin"]) + concept, atol=1e-6)
    assert set(crit.losses) == {"quan", "concept", "cont", "bin"}
    with pytest.raises(NotImplementedError):
        LGHLoss(loss_scales={"attn_div_loss": 1}, nregs=2)           # register tokens: the model has none
    with pytest.raises(RuntimeError, match="concept_attention_layers"):
        LGHLoss(loss_scales={"attn_div_loss": 1}, avg_attn=True)(out, y)   # avg_attn reads every layer's concept rows
    with pytest.raises(RuntimeError, match="concept_attention"):
        LGHLoss(loss_scales={"attn_div_loss": 1})(out, y)            # the term is built, but the model must hand the rows over


def test_loss_equals_the_reference_loss_on_its_own_outputs():
    """tests/golden/train_tiny.npz: the reference's LGHLoss (shipped settings) evaluated by the reference on its own model's
    train-mode outputs; the product criterion on the same logits must give the same three terms -- and is differentiable."""
    from models.loss.coop import LGHLoss
    _, z = load_fixture("train_tiny")
    out = {k: torch.from_numpy(z["out/" + k]).clone().requires_grad_(k != "codes")
           for k in ("codes", "logits_cont", "logits_bin", "logits_concept")}
    crit = LGHLoss(margin=0.2, scale=8, loss_scales=dict(logits=0, hash_logits=0, bin_logits=1, cont_logits=1, l2=0, attn_div_loss=0,
                                                         concept_logits=1), avg_before_softmax=False, lmbd=0.5, div_method=1, ncontext=4)
    total = crit(out, torch.from_numpy(z["in/labels"]))
    assert abs(float(total.detach()) - float(z["out/loss"])) < 1e-5
    for k in ("concept", "cont", "bin", "quan"):
        assert abs(float(crit.losses[k]) - float(z["out/loss_" + k])) < 1e-5, k
    total.backward()
    assert all(out[k].grad is not None and float(out[k].grad.abs().sum()) > 0 for k in ("logits_cont", "logits_bin", "logits_concept"))


def test_schedulers_and_training_config_compose(tmp_path):
    from utils.lr_scheduler import cosine_decay_linear_warmup, no_decay
    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=1.0)
    sch = cosine_decay_linear_warmup(opt, epochs=20, warmup_epochs=4)
    lrs = []
    for _ in range(20):
        lrs.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    assert lrs[:4] == [0.25, 0.5, 0.75, 1.0] and abs(lrs[4] - 1.0) < 1e-12 and abs(lrs[12] - 0.5) < 1e-12 and lrs[-1] < 0.02
    assert all(a >= b for a, b in zip(lrs[4:], lrs[5:]))
    assert no_decay(torch.optim.SGD(p, lr=0.3)).get_last_lr() == [0.3]
    cfg = cfglib.compose(CONFIGS, "train.yaml", ["dataset=synthetic_cub200", "optim=sgd", "model.nbit=64", "epochs=3", "eval_interval=0",
                                                 "scheduler=no_decay"], cwd=str(tmp_path))
    assert cfg.exp == "hashing" and cfg.optim["_target_"] == "torch.optim.sgd.SGD" and cfg.optim.momentum == 0.9
    assert cfg.optim.lr == 0.001                                  # the model config overrides optim.lr (reference ...apt.yaml:76-77)
    assert cfg.backbone_lr_scale == 0 and cfg.batch_size == 32 and cfg.dataset.train_dataset["_target_"].endswith("SyntheticHashingDataset")
    assert cfg.scheduler["_target_"] == "utils.lr_scheduler.no_decay"
    opt = cfglib.instantiate(cfg.optim, [{"params": p}])
    assert isinstance(opt, torch.optim.SGD) and opt.defaults["weight_decay"] == 0.0005


def test_utils_hashing_rejects_unbuilt_options_before_touching_the_gpu():
    from utils import hashing
    c = torch.randn(4, 16)
    l = torch.eye(4)
    with pytest.raises(NotImplementedError):
        hashing.calculate_mAP(c, l, c, l, -1, dist_metric="euclidean")
    with pytest.raises(NotImplementedError):
        hashing.calculate_mAP(c, l, c, l, -1, threshold=0.5)
    assert hashing.pr_curve_points(5994)[-1] == 5994 and hashing.pr_curve_points(5994)[:4] == [1, 2, 5, 10]


def test_clip_backbone_from_local_hf_directory(tmp_path):
    """`model.backbone.name=<local dir>` (HF layout: config.json + model.safetensors) replaces the reference's
    from_pretrained(<hub name>) (models/backbone/clip.py:112-118), which needs the network."""
    import json
    from safetensors.torch import save_file
    from models.backbone.clip import CLIP, CLIPModelShell
    dims = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, patch_size=16,
                image_size=64, projection_dim=64, hidden_act="quick_gelu")
    torch.manual_seed(5)
    src = CLIPModelShell(dims, text_dim=32)
    sd = {k: v.detach().clone() + 0.01 for k, v in src.state_dict().items()}
    sd["text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 32)      # extra text-tower tensors are ignored
    d = tmp_path / "clip_local"
    d.mkdir()
    save_file(sd, str(d / "model.safetensors"))
    json.dump({"projection_dim": 64, "vision_config": dict(dims), "text_config": {"hidden_size": 32}}, open(d / "config.json", "w"))
    bb = CLIP(str(d))
    got = bb.model.state_dict()
    for k in ("vision_model.embeddings.patch_embedding.weight", "vision_model.encoder.layers.1.mlp.fc2.bias",
              "vision_model.pre_layrnorm.weight", "visual_projection.weight"):
        assert torch.equal(got[k], sd[k]), k
    assert bb.features_size == 128 and bb.model.vision_model.config.projection_dim == 64
    (d / "model.safetensors").unlink()
    save_file({k: v for k, v in sd.items() if "fc2" not in k}, str(d / "model.safetensors"))
    with pytest.raises(KeyError, match="missing"):
        CLIP(str(d))
