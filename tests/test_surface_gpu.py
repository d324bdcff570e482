"""GPU: the drop-in surface end to end -- `models.arch.coop.LGHWithFixedPrompt` against the reference golden vectors,
`utils.hashing` against the oracle, and `main_v2.py --config-name val.yaml ...` on a synthetic run directory."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_fixture

pytestmark = pytest.mark.gpu


def _rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().pow(2).mean().sqrt().clamp_min(1e-12))


def test_model_class_reproduces_reference_outputs():
    from test_surface_cpu import _model_like_fixture
    sd, z = load_fixture("encode_hd64")
    model = _model_like_fixture(z, sd)
    model.load_state_dict(sd)
    model = model.to("cuda").eval()
    x = torch.from_numpy(z["in/images"]).cuda()
    with torch.no_grad():
        feats, out = model(x)
    assert set(out) == {"logits_cont", "logits_bin", "codes", "image_hidden_states", "hash_features", "attn_cache",
                        "logits_concept"}
    for key, got in (("codes", out["codes"]), ("logits_cont", out["logits_cont"]), ("logits_bin", out["logits_bin"]),
                     ("logits_concept", out["logits_concept"]), ("hash_features", out["hash_features"]),
                     ("image_features", feats)):
        ref = torch.from_numpy(z["out/" + key])
        assert got.shape == ref.shape and _rel_err(got.cpu(), ref) < 2e-2, key
    assert torch.allclose(model.get_center().cpu(), torch.from_numpy(
        np.maximum(z["sd/center"] @ z["sd/text_projection.0.weight"].T + z["sd/text_projection.0.bias"], 0)
        @ z["sd/text_projection.2.weight"].T + z["sd/text_projection.2.bias"]), atol=1e-4)
    # the reference's `image_hidden_states` contract, on request: embeddings + one state per layer, last one's concept rows = hash_features
    model.return_hidden_states = True
    with torch.no_grad():
        hs = model(x)[1]["image_hidden_states"]
    model.return_hidden_states = False
    assert len(hs) == 3 and hs[0].shape == (x.shape[0], 21, 128)
    for i, key in ((0, "h0"), (1, "h1"), (2, "h_last")):
        assert _rel_err(hs[i].cpu(), torch.from_numpy(z["out/" + key])) < 2e-2, key
    assert _rel_err(hs[-1][:, -4:, :].cpu(), out["hash_features"].cpu()) < 1e-6
    # parameters changed in place -> the engine is rebuilt, outputs change
    with torch.no_grad():
        model.hash_bn.bias.add_(1.0)
        _, out2 = model(x)
    assert torch.allclose(out2["codes"], out["codes"] + 1.0, atol=1e-5)


def test_utils_hashing_against_oracle():
    from oracle import hamming_oracle as ho
    from utils import hashing
    rng = np.random.default_rng(3)
    C, nbit = 11, 64
    centres = rng.standard_normal((C, nbit)).astype(np.float32)
    ql, gl = rng.integers(0, C, 150), rng.integers(0, C, 1200)
    qc = torch.from_numpy(centres[ql] + 0.9 * rng.standard_normal((150, nbit)).astype(np.float32))
    gc = torch.from_numpy(centres[gl] + 0.9 * rng.standard_normal((1200, nbit)).astype(np.float32))
    qoh, goh = torch.eye(C)[ql], torch.eye(C)[gl]                       # one-hot CPU tensors, as the evaluator passes them
    q, g = ho.pack(qc.numpy()), ho.pack(gc.numpy())
    mAP, recalls, precisions = hashing.calculate_mAP(gc, goh, qc, qoh, -1, threshold=0, dist_metric="hamming", PRs=[1, 5, 10])
    ref = ho.mean_ap(q, g, ql, gl, R=-1, ks=(1, 5, 10))
    assert abs(mAP - ref["mAP"]) < 1e-12 and np.allclose(recalls, ref["recalls"]) and np.allclose(precisions, ref["precisions"])
    mAPs, _, _ = hashing.calculate_mAP(gc, goh, qc, qoh, [50, -1], PRs=[1])
    assert abs(mAPs[0] - ho.mean_ap(q, g, ql, gl, R=50)["mAP"]) < 1e-12 and abs(mAPs[1] - ref["mAP"]) < 1e-12
    m2, _, p2 = hashing.calculate_mAP(qc, qoh, qc, qoh, -1, PRs=[1, 5], remove_first_retrieved=True)   # test-as-database
    ref2 = ho.mean_ap(q, q, ql, ql, R=-1, ks=(1, 5), remove_first=True)
    assert abs(m2 - ref2["mAP"]) < 1e-12 and np.allclose(p2, ref2["precisions"])
    rec, prec, Rs = hashing.calculate_pr_curve(gc, goh, qc, qoh, Rs=[1, 10, 100, 1200])
    for R, r_, p_ in zip(Rs, rec, prec):
        o = ho.mean_ap(q, g, ql, gl, R=R, ks=())
        assert abs(p_ - float((o["nrel"] / R).mean())) < 1e-12
        assert abs(r_ - float((o["nrel"] / np.maximum(o["total"], 1)).mean())) < 1e-12
    hd = hashing.get_hamm_dist(qc, torch.from_numpy(centres), normalize=True).cpu().numpy()
    sa, sb = np.where(qc.numpy() > 0, 1.0, -1.0), np.where(centres > 0, 1.0, -1.0)
    assert np.allclose(hd, 0.5 * (nbit - sa @ sb.T) / nbit)              # == get_hd, trainers/orthohash.py:263-264


def test_main_v2_validation_end_to_end(tmp_path):
    """make a run dir with a seeded checkpoint -> `main_v2.py --config-name val.yaml` -> history.json == oracle on the
    saved codes; then the option branches (sub_code_eval slice, zero_mean_eval, test_as_database, P/R curve)."""
    from oracle import hamming_oracle as ho
    logdir = str(tmp_path / "run")
    env = dict(os.environ, PYTHONPATH=ROOT)
    common = ["dataset=synthetic_cub200", "dataset.limit=96", "data_dir=" + str(tmp_path)]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_logdir.py"), logdir,
                    "model.backbone.name=synthetic/clip-vit-small-patch16", "model.nbit=64"] + common, check=True, env=env,
                   cwd=str(tmp_path))
    assert os.path.exists(os.path.join(logdir, "config.yaml")) and os.path.exists(os.path.join(logdir, "models", "best.pth"))

    def run(extra, name):
        ev = str(tmp_path / name)
        subprocess.run([sys.executable, os.path.join(ROOT, "main_v2.py"), "--config-name", "val.yaml", "logdir=" + logdir,
                        "batch_size=32", "save_code=True", "eval_logdir=" + ev] + common + extra, check=True, env=env,
                       cwd=str(tmp_path))
        return json.load(open(os.path.join(ev, "history.json"))), torch.load(os.path.join(ev, "outputs.pth"))

    hist, outs = run([], "ev0")
    db, te = outs["db"], outs["test"]
    assert db["codes"].shape == (96, 64) and te["codes"].shape == (96, 64) and te["labels"].shape == (96, 200)
    q, g = ho.pack(te["codes"].numpy()), ho.pack(db["codes"].numpy())
    ql, gl = te["labels"].argmax(1).numpy(), db["labels"].argmax(1).numpy()
    ref = ho.mean_ap(q, g, ql, gl, R=-1, ks=(1, 5, 10))
    assert abs(hist["mAP"] - ref["mAP"]) < 1e-12 and np.allclose(hist["precisions"], ref["precisions"])
    assert {"test_loss", "test_acc_cont", "test_acc_bin", "test_acc_concept", "test_quan", "db_loss"} <= set(hist)

    h1, _ = run(["sub_code_eval=True", "sub_code_eval_setting.start_bit=16", "sub_code_eval_setting.end_bit=32",
                 "sub_code_eval_setting.rand_bits=1", "zero_mean_eval=True"], "ev1")     # concept 1's sub-code, zero-mean
    dbc, tec = db["codes"][:, 16:32], te["codes"][:, 16:32]
    mean = dbc.mean(0, keepdim=True)
    ref1 = ho.mean_ap(ho.pack((tec - mean).numpy()), ho.pack((dbc - mean).numpy()), ql, gl)
    assert abs(h1["mAP"] - ref1["mAP"]) < 1e-12

    h2, _ = run(["test_as_database=True", "R=[10,-1]"], "ev2")
    assert abs(h2["mAP"][0] - ho.mean_ap(q, q, ql, ql, R=10, remove_first=True)["mAP"]) < 1e-12
    assert abs(h2["mAP"][1] - ho.mean_ap(q, q, ql, ql, R=-1, remove_first=True)["mAP"]) < 1e-12

    h3, _ = run(["compute_mAP=False"], "ev3")
    has_rel = float(np.mean([(gl == c).any() for c in ql]))          # recall@G is 1 for queries with >= 1 relevant row, else 0
    assert h3["Rs"][-1] == 96 and len(h3["recalls"]) == len(h3["Rs"]) and abs(h3["recalls"][-1] - has_rel) < 1e-12
    assert abs(h3["precisions"][0] - ref["precisions"][0]) < 1e-12   # P@1 of the curve == P@1 of the mAP run


def test_main_v2_two_ranks_on_one_gpu_match_the_single_process_run(tmp_path):
    """The multi-rank evaluation path on hardware (2 ranks sharing the one GPU, collectives over gloo -- a rehearsal of the
    one-process-per-GPU layout, not a scaling run): every rank encodes its contiguous block of each split, outputs are gathered
    in dataset order, the gallery is row-sharded for retrieval (histogram all-gather -> global prefix -> integer all-reduce),
    `eval_logdir` is decided on rank 0 and broadcast, only rank 0 writes files.  history.json and the saved codes equal the
    single-process run's bit for bit; 5 test images with batch size 4 also leave rank 1 with a single short batch."""
    import socket
    logdir = str(tmp_path / "run")
    env = dict(os.environ, PYTHONPATH=ROOT)
    common = ["dataset=synthetic_cub200", "dataset.limit=37", "data_dir=" + str(tmp_path)]
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_logdir.py"), logdir,
                    "model.backbone.name=synthetic/clip-vit-small-patch16", "model.nbit=64"] + common, check=True, env=env,
                   cwd=str(tmp_path))
    args = ["--config-name", "val.yaml", "logdir=" + logdir, "batch_size=16", "save_code=True", "R=[5,-1]"] + common
    ev1 = str(tmp_path / "ev_single")
    subprocess.run([sys.executable, os.path.join(ROOT, "main_v2.py")] + args + ["eval_logdir=" + ev1], check=True, env=env,
                   cwd=str(tmp_path))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ev2 = str(tmp_path / "ev_two_ranks")
    procs = []
    for r in range(2):
        e = dict(env, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 CH_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "main_v2.py")] + args + ["eval_logdir=" + ev2], env=e,
                                      cwd=str(tmp_path)))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    h1, h2 = json.load(open(os.path.join(ev1, "history.json"))), json.load(open(os.path.join(ev2, "history.json")))
    assert h1["mAP"] == h2["mAP"] and h1["precisions"] == h2["precisions"] and h1["recalls"] == h2["recalls"]
    o1, o2 = torch.load(os.path.join(ev1, "outputs.pth")), torch.load(os.path.join(ev2, "outputs.pth"))
    for split in ("test", "db"):
        assert torch.equal(o1[split]["codes"], o2[split]["codes"]) and torch.equal(o1[split]["labels"], o2[split]["labels"])
    for k in h1:
        if k.startswith(("test_", "db_")):
            assert abs(h1[k] - h2[k]) < 1e-6, k          # sample-weighted meters, reduced over ranks
    assert sorted(os.listdir(ev2)) == sorted(os.listdir(ev1))      # one set of files, written once
    # the code post-processing options on sharded outputs: a bit-range sub-code and the database-mean shift (the mean is taken
    # with the single-process arithmetic on rank 0 and broadcast) -- history.json again equal bit for bit
    extra = ["zero_mean_eval=True", "sub_code_eval=True", "sub_code_eval_setting.rand_bits=1", "sub_code_eval_setting.start_bit=8",
             "sub_code_eval_setting.end_bit=56", "save_code=False"]
    ev3, ev4 = str(tmp_path / "ev_single_pp"), str(tmp_path / "ev_two_ranks_pp")
    subprocess.run([sys.executable, os.path.join(ROOT, "main_v2.py")] + args + extra + ["eval_logdir=" + ev3], check=True, env=env,
                   cwd=str(tmp_path))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        e = dict(env, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 CH_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "main_v2.py")] + args + extra + ["eval_logdir=" + ev4], env=e,
                                      cwd=str(tmp_path)))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    h3, h4 = json.load(open(os.path.join(ev3, "history.json"))), json.load(open(os.path.join(ev4, "history.json")))
    assert h3["mAP"] == h4["mAP"] and h3["precisions"] == h4["precisions"] and h3["recalls"] == h4["recalls"]
    assert h3["mAP"] != h1["mAP"]                                  # the options did change the codes that were scored
    assert not os.path.exists(os.path.join(ev4, "outputs.pth"))


@pytest.mark.gpu
def test_module_to_device_moves_a_module_in_one_copy_per_dtype():
    """`utils.misc.module_to_device` (BaseTrainer.to_device): every parameter / buffer lands on the GPU with the values, dtypes, shapes
    and flags `module.to` gives, as views of one buffer per dtype; `_apply` overrides still see the move."""
    import copy
    from utils.misc import module_to_device

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.BatchNorm1d(19), torch.nn.Linear(19, 3, bias=False))
            self.frozen = torch.nn.Parameter(torch.randn(5, 7, 3), requires_grad=False)
            self.register_buffer("ids", torch.arange(11))
            self.register_buffer("strided", torch.randn(6, 4).t())       # not contiguous: moved by module.to itself
            self.tied = self.body[0].weight                                # one Parameter under two names
            self.applied = 0

        def _apply(self, fn, *a, **k):
            self.applied += 1
            return super()._apply(fn, *a, **k)

    torch.manual_seed(0)
    net = Net()
    ref = copy.deepcopy(net).to("cuda:0")
    out = module_to_device(net, "cuda:0")
    assert out is net and net.applied == 1
    got, want = dict(net.state_dict()), dict(ref.state_dict())
    assert got.keys() == want.keys()
    for k in want:
        assert got[k].device == want[k].device and got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert torch.equal(got[k], want[k]), k
    assert net.tied is net.body[0].weight and net.body[0].weight.requires_grad and not net.frozen.requires_grad
    f32 = [p for p in net.parameters()] + [net.body[1].running_mean, net.body[1].running_var]
    base = f32[0].untyped_storage().data_ptr()
    assert all(t.untyped_storage().data_ptr() == base for t in f32)       # one device buffer for the fp32 tensors
    assert all(t.data_ptr() % 256 == 0 for t in f32)
    x = torch.randn(8, 37, device="cuda:0")
    assert torch.equal(net.body(x), ref.body(x))
    net.body(x).sum().backward()                                           # the views are ordinary leaf parameters
    assert net.body[0].weight.grad is not None and net.frozen.grad is None
