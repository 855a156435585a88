"""a real distillation step (stock torch student / teacher, hooks, loss, backward, flat gradient
all-reduce incl. the selector temperatures, optimizer step).  CPU tests cover the plumbing with the oracle as the loss
(model layout, probing, mixing, sharding, two gloo ranks staying bit-identical); the GPU test runs the step on the HIP
library and checks the loss of the captured tensors against the oracle."""
import os
import socket
import sys
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _config(points=4, classes=10):
    return SimpleNamespace(training=SimpleNamespace(label_smoothing=0.1, learning_rate=1e-3, weight_decay=0.05),
                           basd=SimpleNamespace(num_extraction_points=points), model=SimpleNamespace(num_classes=classes))


class OracleBASD(nn.Module):
    """The oracle behind the reference constructor's signature (test-side stand-in for the loss module on CPU)."""

    def __init__(self, base_criterion, student_dim, teacher_dim, student_depth, num_student_tokens, *, config,
                 teacher_has_cls_token):
        super().__init__()
        from oracle import basd_oracle as O
        self.O = O
        self.base_criterion, self.has_cls, self.n_s = base_criterion, teacher_has_cls_token, num_student_tokens
        self.token_layers = O.extraction_layers(student_depth, config.num_extraction_points)
        st = O.SelectorState.create(len(self.token_layers), student_dim, teacher_dim)
        self.register_buffer("proj_s", st.proj_s)
        self.register_buffer("proj_t", st.proj_t)
        self.log_temperatures = nn.Parameter(st.log_temperatures.detach().clone())

    def forward(self, logits, targets, s_tokens, t_tokens, t_attns):
        st = self.O.SelectorState(self.proj_s, self.proj_t, self.log_temperatures)
        t_attns = {k: v.contiguous() for k, v in t_attns.items()}
        return self.O.basd_forward(st, self.base_criterion, self.token_layers, self.n_s, self.has_cls, logits, targets,
                                   {k: v.float() for k, v in s_tokens.items()},
                                   {k: v.float() for k, v in t_tokens.items()}, t_attns)[0]


def _structured_images(B, size, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(B, 3, 1, 1, generator=g) * 2.0
    return base + torch.randn(B, 3, size, size, generator=g)


def _toy_models(kind, dev="cpu"):
    from basd_amd import trainer as T
    from tools import stock_models as SM
    torch.manual_seed(3)
    student = SM.StockViT(img_size=32, patch_size=8, embed_dim=48, depth=6, num_heads=4, num_classes=10).to(dev)
    if kind == "cnn":
        teacher = SM.StockResNet(layers=(1, 1, 1, 1), bottleneck=False, width=16).to(dev)   # 128 channels, 1 x 1 map at 32^2
    else:
        teacher = SM.StockViT(img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=0).to(dev)
    return student, SM.make_teacher(teacher, 32)


def test_stock_models_expose_the_probed_layout():
    from basd_amd import trainer as T
    from tools import stock_models as SM
    student, teacher = _toy_models("cnn")
    info = SM.probe_model(student, 32)
    assert info["layer_paths"] == [f"blocks.{i}" for i in range(6)] and info["attn_subpath"] == "attn"
    assert info["has_cls_token"] and info["feature_format"] == "token" and info["num_tokens"] == 16
    assert info["embed_dim"] == 48 and info["heads_per_layer"] == [4] * 6 and info["mlp_ratio"] == 4.0
    assert teacher.feature_format == "nchw" and teacher.heads_per_layer == [1] and teacher.embed_dim == 128
    assert teacher.layer_paths == [f"stages.{i}" for i in range(4)] and not teacher.has_cls_token
    assert not any(p.requires_grad for p in teacher.model.parameters()) and not teacher.model.training
    # DeiT-S / ResNet-50 at the BASELINE shapes (meta device: nothing is allocated)
    with torch.device("meta"):
        deit_s = SM.StockViT()
        r50 = SM.StockResNet()
    assert sum(p.numel() for p in deit_s.parameters()) == 22_050_664          # the all-reduce volume bench.py uses
    assert r50.num_features == 2048 and sum(p.numel() for p in r50.parameters()) == 23_508_032


def test_mixup_cutmix_targets():
    from basd_amd import trainer as T
    from tools import stock_models as SM
    torch.manual_seed(0)
    x = torch.randn(8, 3, 16, 16)
    y = torch.arange(8) % 5
    for _ in range(6):
        xm, ym = T.mixup_cutmix(x, y, 5)
        assert xm.shape == x.shape and ym.shape == (8, 5)
        assert torch.allclose(ym.sum(1), torch.ones(8)) and (ym >= 0).all()


@pytest.mark.parametrize("kind", ["cnn", "vit"])
def test_train_step_on_cpu_with_the_oracle_loss(kind):
    from basd_amd import trainer as T
    from tools import stock_models as SM
    student, teacher = _toy_models(kind)
    torch.manual_seed(42)
    tr = T.Trainer(student, _config(), teacher, student_info=SM.probe_model(student, 32), loss_cls=OracleBASD)
    assert tr.basd_loss.token_layers == [0, 2, 3, 5]
    assert len(tr.optimizer.param_groups) == 2 and tr.optimizer.param_groups[1]["params"][0] is tr.basd_loss.log_temperatures
    before = [p.detach().clone() for p in student.parameters()]
    batch = {"clean": _structured_images(8, 32, 1), "augmented": _structured_images(8, 32, 2),
             "label": torch.arange(8) % 10}
    out = tr.train_step(batch)
    assert torch.isfinite(out["loss"]) and out["n"] == 8
    assert any(not torch.equal(a, b) for a, b in zip(before, student.parameters()))
    assert all(p.grad is None or not p.grad.any() for p in student.parameters())      # zero_grad ran


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    from basd_amd import trainer as T
    from tools import stock_models as SM
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        student, teacher = _toy_models("vit")
        torch.manual_seed(100 + rank)                  # different seeds: the constructor must make the replicas equal
        for p in student.parameters():
            p.data.add_(0.01 * torch.randn_like(p))
        tr = T.Trainer(student, _config(), teacher, student_info=SM.probe_model(student, 32), loss_cls=OracleBASD,
                       mixup=False)
        data = torch.utils.data.TensorDataset(_structured_images(16, 32, 7), torch.arange(16) % 10)
        loader = T.shard_loader(data, 4, shuffle=False)
        seen = []
        for x, y in loader:
            seen += y.tolist()
            tr.train_step({"clean": x, "augmented": x, "label": y})
        flat = torch.cat([p.detach().reshape(-1) for p in student.parameters()] + [tr.basd_loss.log_temperatures.detach()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        # every gradient still IS its slice of the flat bucket (no pack / unpack copies were needed)
        buf = tr._bucket.buffer
        lo, hi = buf.data_ptr(), buf.data_ptr() + 4 * buf.numel()
        in_bucket = all(p.grad is not None and lo <= p.grad.data_ptr() < hi
                        for p in list(student.parameters()) + [tr.basd_loss.log_temperatures])
        # a shuffling sharded loader: ``_train_epoch`` hands the epoch to its sampler, so two epochs see different orders
        imgs, labels = _structured_images(16, 32, 9), torch.arange(16) % 10

        class Recording(torch.utils.data.Dataset):
            seen: list = []

            def __len__(self):
                return 16

            def __getitem__(self, i):
                self.seen.append(int(i))
                return {"clean": imgs[i], "augmented": imgs[i], "label": labels[i]}

        rec = Recording()
        shuffled = T.shard_loader(rec, 4, seed=3)
        orders = []
        for epoch in range(2):
            rec.seen = []
            tr._train_epoch(shuffled, epoch)
            orders.append(list(rec.seen))
        out[rank] = (all(torch.equal(g, gathered[0]) for g in gathered), len(seen),
                     float((tr.basd_loss.log_temperatures.detach() - 0.5413).abs().max()), in_bucket,
                     tr.reattached, orders[0] != orders[1], orders[0])
    finally:
        dist.destroy_process_group()


def test_two_ranks_stay_identical_including_the_temperatures():
    """Two gloo ranks, sharded loader, two steps each: student parameters AND the selector temperatures (which the
    reference never reduces) are bit-identical afterwards, and the temperatures moved (multi-layer ViT teacher)."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rank_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for same, n_seen, moved, in_bucket, reattached, reshuffled, _ in out.values():
        assert same and n_seen == 8 and moved > 1e-5
        assert in_bucket and reattached == 0            # gradient views: nothing was copied into or out of the bucket
        assert reshuffled
    assert not set(out[0][6]) & set(out[1][6])          # the two ranks' shards are disjoint


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,autocast", [("cnn", None), ("vit", None), ("vit", torch.bfloat16)])
def test_train_step_on_the_hip_library(dev, kind, autocast):
    """A full step on the GPU; the loss of the captured tensors agrees with the oracle (fp32 capture) and the
    checkpoint round trip restores the step counter and the selector state."""
    from basd_amd import trainer as T, capture
    from tools import stock_models as SM
    from oracle import basd_oracle as O
    student, teacher = _toy_models(kind, dev)
    torch.manual_seed(42)
    tr = T.Trainer(student, _config(), teacher, student_info=SM.probe_model(student, 32), autocast_dtype=autocast,
                   mixup=False)
    batch = {"clean": _structured_images(16, 32, 1), "augmented": _structured_images(16, 32, 2),
             "label": torch.arange(16) % 10}
    if autocast is None:
        with torch.no_grad():
            logits, s_tok = capture._extract_student(student, batch["augmented"].to(dev), tr.basd_loss.token_layers,
                                                     layer_paths=tr._student_layer_paths, has_cls_token=True)
            t_tok, t_att = capture.extract_intermediates(teacher, batch["clean"].to(dev))
            st = O.SelectorState(tr.basd_loss.layer_selector.proj_s.cpu(), tr.basd_loss.layer_selector.proj_t.cpu(),
                                 tr.basd_loss.layer_selector.log_temperatures.detach().cpu())
            ref = O.basd_forward(st, tr.criterion, tr.basd_loss.token_layers, 16, teacher.has_cls_token, logits.cpu(),
                                 batch["label"], {k: v.cpu().contiguous() for k, v in s_tok.items()},
                                 {k: v.cpu().contiguous() for k, v in t_tok.items()},
                                 {k: v.cpu().contiguous() for k, v in t_att.items()})[0]
    before = [p.detach().clone() for p in student.parameters()]
    out = tr.train_step(batch)
    assert torch.isfinite(out["loss"])
    if autocast is None:
        assert abs(out["loss"].item() - ref.item()) <= 1e-4 * abs(ref.item()), (out["loss"].item(), ref.item())
    assert any(not torch.equal(a, b) for a, b in zip(before, student.parameters()))
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"basd_trainer_{kind}_{os.getpid()}.pth")
    tr.save_checkpoint(path, epoch=3)
    tr.basd_loss.layer_selector.log_temperatures.data.add_(1.0)
    assert tr.load_checkpoint(path) == 4
    os.remove(path)
    assert (tr.basd_loss.layer_selector.log_temperatures - 0.5413).abs().max() < 1e-2
