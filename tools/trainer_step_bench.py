"""A real DeiT-S <- ResNet-50 distillation step at batch 256 (stock torch models, random init, synthetic images) around
the HIP loss path: time of the whole step and of its parts, so that the share of the loss is a measured number.
usage: trainer_step_bench.py [--batch 256] [--steps 8] [--dtype bf16|fp32]"""
import argparse, os, sys, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
import torch
from basd_amd import trainer as T, capture
from tools import stock_models as SM

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
student = SM.StockViT().to(dev)                      # DeiT-S
teacher = SM.make_teacher(SM.StockResNet().to(dev), 224)
cfg = SimpleNamespace(training=SimpleNamespace(label_smoothing=0.1, learning_rate=1e-3, weight_decay=0.05),
                      basd=SimpleNamespace(num_extraction_points=4), model=SimpleNamespace(num_classes=1000))
torch.manual_seed(42)
ac = torch.bfloat16 if args.dtype == "bf16" else None
tr = T.Trainer(student, cfg, teacher, student_info=SM.probe_model(student, 224), autocast_dtype=ac, mixup=True)
g = torch.Generator().manual_seed(1)
B = args.batch
# images with per-image structure (a random colour cast + noise) so that the teacher features are not pure noise
imgs = (torch.randn(B, 3, 1, 1, generator=g) * 2 + torch.randn(B, 3, 224, 224, generator=g)).to(dev)
batch = {"clean": imgs, "augmented": imgs.flip(3), "label": torch.randint(0, 1000, (B,), generator=g).to(dev)}


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(2):
    out = tr.train_step(batch)
step_ms = timed(lambda: tr.train_step(batch), args.steps)
acx = torch.autocast("cuda", dtype=ac, enabled=ac is not None)


def fwd_student():
    with acx:
        return capture._extract_student(tr.model, batch["augmented"], tr.basd_loss.token_layers,
                                        layer_paths=tr._student_layer_paths, has_cls_token=True)


def fwd_teacher():
    with acx:
        return capture.extract_intermediates(teacher, batch["clean"])


with torch.no_grad():
    s_ms = timed(fwd_student, args.steps)
t_ms = timed(fwd_teacher, args.steps)
logits, s_tok = fwd_student()
t_tok, t_att = fwd_teacher()
s_leaf = {k: v.detach().requires_grad_(True) for k, v in s_tok.items()}
lg = logits.detach().float().requires_grad_(True)


def loss_only():
    loss = tr.basd_loss(lg, batch["label"], s_leaf, t_tok, t_att)
    loss.backward()


loss_only()
l_ms = timed(loss_only, args.steps)
print({"batch": B, "dtype": args.dtype, "step_ms": round(step_ms, 2), "images_per_s": round(B / step_ms * 1e3, 1),
       "student_fwd_nograd_ms": round(s_ms, 2), "teacher_fwd_ms": round(t_ms, 2), "loss_fwd_bwd_ms": round(l_ms, 2),
       "loss_share": round(l_ms / step_ms, 3), "loss": float(out["loss"]), "ranks": dict(tr.basd_loss.layer_selector.subspace_ranks)})
