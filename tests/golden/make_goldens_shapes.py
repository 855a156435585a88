"""Toy shapes shared by make_goldens.py and the tests (no reference import here)."""
from basd_amd import synth

TOY_VIT = synth.LossShape("toy ViT teacher", 8, 49, 96, 12, 64, 128, 6, 4, True, 50, r_s=6, r_t=0)
TOY_CNN = synth.LossShape("toy CNN teacher", 6, 36, 64, 12, 9, 160, 1, 1, False, 20, r_s=5, r_t=4)
TOY_VIT_SAME = synth.LossShape("toy ViT same grid", 4, 25, 48, 6, 25, 64, 3, 2, True, 10, points=2, r_s=4, r_t=0)
SMALL = {"vit": (TOY_VIT, 3), "cnn": (TOY_CNN, 5), "vit_same": (TOY_VIT_SAME, 9)}
