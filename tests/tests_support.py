"""Helpers shared by the GPU test modules."""
import torch


def s16_decode(flat, rows, ld, c):
    """split rows (FMT_S16) device buffer -> f32 [rows, c] on the host."""
    u = flat.view(torch.float16).view(rows, ld // 8, 2, 8).float().cpu()
    return (u[:, :, 0] + u[:, :, 1]).reshape(rows, ld)[:, :c]
