"""Analytic work model of one flow field (DESIGN.md "Algorithmic work").
FLOPs count one multiply-add as 2; bytes are the algorithmic HBM bytes of the memory-bound stages."""


def encoder_flops(H, W):
    """RAFT BasicEncoder on one H x W frame."""
    p2, p4, p8 = (H // 2) * (W // 2), (H // 4) * (W // 4), (H // 8) * (W // 8)
    f = p2 * 2 * (3 * 49) * 64
    f += 4 * p2 * 2 * (64 * 9) * 64
    f += p4 * 2 * (64 * 9) * 96 + 3 * p4 * 2 * (96 * 9) * 96 + p4 * 2 * 64 * 96
    f += p8 * 2 * (96 * 9) * 128 + 3 * p8 * 2 * (128 * 9) * 128 + p8 * 2 * 96 * 128
    f += p8 * 2 * 128 * 256
    return f


def update_flops_per_cell(cor_planes=324):
    """One iteration of the update block, per 1/8-resolution cell of one centre frame."""
    k = [(2 * cor_planes, 256), (256 * 9, 192), (4 * 49, 128), (128 * 9, 64), (256 * 9, 124), (384, 128),
         (512 * 5, 256), (512 * 5, 128), (512 * 5, 256), (512 * 5, 128), (128 * 9, 256), (256 * 9, 4)]
    return sum(2 * a * b for a, b in k)


def field_work(H, W, T=5, depth=12, levels=4, radius=4, feat=256, cached_encoders=False):
    """dict of FLOPs / bytes for one flow field of a T-frame window."""
    h, w = H // 8, W // 8
    P, M = h * w, T - 2
    S, hl, wl = [], h, w
    for _ in range(levels):
        S.append(hl * wl)
        hl, wl = hl // 2, wl // 2
    win = (2 * radius + 1) ** 2
    enc_frames = 2 if cached_encoders else (T + M)
    out = {
        "encoder_flops": enc_frames * encoder_flops(H, W),
        "corr_flops": 2 * M * 2 * P * sum(S) * feat,
        "corr_bytes": 2 * M * P * sum(S) * 4 + 2 * M * 2 * P * feat * 4,
        "update_flops": depth * M * P * update_flops_per_cell(levels * win),
        "mask_flops": M * P * 2 * (128 * 9 * 256 + 256 * 1152),
        "lookup_bytes_per_iter": 2 * M * P * (levels * (2 * radius + 2) ** 2 * 4 + levels * win * 4),
        "upsample_bytes": P * 576 * 4 + P * 2 * 4 + 64 * P * 2 * 4,
    }
    out["total_flops"] = out["encoder_flops"] + out["corr_flops"] + out["update_flops"] + out["mask_flops"]
    return out
