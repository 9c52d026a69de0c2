"""Parameter table of the MOF network, seeded checkpoints, and NHWC weight packing.

The reference loads `VideoFlow_ckpt/{ARCH}_{dataset}[_288960noise].pth` with
`torch.load` + strict `load_state_dict` after stripping a `module.` prefix
(processing/videoflow_core.py:79-85,104-110).  The published checkpoints are not available
(.MISSING_LARGE_BLOBS), so `write_seeded_checkpoint` produces a deterministic stand-in with the
same file name and key layout (`fnet.*`, `cnet.*`, `update_block.{encoder,tprop,gru,flow_head,mask}.*`).
"""
import math
import os

import torch


def _encoder_spec(prefix, out_dim):
    s = [(f"{prefix}.conv1", 64, 3, 7, 7)]
    cin = 64
    for li, (planes, stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
        for bi in range(2):
            st = stride if bi == 0 else 1
            s.append((f"{prefix}.layer{li}.{bi}.conv1", planes, cin, 3, 3))
            s.append((f"{prefix}.layer{li}.{bi}.conv2", planes, planes, 3, 3))
            if st != 1:
                s.append((f"{prefix}.layer{li}.{bi}.downsample.0", planes, cin, 1, 1))
            cin = planes
    s.append((f"{prefix}.conv2", out_dim, 128, 1, 1))
    return s


BASE_CORR_LEVELS, BASE_CORR_RADIUS = 4, 4   # what a checkpoint's correlation-input weights are shaped for


def corr_channel_subset(levels, radius, base_levels=BASE_CORR_LEVELS, base_radius=BASE_CORR_RADIUS):
    """Channels of a base (4-level, radius-4) lookup that a smaller lookup produces, in its own order:
    level l < levels, window offsets within +-radius (the centred sub-window).  The reference's --fast
    mode lowers corr_levels / corr_radius on the cfg of an already-trained network
    (processing/videoflow_core.py:91-94); the checkpoint keeps its shapes, the engine uses the matching
    input columns of the first motion-encoder convolution."""
    if levels > base_levels or radius > base_radius:
        raise ValueError(f"corr_levels <= {base_levels} and corr_radius <= {base_radius} required")
    bw, d = 2 * base_radius + 1, base_radius - radius
    return [l * bw * bw + (i + d) * bw + (j + d)
            for l in range(levels) for i in range(2 * radius + 1) for j in range(2 * radius + 1)]


def conv_spec(cfg):
    """[(name, cout, cin, kh, kw)] for every convolution, in state-dict order (checkpoint shapes: the
    correlation input is always the base 4-level radius-4 lookup)."""
    cor = BASE_CORR_LEVELS * (2 * BASE_CORR_RADIUS + 1) ** 2
    hid = cfg.feat_dim // 2
    s = _encoder_spec("fnet", cfg.feat_dim) + _encoder_spec("cnet", cfg.feat_dim)
    ub = "update_block"
    s += [
        (f"{ub}.encoder.convc1", 256, 2 * cor, 1, 1),
        (f"{ub}.encoder.convc2", 192, 256, 3, 3),
        (f"{ub}.encoder.convf1", 128, 4, 7, 7),
        (f"{ub}.encoder.convf2", 64, 128, 3, 3),
        (f"{ub}.encoder.conv", 128 - 4, 192 + 64, 3, 3),
        (f"{ub}.tprop", 128, 3 * 128, 1, 1),
    ]
    gin = hid + 3 * 128
    for nm, kh, kw in (("z1", 1, 5), ("r1", 1, 5), ("q1", 1, 5), ("z2", 5, 1), ("r2", 5, 1), ("q2", 5, 1)):
        s.append((f"{ub}.gru.conv{nm}", hid, gin, kh, kw))
    s += [
        (f"{ub}.flow_head.conv1", 256, hid, 3, 3),
        (f"{ub}.flow_head.conv2", 4, 256, 3, 3),
        (f"{ub}.mask.0", 256, hid, 3, 3),
        (f"{ub}.mask.2", 2 * 64 * 9, 256, 1, 1),
    ]
    return s


def seeded_state_dict(cfg, seed=0):
    """PyTorch's default Conv2d initialisation (uniform +-1/sqrt(fan_in) for weight and bias),
    drawn layer by layer from one seeded generator, so the values depend only on (cfg, seed)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, cout, cin, kh, kw in conv_spec(cfg):
        bound = 1.0 / math.sqrt(cin * kh * kw)
        sd[f"{name}.weight"] = (torch.rand(cout, cin, kh, kw, generator=g) * 2 - 1) * bound
        sd[f"{name}.bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound
    return sd


def load_checked(model, state, what):
    """Strict `load_state_dict` (reference processing/videoflow_core.py:110) with an error that names what does not
    fit: the engine hard-codes the RAFT BasicEncoder key layout (`fnet.conv1`, `fnet.layer1.0.conv1`, ...), while the
    released MOF / BOF configurations select Twins-SVT encoders (`fnet.svt.*`) - such a checkpoint
    is refused with its unexpected key FAMILIES listed instead of hundreds of raw keys.  Numerical parity with the
    published checkpoints is unverified: they are not available here (.MISSING_LARGE_BLOBS)."""
    want = set(model.state_dict().keys())
    have = set(state.keys())
    if want != have:
        def families(keys):
            fam = {}
            for k in keys:
                parts = k.split(".")
                f = ".".join(parts[:2]) + ".*" if len(parts) > 2 else k
                fam[f] = fam.get(f, 0) + 1
            return ", ".join(f"{f} ({n})" for f, n in sorted(fam.items())) or "none"
        raise RuntimeError(
            f"{what}: checkpoint does not match the network this engine builds (CNN BasicEncoder fnet/cnet, "
            f"SepConvGRU update block).  Unexpected key families: {families(have - want)}.  Missing key families: "
            f"{families(want - have)}.  Checkpoints of the upstream Twins-SVT configurations (fnet.svt.* / cnet.svt.*) "
            f"are not supported.")
    bad = [f"{k}: checkpoint {tuple(state[k].shape)} vs network {tuple(v.shape)}"
           for k, v in model.state_dict().items() if tuple(state[k].shape) != tuple(v.shape)]
    if bad:
        raise RuntimeError(f"{what}: parameter shapes differ: " + "; ".join(bad[:8]) + (" ..." if len(bad) > 8 else ""))
    model.load_state_dict(state)


def checkpoint_name(architecture="mof", dataset="sintel", variant="standard"):
    """File name rule of reference processing/videoflow_core.py:79-85."""
    arch = architecture.upper()
    if variant == "noise" and dataset == "things":
        return f"{arch}_{dataset}_288960noise.pth"
    return f"{arch}_{dataset}.pth"


def write_seeded_checkpoint(root, cfg, seed=0, architecture="mof", dataset="sintel", variant="standard",
                            dataparallel_prefix=False):
    """Write `<root>/VideoFlow_ckpt/<name>.pth`; returns the path."""
    d = os.path.join(root, "VideoFlow_ckpt")
    os.makedirs(d, exist_ok=True)
    sd = seeded_state_dict(cfg, seed)
    if dataparallel_prefix:
        sd = {"module." + k: v for k, v in sd.items()}
    path = os.path.join(d, checkpoint_name(architecture, dataset, variant))
    torch.save(sd, path)
    return path


def pack_conv_weight(w, cin_pad=None, cblock=False):
    """[cout, cin, kh, kw] -> flat [cout][kh][kw][cin(+pad)] float32 (the kernels' K order), or with
    `cblock` the channel-block order [cout][cin/32][kh][kw][32] (cin zero padded to a multiple of 32;
    include/vfml.h VFML_KORDER_CBLOCK)."""
    cout, cin, kh, kw = w.shape
    if cblock:
        blk = 64 if cblock == 64 else 32       # cblock=64: VFML_KORDER_CBLOCK64 (cin a multiple of 64)
        cp = (cin + blk - 1) // blk * blk
        w = torch.nn.functional.pad(w.detach().to(torch.float32), (0, 0, 0, 0, 0, cp - cin))
        return w.reshape(cout, cp // blk, blk, kh, kw).permute(0, 1, 3, 4, 2).contiguous().reshape(-1)
    w = w.detach().to(torch.float32).permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad > cin:
        w = torch.nn.functional.pad(w, (0, cin_pad - cin))
    return w.contiguous().reshape(-1)
