"""MOFNetHIP — the multi-frame optical-flow network executed by the vfml HIP kernels.

`build_network(cfg)` stands in for `core.Networks.build_network` of the VideoFlow submodule
(reference processing/videoflow_core.py:28,101).  The returned module honours the call the
reference makes at :188, `model(images[B,N,3,H,W], {}) -> (flow[B,2(N-2),2,H,W], aux)`, with the
forward flows of the N-2 centre frames first and the backward flows after them, so that the
reference's `flow[0, shape[1]//2]` pick (:194-195) lands on the same field.

Python only sequences kernels; all arithmetic runs in libvfml_hip.so (include/vfml.h) on NHWC
fp32 buffers.  There is no CPU execution path in this module: on a non-GPU device `forward`
raises.

Pipeline per call (names from SURVEY.md §2b):
  K1 frames -> NHWC4, normalised           vfml_frames_to_nhwc4
  K2 fnet (N frames) / cnet (N-2 frames)   vfml_conv2d + vfml_instnorm_{stats,apply}
  K3/K4 correlation pyramid, 2(N-2) problems: level l = GEMM of centre features against the
        2^l-pooled target features (avg-pool commutes with the dot product)   vfml_avgpool2x2, vfml_conv2d
  loop decoder_depth times:
     K5 lookup (fwd, bwd)                  vfml_corr_lookup
     K6 motion encoder, temporal fusion, SepConvGRU (gates fused into the conv epilogues),
        flow head                          vfml_conv2d, vfml_coords_update
  mask head (last iteration only) + K8 8x convex upsampling   vfml_conv2d, vfml_convex_upsample
"""
import os

import torch
import torch.nn as nn

from . import hip
from .weights import conv_spec, corr_channel_subset, pack_conv_weight


def cor_pad(cor, split):
    """Channels of one direction's lookup block in the motion encoder's input: whole float4s (f32 path), or, in split
    rows, a multiple of 16 - the two directions then make whole 32-channel K steps (324 -> 336, K = 672) and the 1x1
    `convc1` takes the uniform-step loader of the LDS-DMA kernel (340 instead of 210 TFLOP/s on it at 1080p); the pad
    channels are zero on both sides (the buffer is zero-filled once, the weights are zero there)."""
    return (cor + 15) // 16 * 16 if split else (cor + 3) // 4 * 4


def take_frames(src, idx):
    """src[idx] along dim 0 without a host->device index upload (a pageable H2D copy would make the
    host wait for all queued GPU work): a view when idx is a contiguous run, device-side copies otherwise."""
    idx = list(idx)
    if all(b == a + 1 for a, b in zip(idx, idx[1:])):
        return src[idx[0]:idx[0] + len(idx)]
    return torch.stack([src[i] for i in idx])


class _Holder(nn.Module):
    """Parameter container; children are created on demand so that dotted checkpoint keys
    (`fnet.layer2.0.downsample.0.weight`) map onto a module tree for load_state_dict."""

    def child(self, name):
        if name not in self._modules:
            self.add_module(name, _Holder())
        return self._modules[name]


class MOFNetHIP(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.hidden_dim = self.context_dim = cfg.feat_dim // 2
        if cfg.feat_dim != 256:
            raise ValueError("MOFNetHIP is built for feat_dim=256")
        # 'BOFNet': the tri-frame member of the family - the same blocks on the centre triple of the
        # window (previous, current, next), output [B, 2, 2, H, W] = (forward, backward) of the centre frame
        self.tri_frame = getattr(cfg, "network", "MOFNetStack") == "BOFNet"
        self._spec = conv_spec(cfg)
        for name, cout, cin, kh, kw in self._spec:
            leaf = self
            for p in name.split("."):
                leaf = leaf.child(p)
            leaf.weight = nn.Parameter(torch.zeros(cout, cin, kh, kw), requires_grad=False)
            leaf.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self._packed = None
        self._packed_key = None
        self._packed_serial = 0
        self._ws = {}
        self._graphs = {}
        self._side2 = {}      # per device: the stream of the flow half of the motion encoder (_run body)
        self._pyr_free = []
        self._pyr_busy = set()    # cache keys of the pyramids the field in flight reads (a prefetch must not recycle them)
        self._side = {}                # per device: the stream the next window's encoders run on (prefetch_frames)
        self._prefetch_done = None     # event: the last prefetch's launches
        self._prefetch_dev = None
        self._pre_body = None          # event: the last field's inputs are ready, its iterations not yet queued
        import collections
        self._feat_cache = collections.OrderedDict()

    # ------------------------------------------------------------------ weights
    def _param(self, name):
        node = self
        for p in name.split("."):
            node = node._modules[p]
        return node

    def _pack(self, device):
        """Repack every conv weight into the kernels' [cout][kh][kw][cin] order (once per load)."""
        split = self._split()
        key = (str(device), split, self._plan_key(), tuple(p._version for p in self.parameters()),
               tuple(p.data_ptr() for p in self.parameters()))
        if self._packed is not None and self._packed_key == key:
            return self._packed
        self._join_prefetch()          # (a prefetch in flight reads the planes this call replaces)
        P, cblock_names, cb64_names = {}, set(), set()
        cout_packed = {}       # layers whose packed matrix has another row count than their bias
        rows7 = set()          # convf1 as a 7x1 convolution over the horizontal taps' rows

        def block_of(layer, c0, ctot, cout):
            """K-axis block of a split-row layer's weight planes: 64 channels where the layer runs one MFMA per product
            over whole 64-channel blocks, more than 32 outputs wide (the kernel then steps 64 channels of hi halves at a
            time: include/vfml.h VFML_KORDER_CBLOCK64), else 32."""
            if (split and self._nm(layer) == 1 and c0 % 64 == 0 and ctot % 64 == 0 and cout > 32
                    and not os.environ.get("VFML_NO_H64")):
                cb64_names.add(layer)
                return 64
            return True
        for name, cout, cin, kh, kw in self._spec:
            leaf = self._param(name)
            w = leaf.weight.detach().to(device=device, dtype=torch.float32)
            if name.endswith(".encoder.convc1"):
                # input columns of the configured lookup (a sub-window / sub-pyramid of the checkpoint's
                # base lookup under --fast); each direction's block is padded to whole float4s /
                # split-row units with zero weights
                base = cin // 2
                sel = torch.tensor(corr_channel_subset(self.cfg.corr_levels, self.cfg.corr_radius), device=device)
                cor = sel.numel()
                cor_p = cor_pad(cor, split)
                wp = torch.zeros(cout, 2 * cor_p, 1, 1, device=device)
                wp[:, :cor] = w[:, sel]
                wp[:, cor_p:cor_p + cor] = w[:, base + sel]
                w = wp
            if (name.endswith(".flow_head.conv2") and split and cout == 4 and (kh, kw) == (3, 3)
                    and not os.environ.get("VFML_NO_TAPSUM")):       # (A/B switch)
                # 3x3 to four channels as a 1x1 to 36 (tap-major) + vfml_tapsum3x3: nine times fewer MFMA steps than a
                # 3x3 convolution padded to a 32-column tile
                w = w.permute(2, 3, 0, 1).reshape(36, cin, 1, 1).contiguous()
                cout_packed[name] = 36
            if (name.endswith(".encoder.convf1") and split and (cin, kh, kw) == (4, 7, 7)
                    and not os.environ.get("VFML_NO_ROWS7")):        # (A/B switch)
                # 7x7 over the 4-channel flow as 7x1 over 32 channels = the seven horizontal taps' quads per pixel
                # (vfml_flow_rows7): [cout][kx*4 + c][ky][1], channels 28..31 zero
                w7 = torch.zeros(cout, 32, 7, 1, device=device)
                w7[:, :28, :, 0] = w.permute(0, 3, 1, 2).reshape(cout, 28, 7)
                w = w7
                rows7.add(name)
            if name.endswith(".tprop"):
                # 1x1 conv over [prev | cur | next] motion features == 3x1 conv along the frame axis
                w = w.reshape(cout, 3, cin // 3, 1).permute(0, 2, 1, 3)  # -> [cout, cin/3, kh=3, kw=1]
            # update-block convolutions read split-row activations (all but convf1, whose input is the
            # 4-channel f32 flow), and so do the encoders behind their 4-channel stem: those weights go in
            # channel-block K order (include/vfml.h)
            cb = split and ((name.startswith("update_block.") and (not name.endswith(".convf1") or name in rows7)) or
                            (self._enc_split_rows() and (
                                (name.split(".")[0] in ("fnet", "cnet") and name.count(".") > 1) or
                                name in ("fnet.conv2", "cnet.conv2"))))
            if cb:
                cblock_names.add(name)
                cb = block_of(name, w.shape[1], w.shape[1], cout)
            P[name] = (pack_conv_weight(w, cin_pad=4 if cin == 3 else None, cblock=cb),
                       leaf.bias.detach().to(device=device, dtype=torch.float32).contiguous())
        # GRU gates.  Input channels are [h | inp | motion | temporal]; `inp` (the context map) does not
        # change over the iterations, so its part of every gate convolution (+ bias) is computed once per
        # frame and added in the epilogue ("addend"); the per-iteration convolutions see [h | motion |
        # temporal] only (K 2560 -> 1920).  z and r share their input: one conv with 2*hidden outputs.
        hid = self.hidden_dim
        self._cout_of = {}
        for k in ("1", "2"):
            self._cout_of[f"update_block.gru.convzr{k}.iter"] = 2 * hid
            self._cout_of[f"update_block.gru.convq{k}.iter"] = hid
            raw = {g: self._param(f"update_block.gru.conv{g}{k}") for g in "zrq"}
            wzr = torch.cat([raw["z"].weight, raw["r"].weight]).detach().to(device=device, dtype=torch.float32)
            bzr = torch.cat([raw["z"].bias, raw["r"].bias]).detach().to(device=device, dtype=torch.float32)
            wq = raw["q"].weight.detach().to(device=device, dtype=torch.float32)
            bq = raw["q"].bias.detach().to(device=device, dtype=torch.float32)
            for nm, wfull, bfull in ((f"update_block.gru.convzr{k}", wzr, bzr), (f"update_block.gru.convq{k}", wq, bq)):
                it = torch.cat([wfull[:, :hid], wfull[:, 2 * hid:]], dim=1)
                P[nm + ".iter"] = (pack_conv_weight(it, cblock=block_of(nm + ".iter", hid, it.shape[1], it.shape[0]) if split else False), None)
                P[nm + ".ctx"] = (pack_conv_weight(wfull[:, hid:2 * hid], cblock=block_of(nm + ".ctx", hid, hid, wfull.shape[0]) if split else False),
                                  bfull.contiguous())
                if split:
                    cblock_names.update((nm + ".iter", nm + ".ctx"))
            for g in "zrq":
                del P[f"update_block.gru.conv{g}{k}"]
        if split:
            # split-f16 planes (hi, lo*2^11) of every [cout][K] matrix, made once per load
            with torch.cuda.device(device):
                for name, (wflat, b) in list(P.items()):
                    cout = self._cout_of[name] if name in self._cout_of else cout_packed.get(name, b.numel())
                    sc = hip.SplitWeight.auto_scale(float(wflat.abs().max()))
                    sw = hip.SplitWeight(cout, wflat.numel() // cout, device).fill(wflat, scale=sc)
                    sw.order = (hip.KORDER_CBLOCK64 if name in cb64_names else
                                hip.KORDER_CBLOCK if name in cblock_names else hip.KORDER_TAP)
                    P[name] = (sw, b)
        if split and os.environ.get("VFML_STEM", "1") != "0":        # (A/B switch: the general kernel for the stems)
            for enc in ("fnet", "cnet"):
                leaf = self._param(f"{enc}.conv1")
                if tuple(leaf.weight.shape) == (64, 3, 7, 7):
                    with torch.cuda.device(device):
                        P[f"{enc}.conv1.stem"] = (hip.pack_stem_weight(leaf.weight, device),
                                                  leaf.bias.detach().to(device=device, dtype=torch.float32).contiguous())
        self._tapsum = bool(cout_packed)
        self._rows7 = bool(rows7)
        self._packed, self._packed_key = P, key
        self._graphs.clear()              # captured launches hold the old planes' addresses
        self._packed_serial += 1          # new weights: cached encoder outputs are stale
        self._feat_cache.clear()
        return P

    PRECISIONS = ("f16x3", "f16x2", "f16", "mixed", "f32")

    def _precision(self):
        p = getattr(self.cfg, "precision", "f16x3")
        if p not in self.PRECISIONS:
            raise ValueError(f"cfg.precision must be one of {self.PRECISIONS}, got {p!r}")
        return p

    def _enc_split_rows(self):
        """Encoder activations as split rows through the LDS-DMA convolution kernels (the default; `instnorm_apply` writes
        split rows, the norm statistics come from the convolution epilogues), or f32 rows split while register-staged
        (VFML_ENC_S16=0).  Measured A/B on one box at 1080p: with the per-tap kernel the split-row path was 0.25 ms per
        field SLOWER (profiles/r02_encoder_paths.md: the 64-channel layers at half resolution re-read every input pixel
        once per filter tap); with one activation stage per filter row on 256 x 64 / 256 x 96 tiles (conv_gemm_tapx.hip)
        and the fast epilogue it is 0.2 ms FASTER (26.27 / 26.26 vs 26.42 / 26.50 ms, profiles/r02_kernel_anatomy.md)."""
        return os.environ.get("VFML_ENC_S16", "1") != "0"

    def _split(self):
        """Every arithmetic but 'f32' runs the split-f16 kernels on split-row activations; they differ in the
        number of MFMAs a layer spends per product (`_nm`)."""
        return self._precision() != "f32"

    def _nm(self, layer):
        """MFMAs per product for `layer` (a conv_spec name, '.iter' / '.ctx' for the two parts of a GRU gate
        convolution, or 'corr' for the correlation GEMMs): 3 = fp32-grade split product, 2 / "2w" = weights as plain f16,
        "2a" = activations as plain f16, 1 = both operands plain f16.  'f16x3' / 'f16x2' / 'f16' = 3 / 2 / 1 everywhere; 'mixed' = cfg.mfma_plan,
        {name prefix: count}, longest prefix wins, 3 where nothing matches."""
        p = self._precision()
        if p == "f16x3":
            nm = 3
        elif p == "f16x2":
            nm = 2
        elif p == "f16":
            nm = 1
        else:
            plan = self._mixed_plan()
            best, nm = -1, 3
            for prefix, n in plan.items():
                if layer.startswith(prefix) and len(prefix) > best:
                    best, nm = len(prefix), (n if isinstance(n, str) else int(n))
            if nm == "2w":
                nm = 2
            if nm not in (1, 2, "2a", 3):
                raise ValueError(f"cfg.mfma_plan[{layer!r}] = {nm!r}: 1, 2 ('2w'), '2a' or 3")
        if layer == "corr" and nm in (2, "2a"):
            nm = 3       # a volume and its transpose must stay the same numbers: the symmetric forms only
        return nm

    def _mixed_plan(self):
        """cfg.mfma_plan, or - as vfml/cfg.py says - DEFAULT_MIXED_PLAN when precision is 'mixed' and none is given
        (an explicit empty dict means "3 everywhere")."""
        plan = getattr(self.cfg, "mfma_plan", None)
        if plan is None:
            from .cfg import DEFAULT_MIXED_PLAN
            plan = DEFAULT_MIXED_PLAN
        return plan

    def _plan_key(self):
        """The arithmetic as part of a cached frame's identity."""
        p = self._precision()
        vol = getattr(self.cfg, "corr_volume", "f32")
        if vol not in self.CORR_VOLUMES:
            raise ValueError(f"cfg.corr_volume must be one of {tuple(self.CORR_VOLUMES)}, got {vol!r}")
        key = (p, tuple(sorted((k, str(v)) for k, v in self._mixed_plan().items()))) if p == "mixed" else p
        if self._tile() is None:
            key = (key, "row-major")
        return key if vol == "f32" else (key, vol)

    # cfg.corr_volume -> mask of the pyramid levels stored as one f16 per value (bit l = level l)
    CORR_VOLUMES = {"f32": 0, "f16": 15, "f16@1": 14, "f16@2": 12, "f16@3": 8}
    _VOL_TILE = hip.VolTile(2, 3)

    def _tile(self):
        """Layout of the correlation volumes: 4 x 8 tiles (hip.VolTile) on the split-row path - a lookup window then lies in
        ~8 lines of 128 bytes instead of ~13, 149 -> 107 us per lookup at 1080p (tools/exp/lookup_tiled.py) - for the price
        of whole edge tiles (+3.4 % GEMM columns at 1080p).  VFML_VOL_TILE=0: row-major (A/B switch; same fields)."""
        if not self._split() or os.environ.get("VFML_VOL_TILE", "1") == "0":
            return None
        return self._VOL_TILE

    # ------------------------------------------------------------------ workspace
    def _buf(self, name, numel, device, dtype=torch.float32, zero=False):
        """Named scratch buffer, cached per exact size (a resolution change reallocates)."""
        t = self._ws.get(name)
        if t is None or t.numel() != numel or t.device != device or t.dtype != dtype:
            if t is not None:
                self._graphs.clear()       # captured launches hold this buffer's address
                self._join_prefetch()      # (the encoders' workspaces are shared with a prefetch in flight)
            t = (torch.zeros if zero else torch.empty)(int(numel), device=device, dtype=dtype)
            self._ws[name] = t
        return t

    _prefetch_done = None      # (class defaults: engines that share this class's plumbing but never prefetch)
    _prefetch_dev = None

    def _join_prefetch(self):
        """Order the current stream behind a prefetch in flight (prefetch_frames): it runs on a side stream over workspaces,
        weights and cache entries that were allocated under this one - whatever frees or replaces any of them, or reads what
        the prefetch produces, has to come after it."""
        if self._prefetch_done is not None:
            torch.cuda.current_stream(self._prefetch_dev).wait_event(self._prefetch_done)
            self._prefetch_done = None

    def release_workspace(self):
        self._join_prefetch()
        self._graphs.clear()
        self._ws.clear()
        self._feat_cache.clear()
        self._ctx_store = None

    # ------------------------------------------------------------------ per-frame context parts of the gate convolutions
    # The context part of the four GRU gate convolutions is computed once per frame and ADDED by the gate convolutions of
    # every iteration of every window the frame is a centre of (vfml_conv_desc.addend): [centre frames][cells][256 | 128]
    # floats, 300 MB per 1080p window.  They live in ONE store per gate, a ring of frame slots handed out in the order frames
    # arrive: the centre frames of a sliding job's window are then consecutive slots, and the gate convolutions - fixed
    # launches of a replayed graph - reach them through a device cell that holds the window's first slot
    # (vfml_conv_desc.addend_ind).  Nothing is gathered per field (round 2: twelve device copies, 0.6 GB of traffic).  The
    # first slots are mirrored behind the ring so that a window that wraps is consecutive too; a window whose centres are not
    # in arrival order (random access) is gathered into slots kept for that.
    CTX_RING = 12            # >= FEATURE_CACHE_FRAMES: a cached context entry owns its slot until the ring comes round
    CTX_GATES = (("zr1", 256), ("q1", 128), ("zr2", 256), ("q2", 128))

    def _context_store(self, dev, Pn, M):
        st = getattr(self, "_ctx_store", None)
        if st is None or st["key"] != (str(dev), Pn) or st["M"] < M:
            self._join_prefetch()
            # (a coming window's centres are prefetched while the current one's are being read: two windows' worth and some)
            R = max(self.CTX_RING, 2 * M + 4)
            for k in [k for k in self._feat_cache if k[0] == "c"]:     # (entries of another store)
                del self._feat_cache[k]
            st = self._ctx_store = {
                "key": (str(dev), Pn), "M": M, "R": R, "next": 0, "owner": [None] * R, "live": set(),
                "serial": getattr(self, "_ctx_serial", 0) + 1,
                "S": {g: torch.empty((R + 2 * M - 1) * Pn * co, device=dev) for g, co in self.CTX_GATES}}
            self._ctx_serial = st["serial"]
        return st

    def _context_slot(self, st, cache_key, busy):
        """The next slot of the ring (not one of `busy`); whoever owned it loses its cache entry."""
        R = st["R"]
        busy = set(busy) | st["live"]          # (the last window's centres may still be read by its launches)
        for _ in range(R):
            s = st["next"]
            st["next"] = (s + 1) % R
            if s not in busy:
                break
        else:
            raise RuntimeError("context store: every slot is in use by the current window")
        old = st["owner"][s]
        if old is not None:
            self._feat_cache.pop(old, None)
        st["owner"][s] = cache_key
        return s

    # ------------------------------------------------------------------ graph replay of the iteration body
    GRAPHS_KEPT = 8

    def _use_graph(self):
        use = getattr(self.cfg, "use_graph", None)
        if use is None:
            use = os.environ.get("VFML_GRAPH", "1") != "0"
        return bool(use) and hip._PROFILE is None       # (per-launch events of the roofline pass cannot be captured)

    def _run_body(self, body, key, dev):
        """Run `body` (a fixed sequence of kernel launches on fixed buffers): eagerly the first time a configuration
        is seen (workspaces get allocated, kernel attributes set), captured into a HIP graph the second time, replayed
        afterwards.  The launches go to torch's current stream through the C ABI, which during capture is the capture
        stream.  Anything that moves a buffer the launches address (a workspace reallocation, new weights) drops the
        graphs."""
        if not self._use_graph():
            body()
            return
        st = self._graphs.get(key)
        if st is None:
            body()
            while len(self._graphs) >= self.GRAPHS_KEPT:
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = "seen"
            return
        if st == "seen":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                body()
            self._graphs[key] = st = g
        st.replay()

    # ------------------------------------------------------------------ encoder
    def _encoder(self, prefix, x, n, H, W, P, dev, out, ldo, out_off, epilogue, split, out_fmt=hip.FMT_F32):
        """RAFT BasicEncoder on NHWC4 input x [n,H,W,4]; writes [n,H/8,W/8,256] into `out`."""
        h2, w2 = H // 2, W // 2
        big = n * h2 * w2 * 64
        raw = self._buf("enc_raw", big, dev)
        raw2 = self._buf("enc_raw2", big, dev)
        act = [self._buf("enc_act0", big, dev), self._buf("enc_act1", big, dev), self._buf("enc_act2", big, dev)]
        st = self._buf("enc_stats", 3 * n * 128 * 2, dev)
        nws = hip.instnorm_workspace_bytes(n, h2 * w2, 128)
        ws = self._buf("enc_statws", (nws + 7) // 8, dev, torch.float64)

        def stats_of(t, hw, c, slot):
            s = st[slot * n * 128 * 2:]
            hip.instnorm_stats(t, n, hw, c, s, ws)
            return s

        # split-f16 path: activations are split rows (written by instnorm_apply, read by the LDS-DMA convolution
        # kernel); every convolution leaves per-block sums of its raw result behind (its tile is in LDS anyway) and
        # only the fold remains.  Blocks (128 output pixels after the f32-source stem, 32 after split-row sources)
        # must not straddle frames.
        split_prec = self._split()
        AF = hip.FMT_S16 if (split_prec and self._enc_split_rows()) else hip.FMT_F32
        part_len = n * ((h2 * w2 + 31) // 32) * 64 * 2            # largest layer: half resolution, 64 channels
        parts = self._buf("enc_part", 3 * part_len, dev, torch.float64) if split_prec else None
        # (the norm fold's first pass; like every encoder workspace it belongs to this engine, and whatever runs the
        # encoders on another stream - prefetch_frames - is ordered against the main stream's use by _join_prefetch)
        fold_ws = self._buf("enc_foldws", min(n, 8) * 64 * 128 * 2, dev, torch.float64) if split_prec else None

        enc_c64 = os.environ.get("VFML_ENC_C64", "1") != "0"         # (A/B switch: the general kernel for layer1's convolutions)

        def conv_stats(src, c, hh_, ww_, name, planes, dst, slot, k, stride=1, pad=0, src_fmt=hip.FMT_F32):
            wgt, b = P[name]
            nm = self._nm(name) if split_prec else 3
            ho_, wo_ = (hh_ + 2 * pad - k) // stride + 1, (ww_ + 2 * pad - k) // stride + 1
            hw = ho_ * wo_
            rows = hip.STATS_ROWS_S16 if src_fmt == hip.FMT_S16 else hip.STATS_ROWS_F32
            if not (split_prec and (n == 1 or hw % rows == 0)):
                hip.conv2d(src, c, c, n, hh_, ww_, wgt, b, planes, k, k, dst, planes, stride=stride, pad_h=pad, pad_w=pad,
                           mfma=nm, in_fmt=src_fmt)
                return stats_of(dst, hw, planes, slot)
            chunks = (hw + rows - 1) // rows
            part = parts[slot * part_len:]
            if (enc_c64 and c == 64 and planes == 64 and k == 3 and stride == 1 and pad == 1 and nm == 3 and src_fmt == hip.FMT_S16
                    and ww_ % 32 == 0 and getattr(wgt, "order", None) == hip.KORDER_CBLOCK and wgt.kp == 576 and wgt.lo is not None):
                # the residual blocks of layer1: persistent workgroups, weights in registers, one patch per tile (same bits)
                hip.conv3x3_c64(src, c, n, hh_, ww_, wgt, b, dst, planes, stats_part=part)
            else:
                hip.conv2d(src, c, c, n, hh_, ww_, wgt, b, planes, k, k, dst, planes, stride=stride, pad_h=pad, pad_w=pad,
                           stats_part=part, mfma=nm, in_fmt=src_fmt)
            s = st[slot * n * 128 * 2:]
            hip.instnorm_finalize(part, n, chunks, planes, hw, s, workspace=fold_ws)
            return s

        stem = P.get(f"{prefix}.conv1.stem")
        if stem is not None and split_prec and self._nm(f"{prefix}.conv1") == 3:
            # the 7x7 / 2 stem as one patch-resident kernel (csrc/stem.hip): its statistics partials are one chunk per
            # 8 x 64-pixel output tile
            chunks = hip.stem_chunks(H, W)
            part = parts[0:]
            hip.stem7x7s2(x, n, H, W, stem[0], stem[1], raw, stats_part=part)
            s0 = st[0:]
            hip.instnorm_finalize(part, n, chunks, 64, h2 * w2, s0, workspace=fold_ws)
        else:
            s0 = conv_stats(x, 4, H, W, f"{prefix}.conv1", 64, raw, 0, 7, stride=2, pad=3)
        cur = act[0]
        hip.instnorm_apply(raw, s0, n, h2 * w2, 64, cur, out_fmt=AF)
        ch, hh, ww, ci = 64, h2, w2, 0
        for li, (planes, stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
            for bi in range(2):
                stv = stride if bi == 0 else 1
                ho, wo = (hh, ww) if stv == 1 else ((hh - 1) // 2 + 1, (ww - 1) // 2 + 1)
                name = f"{prefix}.layer{li}.{bi}"
                y = act[(ci + 1) % 3]
                nxt = act[(ci + 2) % 3]
                s1 = conv_stats(cur, ch, hh, ww, f"{name}.conv1", planes, raw, 0, 3, stride=stv, pad=1, src_fmt=AF)
                hip.instnorm_apply(raw, s1, n, ho * wo, planes, y, out_fmt=AF)
                s2 = conv_stats(y, planes, ho, wo, f"{name}.conv2", planes, raw, 1, 3, pad=1, src_fmt=AF)
                if stv != 1:
                    s3 = conv_stats(cur, ch, hh, ww, f"{name}.downsample.0", planes, raw2, 2, 1, stride=stv, src_fmt=AF)
                    hip.instnorm_apply(raw, s2, n, ho * wo, planes, nxt, res=raw2, res_stats=s3, out_fmt=AF)
                else:
                    hip.instnorm_apply(raw, s2, n, ho * wo, planes, nxt, res=cur, out_fmt=AF)
                cur, ci = nxt, (ci + 2) % 3
                ch, hh, ww = planes, ho, wo
        wgt, b = P[f"{prefix}.conv2"]
        hip.conv2d(cur, 128, 128, n, hh, ww, wgt, b, 256, 1, 1, out, ldo, out_off=out_off, epilogue=epilogue,
                   split=split, out_fmt=out_fmt, in_fmt=AF, mfma=self._nm(f"{prefix}.conv2") if split_prec else 3)
        return hh, ww

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, images, data=None, return_lowres=True):
        """images: float [B=1, N, 3, H, W] on the GPU (the tensor the reference builds at
        processing/videoflow_processor.py:160-161). Returns (flow [1, 2(N-2), 2, H, W], low-res flows)."""
        if not isinstance(images, torch.Tensor) or not images.is_cuda:
            raise RuntimeError("MOFNetHIP runs on an MI355X (HIP) device only; got "
                               f"{getattr(images, 'device', type(images))}. There is no CPU fallback in the shipped engine.")
        if images.dim() != 5 or images.shape[2] != 3:
            raise ValueError(f"images must be [B,N,3,H,W], got {tuple(images.shape)}")
        if images.shape[0] != 1:
            raise ValueError(f"Batch size must be 1, got {images.shape[0]}")
        src = images[0]
        if src.dtype != torch.float32:
            src = src.float()
        return self._run(src.contiguous(), src.shape[0], src.shape[2], src.shape[3], return_lowres)

    @torch.no_grad()
    def forward_u8(self, frames, return_lowres=True, frame_keys=None, tri_batch=False, pick_only=False):
        """frames: uint8 [N, H, W, 3] RGB on the GPU; /255 happens in the K1 kernel (same fp32 ops
        as the reference's host-side conversion), saving the 4x larger float upload.

        frame_keys: optional list of N hashable ids, one per frame.  The encoders act on each frame
        independently (instance norm is per image), so their outputs depend on the frame alone:
        with keys the engine keeps the per-frame feature maps, pooled target pyramids and context
        maps of recently seen frames and encodes only the frames it has not seen - in a sliding
        window that is 1 of N.  The caller promises that equal keys mean equal pixels.

        tri_batch (tri-frame networks, `--vf-architecture bof`): the N frames are B = N-2 overlapping triples
        (j-1, j, j+1) - B consecutive fields of a job in one pass, every centre frame its own problem; the
        output holds their 2B flows (forward flows first), field j's pick is flow B + j.  Same values as B
        separate calls; at 720p one triple fills a third of the chip.

        pick_only (with return_lowres=False): return only the flow the reference takes from the output,
        `[0, shape[1]//2]` = the backward flow of the first centre frame, as [1, 1, 2, H, W]; the last
        iterations then skip the centre frames that can no longer influence it (same bits for that flow)."""
        if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8
                and frames.dim() == 4 and frames.shape[3] == 3):
            raise ValueError("forward_u8 expects a uint8 [N,H,W,3] device tensor")
        if frame_keys is not None and len(frame_keys) != frames.shape[0]:
            raise ValueError("frame_keys must have one entry per frame")
        if tri_batch and not self.tri_frame:
            raise ValueError("tri_batch applies to tri-frame (BOF) networks")
        return self._run(frames.contiguous(), frames.shape[0], frames.shape[1], frames.shape[2], return_lowres,
                         frame_keys, tri_batch, pick_only and not return_lowres)

    # ------------------------------------------------------------------ per-frame encoder cache
    FEATURE_CACHE_FRAMES = 12
    FMAP_ROW_SCALE = 16.0       # power of two: exact; undone by the correlation GEMM's out_scale

    def clear_feature_cache(self):
        self._join_prefetch()
        self._feat_cache.clear()
        self._pyr_free = []

    def prefetch_frames(self, frames, frame_keys):
        """The encoders a COMING window will need - feature maps / target planes of its frames, context maps of its centre
        frames, whatever of them is not cached yet - queued on a side stream, behind the inputs of the field just launched
        and BESIDE its update iterations: the encoders' short-K convolutions and HBM-bound norm passes fill what the
        iterations' MFMA-bound launches leave idle (tails of half-empty rounds, memory bandwidth).  Same kernels, same
        inputs: the cached results are the bits a later forward_u8 would compute itself.  frames: uint8 [N,H,W,3] of the
        coming window, frame_keys as for forward_u8.

        ON by default (VFML_PREFETCH=0 turns it off; VFML_PREFETCH_DBG=serial queues the same work BEHIND the field -
        what bench.py's per-launch roofline pass does, so that every timed launch has the GPU to itself).  Fields are
        bit-identical with and without it (tests/test_gpu_e2e.py::test_prefetched_encoders_give_the_same_fields); they were
        not while the fixed-radius lookup mixed its samples with a packed-f32 op straight behind a ds_read, which reads a
        stale register when another kernel's MFMAs share the SIMD (profiles/r02_kernel_anatomy.md section 7)."""
        if (frame_keys is None or self._pre_body is None or os.environ.get("VFML_PREFETCH", "1") == "0" or self.tri_frame):
            return
        cfg = self.cfg
        N, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        if N < 3 or H % 8 or W % 8 or len(frame_keys) != N:
            return
        L, dev = cfg.corr_levels, frames.device
        h, w = H // 8, W // 8
        AF = hip.FMT_S16 if self._split() else hip.FMT_F32
        geo = self._volume_geometry(H, W, L, AF)
        hl, wl, Sl = geo.hl, geo.wl, geo.Sl
        keys = [(k, H, W, L, self._plan_key(), self._packed_serial) for k in frame_keys]
        need_f = [j for j in range(N) if ("f", keys[j]) not in self._feat_cache]
        need_c = [j for j in range(1, N - 1) if ("c", keys[j]) not in self._feat_cache]
        # ... and the window's new correlation pyramids: HBM-write-bound GEMMs beside MFMA-bound iterations
        need_p = self._prefetch_pyramids() and any(("p", keys[c], keys[t]) not in self._feat_cache
                                                   for c in range(1, N - 1) for t in (c + 1, c - 1))
        if not need_f and not need_c and not need_p:
            return
        main = torch.cuda.current_stream(dev)
        side = self._side.get(dev)
        if side is None:
            # (VFML_PREFETCH_PRIO: HIP stream priority of the prefetch stream - larger = lower; A/B switch)
            side = self._side[dev] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("VFML_PREFETCH_PRIO", "0")))
        P = self._pack(dev)
        dbg = os.environ.get("VFML_PREFETCH_DBG", "")
        if "serial" in dbg:
            side.wait_stream(main)
        else:
            side.wait_event(self._pre_body)
        if "nof" in dbg:
            need_f = []
        if "noc" in dbg:
            need_c = []
        frames.record_stream(side)       # (a view of the caller's clip: its block must outlive the side stream's reads)
        with torch.cuda.stream(side):
            new = []
            feats = None
            if need_f or need_p:
                feats = self._frame_features(frames, list(range(N)) if need_p else need_f, keys, H, W, P, dev, L, hl, wl, Sl,
                                             vt=geo.VT)
                new += list(feats.values())
            if need_c:
                new += list(self._frame_context(frames, need_c, keys, H, W, P, dev, h * w, N - 2).values())
            if need_p:
                new.append(self._window_pyramids(feats, keys, N, geo, dev, protect=self._pyr_busy))
            done = torch.cuda.Event()
            done.record(side)
        # the cached tensors were allocated on the side stream and will be read (and one day freed) under the main one

        def walk(x):
            if isinstance(x, torch.Tensor):
                x.record_stream(main)
            elif isinstance(x, (list, tuple)):
                for y in x:
                    walk(y)
            elif isinstance(x, dict):
                for y in x.values():
                    walk(y)
            elif hasattr(x, "planes") and isinstance(getattr(x, "planes"), torch.Tensor):
                x.planes.record_stream(main)
        walk(new)
        self._prefetch_done, self._prefetch_dev = done, dev

    def _pyramid_buffers(self, sizes, dev, limit, protect=()):
        """Level buffers for a new correlation pyramid (5.6 GB at 1080p).  When the pyramid cache is at its
        limit the least recently used pyramid is retired FIRST and its buffers are handed to the new one:
        in the steady state of a sliding job no field allocates (a 5.6 GB hipMalloc costs up to 150 ms, and
        retiring only after the new allocation made the third field of every job pay for one)."""
        while sum(1 for k in self._feat_cache if k[0] == "p") >= limit:
            oldest = next((k for k in self._feat_cache if k[0] == "p" and k not in protect), None)
            if oldest is None:        # everything cached is being read on another stream: a fresh allocation it is
                break
            self._pyr_free.append(self._feat_cache.pop(oldest))
        for i, bufs in enumerate(self._pyr_free):
            if [b.numel() for b in bufs] == sizes and bufs[0].device == dev:
                return self._pyr_free.pop(i)
        del self._pyr_free[:]          # other geometry: let the allocator have the memory back
        return [torch.empty(n, device=dev) for n in sizes]

    def _cache_get(self, kind, key):
        ent = self._feat_cache.get((kind, key))
        if ent is not None:
            self._feat_cache.move_to_end((kind, key))
        return ent

    def _cache_put(self, kind, key, value, limit=None):
        self._feat_cache[(kind, key)] = value
        while sum(1 for k in self._feat_cache if k[0] == kind) > (limit or self.FEATURE_CACHE_FRAMES):
            oldest = next(k for k in self._feat_cache if k[0] == kind)
            del self._feat_cache[oldest]

    def _volume_geometry(self, H, W, L, AF):
        """Shapes of a window's correlation volumes: level sizes, the tile layout, element format, row strides, buffer sizes."""
        import types
        cfg = self.cfg
        h, w = H // 8, W // 8
        hl, wl = [h], [w]
        for l in range(1, L):
            hl.append(hl[-1] // 2)
            wl.append(wl[-1] // 2)
        Sl = [hl[l] * wl[l] for l in range(L)]
        # volume geometry: Nl columns per level, Pv rows - the pixels, or the whole tiles of a tiled volume (_tile)
        VT = self._tile()
        Nl = [VT.count(hl[l], wl[l]) for l in range(L)] if VT is not None else Sl
        Pv = Nl[0]
        TILE = VT.code if VT is not None else 0
        # cfg.corr_volume 'f16' / 'f16@k': the pyramids (from level k up) as one f16 per value, written by the GEMM form
        # (which needs every level's width a multiple of 4 and split-row query features) - other geometries keep f32
        # volumes.  The lookup reads as many texels of every level (a (2r+2)^2 window each), so levels 1-3 carry three
        # quarters of what f16 texels save it, at 13 % of the volume's bytes
        mask = self.CORR_VOLUMES[getattr(cfg, "corr_volume", "f32")] & ((1 << L) - 1)
        if not (mask and AF == hip.FMT_S16 and Pv % 4 == 0 and all(s % 4 == 0 for s in Nl) and L <= 4
                and cfg.corr_radius in (3, 4)):
            mask = 0
        v16 = [bool((mask >> l) & 1) for l in range(L)]
        VFl = [hip.FMT_F16 if v16[l] else hip.FMT_F32 for l in range(L)]
        VF = (hip.FMT_F32 if not mask else hip.FMT_F16 if mask == (1 << L) - 1 else hip.vol_f16_levels(mask))   # the lookups' vol_fmt
        # row stride of a level: whole 128-byte lines, an ODD number of them - the transposed second output of the
        # level-0 GEMM walks down a column, and at an even multiple (32640 floats = 255 x 512 bytes at 1080p) its
        # stores queue on half the memory channels: 1989 us per launch against 1652 (tools/exp/volume_gemm_shapes.py)
        ldl = []
        for l in range(L):
            unit = 64 if v16[l] else 32
            n = (Nl[l] + unit - 1) // unit * unit
            ldl.append(n if (n // unit) % 2 else n + unit)
        psz = [(Pv * ldl[l] + 1) // 2 if v16[l] else Pv * ldl[l] for l in range(L)]     # floats per level buffer
        return types.SimpleNamespace(L=L, AF=AF, hl=hl, wl=wl, Sl=Sl, VT=VT, Nl=Nl, Pv=Pv, TILE=TILE, vol16=mask, VF=VF,
                                     VFl=VFl, ldl=ldl, psz=psz)

    def _prefetch_pyramids(self):
        """VFML_PREFETCH_PYR=1: the next window's new correlation pyramids are built on the prefetch stream too.  Off: the
        resident workgroups of the persistent volume GEMM take a workgroup slot of every CU from the iterations'
        convolutions for as long as they run - 24.75 ms per field with it against 24.70 without (fields bit-identical)."""
        return os.environ.get("VFML_PREFETCH", "1") != "0" and os.environ.get("VFML_PREFETCH_PYR", "0") == "1" and not self.tri_frame

    def _window_pyramids(self, feats, keys, N, geo, dev, protect=()):
        """K3/K4: the correlation pyramids of a window, one per problem (query frame -> target frame): level l is one GEMM of
        the query frame's features against the 2^l-pooled features of the target frame.  A pyramid depends on its two frames
        only, so with frame keys it is kept across windows: consecutive sliding windows share 2(N-3) of their 2(N-2)
        problems.  Returns {"f": [...], "b": [...]} (per centre frame, a list of level buffers).
        protect: cache keys of pyramids another stream is reading - their buffers are not recycled for new ones."""
        D, L = self.cfg.feat_dim, geo.L
        AF, VFl, Nl, Pv, ldl, psz = geo.AF, geo.VFl, geo.Nl, geo.Pv, geo.ldl, geo.psz
        scale = 1.0 / float(D) ** 0.5
        if AF == hip.FMT_S16:
            scale /= self.FMAP_ROW_SCALE       # the split-row query features carry a factor 16
        # (two more than a window and its reverse twin need when the next window's are built beside this one's iterations)
        lim = 2 * (N - 2) + 2 + (2 if self._prefetch_pyramids() else 0)
        pyrs = {"f": [], "b": []}
        for c in range(1, N - 1):
            for d, tgt in (("f", c + 1), ("b", c - 1)):
                pk = ("p", keys[c], keys[tgt]) if keys is not None else None
                pyr = self._cache_get("p", pk) if pk is not None else None
                if pyr is None:
                    if pk is None:    # uncached call: reuse one workspace set per problem slot
                        pyr = [self._buf(f"pyr_{d}{c}_{l}", psz[l], dev) for l in range(L)]
                    else:
                        # (registered before it is filled: same stream, and the next allocation then sees
                        # the right count and retires a stale pyramid instead of asking the allocator)
                        pyr = self._pyramid_buffers(psz, dev, limit=lim, protect=protect)
                        self._cache_put("p", pk, pyr, limit=lim)
                    # Level 0 of the reverse problem (tgt -> c) is the transpose of this one's: when it is
                    # going to be needed (tgt is, or next field becomes, a centre frame: the "f" problems of
                    # a forward-sliding job) the same pass of MFMAs stores it too (vfml_conv_desc.out_t), and
                    # only its three pooled levels are separate GEMMs.  In the steady state a field then
                    # builds its two new pyramids with one 32400 x 32400 GEMM instead of two.
                    # (a backward problem's level 0 computed directly uses VFML_CONV_SWAP_CROSS, the addition
                    # order of a transposed forward volume: both routes give the same bits)
                    rk = ("p", keys[tgt], keys[c]) if pk is not None else None
                    gemm_form = AF == hip.FMT_S16 and Pv % 4 == 0 and Nl[0] >= 1024
                    dual = (rk is not None and d == "f" and gemm_form and ("p", rk) not in self._feat_cache
                            and not os.environ.get("VFML_NO_DUAL"))      # (A/B switch; results are identical)
                    rev = None
                    if dual:
                        rev = self._pyramid_buffers(psz, dev, limit=lim, protect=protect)
                        self._cache_put("p", rk, rev, limit=lim)
                    cnm = self._nm("corr") if self._split() else 3
                    for l in range(L):
                        # (one MFMA per product is symmetric in its operands: no swapped cross terms to order)
                        hip.conv2d(feats[c][0], D, D, 1, 1, Pv, feats[tgt][1][l], None, Nl[l], 1, 1, pyr[l],
                                   ldl[l], out_scale=scale, in_fmt=AF, out_fmt=VFl[l],
                                   swap_cross=(d == "b" and l == 0 and gemm_form and cnm == 3),
                                   out_t=rev[0] if dual and l == 0 else None, ld_out_t=ldl[0] if dual and l == 0 else 0,
                                   mfma=cnm)
                    if dual:
                        for l in range(1, L):
                            hip.conv2d(feats[tgt][0], D, D, 1, 1, Pv, feats[c][1][l], None, Nl[l], 1, 1, rev[l],
                                       ldl[l], out_scale=scale, in_fmt=AF, out_fmt=VFl[l], mfma=cnm)
                pyrs[d].append(pyr)
        return pyrs

    def _frame_features(self, src, sel, keys, H, W, P, dev, L, hl, wl, Sl, vt=None):
        """Feature map + target pyramid of frames `sel` of the window.
        Returns {j: (fmap [Pn*256] f32, [target operand per level])}.
        vt (hip.VolTile): both operands' rows in tile order (zero rows at the positions no pixel has), so that the volume
        GEMMs write tiled level images under tile-ordered rows."""
        D = self.cfg.feat_dim
        split = self._split()
        out, todo = {}, []
        for j in sel:
            ent = self._cache_get("f", keys[j]) if keys is not None else None
            if ent is not None:
                out[j] = ent
            else:
                todo.append(j)
        if todo:
            m, Pn = len(todo), hl[0] * wl[0]
            frames = self._buf("frames", m * H * W * 4, dev)
            hip.frames_to_nhwc4(take_frames(src, todo).contiguous(), m, H, W, float(self.cfg.input_scale), float(self.cfg.input_shift),
                                frames)
            fmap = torch.empty(m * Pn * D, device=dev)          # owned by the cache entries (views)
            self._encoder("fnet", frames, m, H, W, P, dev, fmap, D, 0, hip.EPI_NONE, 0)
            levels = [fmap]
            for l in range(1, L):
                t = torch.empty(m * Sl[l] * D, device=dev)
                hip.avgpool2x2(levels[-1], m, hl[l - 1], wl[l - 1], D, t)
                levels.append(t)
            for i, j in enumerate(todo):
                tg = []
                for l in range(L):
                    f = levels[l][i * Sl[l] * D:(i + 1) * Sl[l] * D]
                    nl = Sl[l]
                    if vt is not None:
                        f, nl = vt.rows(f, hl[l], wl[l], D), vt.count(hl[l], wl[l])
                    # features are O(1): x16 keeps the lo halves of the split normal
                    tg.append(hip.SplitWeight(nl, D, dev).fill(f, scale=16.0) if split else f)
                fm = fmap[i * Pn * D:(i + 1) * Pn * D]
                if vt is not None:
                    fm, Pn_rows = vt.rows(fm, hl[0], wl[0], D), vt.count(hl[0], wl[0])
                else:
                    Pn_rows = Pn
                if split:
                    # the GEMM's A operand in split rows, made once per frame - of the SAME x16 copy the target
                    # planes are split from, so that a frame's hi / lo halves are the same numbers on either
                    # side of a correlation GEMM (then <a, b> and <b, a> are the same products, and with
                    # VFML_CONV_SWAP_CROSS the same sums: a volume and its transpose are bit-identical)
                    fm16 = torch.empty(Pn_rows * D, device=dev)
                    hip.to_s16(fm, Pn_rows, D, D, fm16, D, scale=self.FMAP_ROW_SCALE)
                    fm = fm16
                ent = (fm, tg)
                out[j] = ent
                if keys is not None:
                    self._cache_put("f", keys[j], ent)
        return out

    def _frame_context(self, src, sel, keys, H, W, P, dev, Pn, M=1):
        """Context maps (tanh | relu halves, [Pn*256] f32) of frames `sel`, and the context part of their gate convolutions:
        out[j] = (map, {gate: view of the frame's slot in the gate's store}, slot)."""
        out, todo = {}, []
        st = self._context_store(dev, Pn, max(M, len(sel)))
        for j in sel:
            ent = self._cache_get("c", keys[j]) if keys is not None else None
            if ent is not None and ent[3] == st["serial"]:
                out[j] = ent
            else:
                todo.append(j)
        if todo:
            m = len(todo)
            frames = self._buf("frames", m * H * W * 4, dev)
            hip.frames_to_nhwc4(take_frames(src, todo).contiguous(), m, H, W,
                                float(self.cfg.input_scale), float(self.cfg.input_shift), frames)
            ctx = torch.empty(m * Pn * 256, device=dev)
            AF = hip.FMT_S16 if self._split() else hip.FMT_F32
            self._encoder("cnet", frames, m, H, W, P, dev, ctx, 256, 0, hip.EPI_TANH_RELU, self.hidden_dim, out_fmt=AF)
            h8, w8 = H // 8, W // 8
            R, Mst = st["R"], st["M"]
            for i, j in enumerate(todo):
                cx = ctx[i * Pn * 256:(i + 1) * Pn * 256]
                busy = {e[2] for e in out.values()}
                slot = self._context_slot(st, ("c", keys[j]) if keys is not None else None, busy)
                # context part of the GRU gate convolutions (+ bias), per pass: [z|r] (256) and q (128), into the frame's slot
                add = {}
                for k, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
                    for g, co in (("zr", 256), ("q", 128)):
                        wgt, b = P[f"update_block.gru.conv{g}{k}.ctx"]
                        S = st["S"][g + k]
                        hip.conv2d(cx, 128, 256, 1, h8, w8, wgt, b, co, kh, kw, S, co, in0_off=128, out_off=slot * Pn * co,
                                   pad_h=kh // 2, pad_w=kw // 2, in_fmt=AF,
                                   mfma=self._nm(f"update_block.gru.conv{g}{k}.ctx") if self._split() else 3)
                        add[g + k] = S[slot * Pn * co:(slot + 1) * Pn * co]
                        if slot < Mst - 1:          # the ring's first slots again behind it: a window that wraps stays consecutive
                            S[(R + slot) * Pn * co:(R + slot + 1) * Pn * co].copy_(add[g + k])
                out[j] = (cx, add, slot, st["serial"])
                if keys is not None:
                    self._cache_put("c", keys[j], out[j])
        return out

    def _run(self, src, N, H, W, return_lowres, frame_keys=None, tri_batch=False, pick_only=False):
        cfg = self.cfg
        if N < 3:
            raise ValueError(f"need at least 3 frames, got {N}")
        if self.tri_frame and N > 3 and not tri_batch:
            lo = N // 2 - 1
            src = src[lo:lo + 3].contiguous()
            frame_keys = frame_keys[lo:lo + 3] if frame_keys is not None else None
            N = 3
        if H % 8 or W % 8:
            raise ValueError("H and W must be multiples of 8 (use InputPadder)")
        L, R, D = cfg.corr_levels, cfg.corr_radius, cfg.feat_dim
        h, w = H // 8, W // 8
        if (h >> (L - 1)) < 2 or (w >> (L - 1)) < 2:
            raise ValueError(f"frame {H}x{W} too small for a {L}-level correlation pyramid")
        dev = src.device
        self._join_prefetch()      # a prefetch shares the encoders' workspaces and fills the caches read below
        M = N - 2
        Pn = h * w          # cells per map
        MP = M * Pn
        P = self._pack(dev)
        win = (2 * R + 1) ** 2
        cor = L * win
        AF = hip.FMT_S16 if self._split() else hip.FMT_F32   # update-block activations
        cor_p = cor_pad(cor, AF == hip.FMT_S16)   # per-direction channel block

        with torch.cuda.device(dev):
            geo = self._volume_geometry(H, W, L, AF)
            hl, wl, Sl, VT, TILE, vol16, VF, ldl = geo.hl, geo.wl, geo.Sl, geo.VT, geo.TILE, geo.vol16, geo.VF, geo.ldl
            keys = None
            if frame_keys is not None:   # geometry and arithmetic are part of a cached frame's identity
                keys = [(k, H, W, L, self._plan_key(), self._packed_serial) for k in frame_keys]

            # K1 + K2: feature maps and pooled target pyramids, per frame (cached across windows)
            feats = self._frame_features(src, list(range(N)), keys, H, W, P, dev, L, hl, wl, Sl, vt=VT)

            # K3/K4 correlation pyramids, one per problem (query frame -> target frame)
            pyrs = self._window_pyramids(feats, keys, N, geo, dev)
            self._pyr_busy = ({("p", keys[c], keys[t]) for c in range(1, N - 1) for t in (c + 1, c - 1)}
                              if keys is not None else set())

            # Recurrent state, one row of GLD floats per cell:  [ z | r*h | h | inp | mf | mt ]
            # (one allocation, so that cat([r*h, x]) and cat([h, x]) are channel slices of it)
            GLD, Z, RH, HH, INP, MF, MT = 768, 0, 128, 256, 384, 512, 640
            G = self._buf("gru_state", MP * GLD, dev)
            # K2 context encoder on the centre frames -> h = tanh(first half), inp = relu(second half)
            ctx = self._frame_context(src, list(range(1, N - 1)), keys, H, W, P, dev, Pn, M)
            Gv = G.view(MP, GLD)
            for c in range(1, N - 1):       # h = tanh half of the context map (the relu half reaches the gates as their addends)
                Gv[(c - 1) * Pn:c * Pn, HH:HH + 128].copy_(ctx[c][0].view(Pn, 256)[:, :128])
            # the gates' context parts: the centre frames' slots, consecutive in a sliding job (through the mirror when the ring
            # wraps) - else gathered into the store's spare slots
            st = self._ctx_store
            RING, Mst = st["R"], st["M"]
            slots = [ctx[c][2] for c in range(1, N - 1)]
            st["live"] = set(slots)
            first = slots[0]
            # (the exact-f32 kernel takes the pointer itself - a captured launch needs it fixed: always the gathered copy)
            # (A/B switch: VFML_CTX_GATHER=1 gathers every window, as round 2 did)
            if (not self._split() or os.environ.get("VFML_CTX_GATHER", "0") == "1" or
                    not all(s == first + i or (first + i >= RING and s == first + i - RING and s < Mst - 1)
                            for i, s in enumerate(slots))):
                first = RING + Mst - 1
                for i, c in enumerate(range(1, N - 1)):
                    for name, co in self.CTX_GATES:
                        st["S"][name][(first + i) * Pn * co:(first + i + 1) * Pn * co].copy_(ctx[c][1][name])
            gate_add = {name: st["S"][name] for name, _ in self.CTX_GATES}
            gate_off = {name: first * Pn * co for name, co in self.CTX_GATES}
            gate_ind = None
            if self._split():      # (the exact-f32 kernel takes the pointer itself)
                cells = self._buf("gate_add_cells", 8, dev, torch.int64)
                hip.ptr_table_set(cells, [st["S"][name][gate_off[name]:] for name, _ in self.CTX_GATES])
                gate_ind = {name: (cells, i) for i, (name, _) in enumerate(self.CTX_GATES)}

            corr = self._buf("corr", MP * 2 * cor_p, dev, zero=True)   # pad channels stay zero
            c1 = self._buf("c1", MP * 256, dev)
            cf = self._buf("cf", MP * 256, dev)
            f1 = self._buf("f1", MP * 128, dev)
            fh = self._buf("fh", MP * 256, dev)
            flow4 = self._buf("flow4", MP * 4, dev)
            delta = self._buf("delta", MP * 4, dev)
            fh_taps = self._buf("fh_taps", 2 * MP * 36, dev)     # (two partial maps with the fused flow head)
            frows = self._buf("flow_rows7", MP * 32, dev)
            coords1 = self._buf("coords1", MP * 4, dev)

            # ---- the update iterations + mask head + upsampling: a FIXED launch sequence for a given geometry,
            # window length and arithmetic - only the correlation pyramids it looks up differ from field to field,
            # and those are named by two device tables.  With cfg.use_graph the sequence is captured into a HIP
            # graph the second time a configuration is seen and replayed from then on (~250 launches per field
            # become one; same kernels, same order: bit-identical results).
            # both directions behind one another in one table: the two lookups of an iteration as ONE launch (at most 8 maps;
            # A/B switch VFML_LOOKUP_BIDIR=0)
            bidir = 2 * M <= 8 and 2 * M * L <= 48 and os.environ.get("VFML_LOOKUP_BIDIR", "1") != "0"
            if bidir:
                tab_fb = self._buf("pyr_table_fb", 64, dev, torch.int64)
                hip.ptr_table_set(tab_fb, [p for d in ("f", "b") for m in pyrs[d] for p in m])
            else:
                tab_f = self._buf("pyr_table_f", 64, dev, torch.int64)
                tab_b = self._buf("pyr_table_b", 64, dev, torch.int64)
                hip.ptr_table_set(tab_f, [p for m in pyrs["f"] for p in m])
                hip.ptr_table_set(tab_b, [p for m in pyrs["b"] for p in m])
            no = 1 if pick_only and not self.tri_frame else M
            nflows = 1 if pick_only and not self.tri_frame else 2 * M
            up_fixed = self._buf("up_out", nflows * H * W * 2, dev)
            mask = self._buf("mask", MP * 1152, dev)

            def body():
                hip.coords_init(coords1, M, h, w)
                hip.coords_update(coords1, None, M, h, w, flow_a=flow4, ld_a=4, flow_b=G, ld_b=GLD, flow_b_off=MF + 124,
                                  fmt_b=AF)
                ub = "update_block"
                mf = self._nm if self._split() else (lambda layer: 3)     # MFMAs per product of a layer (cfg.precision)
                branch = None
                if os.environ.get("VFML_FLOW_BRANCH", "0") == "1" and self._split():
                    branch = self._side2.get(dev)
                    if branch is None:
                        branch = self._side2[dev] = torch.cuda.Stream(device=dev)
                # (A/B switch: VFML_FUSE_HEAD=0 runs the flow head's two layers as two launches)
                fuse_head = os.environ.get("VFML_FUSE_HEAD", "1") != "0" and AF == hip.FMT_S16
                fuse_flow = os.environ.get("VFML_FUSE_FLOW", "1") != "0" and AF == hip.FMT_S16      # (A/B switch, as above)
                for it in range(cfg.decoder_depth):
                    # pick_only: the caller takes flow M (the backward flow of the first centre frame - the reference's
                    # `[0, shape[1]//2]`).  Going back from the last iteration, centre 0's result depends on one centre
                    # more per iteration (through the temporal fusion), so the last iterations run on the centres
                    # that still matter only: GRU + flow head on `ng`, lookups + motion encoder on `nm` of them.
                    left = cfg.decoder_depth - 1 - it
                    ng = min(M, left + 1) if pick_only and not self.tri_frame else M
                    nm = min(M, left + 2) if pick_only and not self.tri_frame else M
                    # The flow half of the motion encoder (flow -> convf1 -> convf2 -> channels 192..255 of `cf`) needs
                    # nothing of the correlation half (lookups -> convc1 -> convc2 -> channels 0..191): on a second stream
                    # its small MFMA-bound convolutions run BESIDE the HBM-bound lookups (a fork / join of two events; inside
                    # a captured graph, two branches).  Same kernels on the same inputs: bit-identical fields.
                    def flow_half(nm=nm):
                        wgt, b = P[f"{ub}.encoder.convf1"]
                        wgt2, b2 = P[f"{ub}.encoder.convf2"]
                        if (fuse_flow and self._rows7 and mf(f"{ub}.encoder.convf1") == 1 and mf(f"{ub}.encoder.convf2") == 1
                                and getattr(wgt2, "order", None) == hip.KORDER_CBLOCK64 and wgt.kp == 224 and wgt2.kp == 1152):
                            # both layers in one launch, the 128-channel map between them in LDS (vfml_flow_half; same bits)
                            hip.flow_half(flow4, nm, h, w, wgt, b, wgt2, b2, cf, 256, out_off=192)
                            return
                        if self._rows7:
                            hip.flow_rows7(flow4, nm, h, w, frows)
                            hip.conv2d(frows, 32, 32, nm, h, w, wgt, b, 128, 7, 1, f1, 128, pad_h=3, epilogue=hip.EPI_RELU,
                                       in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.encoder.convf1"))
                        else:
                            hip.conv2d(flow4, 4, 4, nm, h, w, wgt, b, 128, 7, 7, f1, 128, pad_h=3, pad_w=3, epilogue=hip.EPI_RELU,
                                       out_fmt=AF, mfma=mf(f"{ub}.encoder.convf1"))
                        wgt, b = P[f"{ub}.encoder.convf2"]
                        hip.conv2d(f1, 128, 128, nm, h, w, wgt, b, 64, 3, 3, cf, 256, out_off=192, pad_h=1, pad_w=1,
                                   epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.encoder.convf2"))

                    join = None
                    if branch is not None:
                        fork = torch.cuda.Event()
                        fork.record()                      # (flow4 of the previous iteration is complete on this stream)
                        with torch.cuda.stream(branch):
                            branch.wait_event(fork)
                            flow_half()
                            join = torch.cuda.Event()
                            join.record()
                    # K5
                    if bidir:
                        hip.corr_lookup(None, hl, wl, ldl, R, Pn, coords1, 0, 4, corr, 0, 2 * cor_p, out_fmt=AF, table=tab_fb,
                                        nmaps=nm, vol_fmt=VF, vol_tile=TILE, bidir=(2, cor_p, M))
                    else:
                        hip.corr_lookup(None, hl, wl, ldl, R, Pn, coords1, 0, 4, corr, 0, 2 * cor_p, out_fmt=AF,
                                        table=tab_f, nmaps=nm, vol_fmt=VF, vol_tile=TILE)
                        hip.corr_lookup(None, hl, wl, ldl, R, Pn, coords1, 2, 4, corr, cor_p, 2 * cor_p, out_fmt=AF,
                                        table=tab_b, nmaps=nm, vol_fmt=VF, vol_tile=TILE)
                    # motion encoder
                    wgt, b = P[f"{ub}.encoder.convc1"]
                    hip.conv2d(corr, 2 * cor_p, 2 * cor_p, nm, h, w, wgt, b, 256, 1, 1, c1, 256, epilogue=hip.EPI_RELU,
                               in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.encoder.convc1"))
                    wgt, b = P[f"{ub}.encoder.convc2"]
                    hip.conv2d(c1, 256, 256, nm, h, w, wgt, b, 192, 3, 3, cf, 256, pad_h=1, pad_w=1, epilogue=hip.EPI_RELU,
                               in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.encoder.convc2"))
                    if join is None:
                        flow_half()
                    else:
                        torch.cuda.current_stream(dev).wait_event(join)
                    wgt, b = P[f"{ub}.encoder.conv"]
                    hip.conv2d(cf, 256, 256, nm, h, w, wgt, b, 124, 3, 3, G, GLD, out_off=MF, pad_h=1, pad_w=1,
                               epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.encoder.conv"))
                    # temporal stack fusion: 3x1 conv along the frame axis of the motion features
                    wgt, b = P[f"{ub}.tprop"]
                    if self.tri_frame:     # every centre frame is its own problem: its neighbours are the zero padding
                        hip.conv2d(G, 128, GLD, M, 1, Pn, wgt, b, 128, 3, 1, G, GLD, in0_off=MF, out_off=MT, pad_h=1,
                                   epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.tprop"))
                    else:
                        # (on the first nm frames: row nm-1 of a shortened stack sees zero padding where its lower
                        # neighbour was - that row is not among the ng < nm the GRU reads)
                        hip.conv2d(G, 128, GLD, 1, nm, Pn, wgt, b, 128, 3, 1, G, GLD, in0_off=MF, out_off=MT, pad_h=1,
                                   epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.tprop"))
                    # SepConvGRU, horizontal then vertical
                    for k, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
                        wgt, _ = P[f"{ub}.gru.convzr{k}.iter"]
                        # [z | r*h] = gates(conv([h | motion | temporal]) + context part)
                        hip.conv2d(G, 128, GLD, ng, h, w, wgt, None, 256, kh, kw, G, GLD, in0_off=HH, out_off=Z,
                                   in1=G, c1=256, ld1=GLD, in1_off=MF, pad_h=kh // 2, pad_w=kw // 2,
                                   epilogue=hip.EPI_GRU_ZR, split=128, aux0=G, ld_aux0=GLD, aux0_off=HH,
                                   addend=gate_add["zr" + k], ld_addend=256, in_fmt=AF, out_fmt=AF, aux_fmt=AF,
                                   addend_off=0 if gate_ind else gate_off["zr" + k],
                                   addend_ind=gate_ind["zr" + k] if gate_ind else None,
                                   mfma=mf(f"{ub}.gru.convzr{k}.iter"))
                        wgt, _ = P[f"{ub}.gru.convq{k}.iter"]
                        # h = (1 - z) h + z tanh(conv([r*h | motion | temporal]) + context part), in place
                        hip.conv2d(G, 128, GLD, ng, h, w, wgt, None, 128, kh, kw, G, GLD, in0_off=RH, out_off=HH,
                                   in1=G, c1=256, ld1=GLD, in1_off=MF, pad_h=kh // 2, pad_w=kw // 2,
                                   epilogue=hip.EPI_GRU_Q, aux0=G, ld_aux0=GLD, aux0_off=Z,
                                   aux1=G, ld_aux1=GLD, aux1_off=HH, addend=gate_add["q" + k], ld_addend=128,
                                   addend_off=0 if gate_ind else gate_off["q" + k],
                                   addend_ind=gate_ind["q" + k] if gate_ind else None,
                                   in_fmt=AF, out_fmt=AF, aux_fmt=AF, mfma=mf(f"{ub}.gru.convq{k}.iter"))
                    # flow head
                    wgt, b = P[f"{ub}.flow_head.conv1"]
                    wgt2, b2 = P[f"{ub}.flow_head.conv2"]
                    nm_head = mf(f"{ub}.flow_head.conv1")
                    if self._tapsum and fuse_head and nm_head in (3, "2a") and mf(f"{ub}.flow_head.conv2") == nm_head:
                        # both layers in ONE launch (vfml_conv_desc.proj_out): the 256-channel map stays in LDS, the launch
                        # leaves two partial 36-column maps (one per 128-channel half) that the tap sum adds
                        hip.conv2d(G, 128, GLD, ng, h, w, wgt, b, 256, 3, 3, fh, 256, in0_off=HH, pad_h=1, pad_w=1,
                                   epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF, mfma=nm_head, proj=wgt2, proj_out=fh_taps,
                                   ld_proj=36)
                        hip.tapsum3x3_update(fh_taps, 36, b2, ng, h, w, coords1, parts=2, part_stride=ng * Pn * 36, flow_a=flow4,
                                             ld_a=4, flow_b=G, ld_b=GLD, flow_b_off=MF + 124, fmt_b=AF)
                        continue
                    hip.conv2d(G, 128, GLD, ng, h, w, wgt, b, 256, 3, 3, fh, 256, in0_off=HH, pad_h=1, pad_w=1,
                               epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.flow_head.conv1"))
                    wgt, b = P[f"{ub}.flow_head.conv2"]
                    if self._tapsum:
                        # 256 -> 4 over 3x3 as a 1x1 to 36 tap-major columns + the nine taps summed per pixel (_pack)
                        hip.conv2d(fh, 256, 256, ng, h, w, wgt, None, 36, 1, 1, fh_taps, 36, in_fmt=AF,
                                   mfma=mf(f"{ub}.flow_head.conv2"))
                        hip.tapsum3x3_update(fh_taps, 36, b, ng, h, w, coords1, flow_a=flow4, ld_a=4, flow_b=G, ld_b=GLD,
                                             flow_b_off=MF + 124, fmt_b=AF)
                        continue
                    else:
                        hip.conv2d(fh, 256, 256, ng, h, w, wgt, b, 4, 3, 3, delta, 4, pad_h=1, pad_w=1, in_fmt=AF,
                                   mfma=mf(f"{ub}.flow_head.conv2"))
                    hip.coords_update(coords1, delta, ng, h, w, flow_a=flow4, ld_a=4, flow_b=G, ld_b=GLD,
                                      flow_b_off=MF + 124, fmt_b=AF)

                # mask head on the final hidden state, then K8 for every flow of the output tensor (pick_only: of
                # the first centre frame, and its backward flow alone)
                wgt, b = P[f"{ub}.mask.0"]
                hip.conv2d(G, 128, GLD, no, h, w, wgt, b, 256, 3, 3, fh, 256, in0_off=HH, pad_h=1, pad_w=1,
                           epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF, mfma=mf(f"{ub}.mask.0"))
                wgt, b = P[f"{ub}.mask.2"]
                hip.conv2d(fh, 256, 256, no, h, w, wgt, b, 1152, 1, 1, mask, 1152, out_scale=0.25, in_fmt=AF,
                           mfma=mf(f"{ub}.mask.2"))
                if pick_only and not self.tri_frame:
                    hip.convex_upsample(coords1, 0, 2, mask, 576, 1152, h, w, up_fixed)
                else:
                    for d in range(2):
                        for c in range(M):
                            hip.convex_upsample(coords1, c * Pn * 4, 2 * d, mask, c * Pn * 1152 + d * 576, 1152, h, w,
                                                up_fixed, out_off=(d * M + c) * H * W * 2)

            gkey = (H, W, N, M, bool(tri_batch), bool(pick_only), cfg.decoder_depth, L, R, self._plan_key(), vol16,
                    self._packed_serial, str(dev), os.environ.get("VFML_FLOW_BRANCH", "0"), os.environ.get("VFML_FUSE_HEAD", "1"), bidir, os.environ.get("VFML_FUSE_FLOW", "1"))
            self._pre_body = torch.cuda.Event()
            self._pre_body.record(torch.cuda.current_stream(dev))      # (what a prefetch of the next window waits for)
            self._run_body(body, gkey, dev)
            up = up_fixed.clone().view(nflows, H, W, 2)          # the caller owns its field; the fixed buffer is reused
            if pick_only and not self.tri_frame:
                return up.permute(0, 3, 1, 2).unsqueeze(0), None        # [1, 1, 2, H, W]: flow M of the full output
            # [2M,H,W,2] stored HWC; expose the reference's [B, 2M, 2, H, W] as a view
            flow = up.permute(0, 3, 1, 2).unsqueeze(0)
            low = None
            if return_lowres:
                low = flow4.view(M, h, w, 4)[..., :].clone()
        return flow, low


def build_network(cfg):
    return MOFNetHIP(cfg)
