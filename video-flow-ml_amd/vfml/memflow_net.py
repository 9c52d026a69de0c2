"""MemFlowNetHIP — the MemFlow pair network on the vfml HIP kernels.

Stands in for what reference processing/memflow_inference_isolated.py:54-107 builds out of the
(absent) MemFlow submodule: `build_network(cfg)` + `InferenceCore(model).step(pair, end=True)` on the
last two frames of the window, with an empty memory bank.  Architecture: DESIGN.md §2b / oracle/
memflow_oracle.py — RAFT encoders and one correlation pyramid (previous -> current frame), an update
block whose motion features get a memory read-out added,  m_global = m + gamma * softmax(q k^T/sqrt(d)) v,
where (with the empty bank) keys and values are the frame's own.

MI355X shape of the read-out: q and k come from the context features and do not change over the
iterations, so the P x P attention matrix is computed ONCE per field, kept resident in HBM as
split-row halves (4.2 GB at 1080p; no flash-style re-computation needed with 288 GB), and every
iteration is one long-K GEMM  attn[P,P] . v[P,128]  on the MFMA kernel with the residual add fused.
"""
import os

import torch
import torch.nn as nn

from . import hip
from .network import MOFNetHIP, _Holder
from .weights import pack_conv_weight

# attention probabilities are stored times 2^14 (split rows hold f16 halves: an unscaled 1080p row, 32400 probabilities
# of 3e-5 on average, would sit in the f16 subnormals); the read-out GEMM divides it out through out_scale
ATT_SCALE = float(os.environ.get("VFML_ATT_SCALE", "16384"))      # (override: precision experiments)
ATT_PLAIN = os.environ.get("VFML_ATT_SPLIT_ROWS", "0") != "1"       # 1: keep the probabilities as split rows (hi + lo)


def memflow_conv_spec(cfg):
    from .weights import _encoder_spec
    cor = cfg.corr_levels * (2 * cfg.corr_radius + 1) ** 2
    hid, ad = cfg.feat_dim // 2, cfg.att_dim
    ub = "update_block"
    s = _encoder_spec("fnet", cfg.feat_dim) + _encoder_spec("cnet", cfg.feat_dim)
    s += [("query", ad, hid, 1, 1), ("key", ad, hid, 1, 1),
          (f"{ub}.encoder.convc1", 256, cor, 1, 1), (f"{ub}.encoder.convc2", 192, 256, 3, 3),
          (f"{ub}.encoder.convf1", 128, 2, 7, 7), (f"{ub}.encoder.convf2", 64, 128, 3, 3),
          (f"{ub}.encoder.conv", 128 - 4, 192 + 64, 3, 3), (f"{ub}.value", ad, 128, 1, 1)]
    for nm, kh, kw in (("z1", 1, 5), ("r1", 1, 5), ("q1", 1, 5), ("z2", 5, 1), ("r2", 5, 1), ("q2", 5, 1)):
        s.append((f"{ub}.gru.conv{nm}", hid, hid + 3 * 128, kh, kw))
    s += [(f"{ub}.flow_head.conv1", 256, hid, 3, 3), (f"{ub}.flow_head.conv2", 2, 256, 3, 3),
          (f"{ub}.mask.0", 256, hid, 3, 3), (f"{ub}.mask.2", 64 * 9, 256, 1, 1)]
    return s


def memflow_cfg():
    from .cfg import Cfg
    return Cfg(restore_ckpt="", network="MemFlowNet", feat_dim=256, down_ratio=8, corr_levels=4, corr_radius=4,
               decoder_depth=12, att_dim=128, precision="f16x3",
               input_scale=1.0, input_shift=0.0)   # frames reach the network already in [-1, 1]


def seeded_memflow_state_dict(cfg, seed=0, gamma=0.5):
    import math
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, cout, cin, kh, kw in memflow_conv_spec(cfg):
        bound = 1.0 / math.sqrt(cin * kh * kw)
        sd[f"{name}.weight"] = (torch.rand(cout, cin, kh, kw, generator=g) * 2 - 1) * bound
        sd[f"{name}.bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound
    sd["update_block.gamma"] = torch.tensor([gamma])      # upstream initialises 0; non-zero exercises the read-out
    return sd


class MemFlowNetHIP(MOFNetHIP):
    def __init__(self, cfg):
        _Holder.__init__(self)
        import collections
        self.cfg = cfg
        self.hidden_dim = self.context_dim = cfg.feat_dim // 2
        self.tri_frame = False
        self._spec = memflow_conv_spec(cfg)
        for name, cout, cin, kh, kw in self._spec:
            leaf = self
            for p in name.split("."):
                leaf = leaf.child(p)
            leaf.weight = nn.Parameter(torch.zeros(cout, cin, kh, kw), requires_grad=False)
            leaf.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
        self.child("update_block").gamma = nn.Parameter(torch.zeros(1), requires_grad=False)
        self._packed = None
        self._packed_key = None
        self._packed_serial = 0
        self._ws = {}
        self._graphs = {}
        self._pyr_free = []
        self._att_planes = {}            # attention matrices as plain f16 planes (hip.PlainWeight), per pair of a pass
        self._feat_cache = collections.OrderedDict()

    def release_workspace(self):
        super().release_workspace()
        self._att_planes.clear()

    # ------------------------------------------------------------------ weights
    def _pack(self, device):
        split = self._precision() == "f16x3"
        key = (str(device), split, tuple(p._version for p in self.parameters()),
               tuple(p.data_ptr() for p in self.parameters()))
        if self._packed is not None and self._packed_key == key:
            return self._packed
        P, hid, cblock_names = {}, self.hidden_dim, set()
        ub = "update_block"
        for name, cout, cin, kh, kw in self._spec:
            leaf = self._param(name)
            w = leaf.weight.detach().to(device=device, dtype=torch.float32)
            b = leaf.bias.detach().to(device=device, dtype=torch.float32).contiguous()
            if name.endswith(".encoder.convc1"):       # lookup block padded to whole units (zero weights)
                cor_p = (cin + 31) // 32 * 32 if split else (cin + 3) // 4 * 4   # whole K steps: the uniform-step loader
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cor_p - cin))
            # update-block convolutions read split-row activations (all but convf1): channel-block K order
            cb = split and ((name.startswith(ub + ".") and not name.endswith(".convf1")) or
                            (self._enc_split_rows() and (
                                (name.split(".")[0] in ("fnet", "cnet") and name.count(".") > 1) or
                                name in ("fnet.conv2", "cnet.conv2"))))
            if cb:
                cblock_names.add(name)
            P[name] = (pack_conv_weight(w, cin_pad=4 if cin in (2, 3) else None, cblock=cb), b)
        self._cout_of = {}
        for k in ("1", "2"):
            raw = {g: self._param(f"{ub}.gru.conv{g}{k}") for g in "zrq"}
            wzr = torch.cat([raw["z"].weight, raw["r"].weight]).detach().to(device=device, dtype=torch.float32)
            bzr = torch.cat([raw["z"].bias, raw["r"].bias]).detach().to(device=device, dtype=torch.float32)
            wq = raw["q"].weight.detach().to(device=device, dtype=torch.float32)
            bq = raw["q"].bias.detach().to(device=device, dtype=torch.float32)
            for nm, wfull, bfull, co in ((f"{ub}.gru.convzr{k}", wzr, bzr, 2 * hid), (f"{ub}.gru.convq{k}", wq, bq, hid)):
                it = torch.cat([wfull[:, :hid], wfull[:, 2 * hid:]], dim=1)
                P[nm + ".iter"] = (pack_conv_weight(it, cblock=split), None)
                P[nm + ".ctx"] = (pack_conv_weight(wfull[:, hid:2 * hid], cblock=split), bfull.contiguous())
                if split:
                    cblock_names.update((nm + ".iter", nm + ".ctx"))
                self._cout_of[nm + ".iter"] = co
            for g in "zrq":
                del P[f"{ub}.gru.conv{g}{k}"]
        if split:
            with torch.cuda.device(device):
                for name, (wflat, b) in list(P.items()):
                    cout = self._cout_of[name] if name in self._cout_of else b.numel()
                    sc = hip.SplitWeight.auto_scale(float(wflat.abs().max()))
                    sw = hip.SplitWeight(cout, wflat.numel() // cout, device).fill(wflat, scale=sc)
                    sw.order = hip.KORDER_CBLOCK if name in cblock_names else hip.KORDER_TAP
                    P[name] = (sw, b)
        self._gamma = float(self._param(ub).gamma.item())   # read once per load: .item() synchronises
        self._packed, self._packed_key = P, key
        self._packed_serial += 1
        self._feat_cache.clear()
        return P

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, pair, data=None, frame_keys=None):
        """pair: float [1, 2, 3, H, W] in [-1, 1] on the GPU (previous, current).
        Returns (flow_low [1,2,h,w], flow [1,2,H,W]) like InferenceCore.step(..., end=True).
        frame_keys: optional pair of hashable ids of the two frames (same id = same normalised pixels): the
        feature encoder output of a frame is then kept, so a frame that is "current" in one call and
        "previous" in the next is encoded once (results bit-identical)."""
        if isinstance(pair, torch.Tensor) and pair.dim() == 5 and pair.shape[1] != 2:
            raise ValueError(f"pair must be [1,2,3,H,W], got {tuple(pair.shape)}")
        return self.forward_pairs(pair, data, frame_keys)

    @torch.no_grad()
    def forward_pairs(self, frames, data=None, frame_keys=None):
        """frames: float [1, B+1, 3, H, W] in [-1, 1] on the GPU: B overlapping pairs (k, k+1) - B consecutive
        fields of a job in one pass (every pair its own fresh-memory step, exactly as B separate calls; at
        1080p one pair leaves most tiles of the update-block convolutions' last round empty).
        Returns (flow_low [B,2,h,w], flow [B,2,H,W])."""
        if not isinstance(frames, torch.Tensor) or not frames.is_cuda:
            raise RuntimeError("MemFlowNetHIP runs on an MI355X (HIP) device only; got "
                               f"{getattr(frames, 'device', type(frames))}. There is no CPU fallback in the shipped engine.")
        if frames.dim() != 5 or frames.shape[0] != 1 or frames.shape[1] < 2 or frames.shape[2] != 3:
            raise ValueError(f"frames must be [1,B+1,3,H,W], got {tuple(frames.shape)}")
        cfg = self.cfg
        src = frames[0].float().contiguous()
        B = src.shape[0] - 1
        H, W = src.shape[2], src.shape[3]
        if H % 8 or W % 8:
            raise ValueError("H and W must be multiples of 8 (use InputPadder)")
        L, R, D, AD = cfg.corr_levels, cfg.corr_radius, cfg.feat_dim, cfg.att_dim
        h, w = H // 8, W // 8
        if (h >> (L - 1)) < 2 or (w >> (L - 1)) < 2:
            raise ValueError(f"frame {H}x{W} too small for a {L}-level correlation pyramid")
        dev = src.device
        Pn = h * w
        MP = B * Pn
        P = self._pack(dev)
        if self._precision() != "f16x3":
            raise ValueError("the MemFlow path is built on the split-f16 kernels: cfg.precision must be 'f16x3'")
        split, AF = True, hip.FMT_S16
        cor = L * (2 * R + 1) ** 2
        cor_p = (cor + 31) // 32 * 32 if split else (cor + 3) // 4 * 4
        ub = "update_block"
        gamma = self._gamma

        with torch.cuda.device(dev):
            hl, wl = [h], [w]
            for l in range(1, L):
                hl.append(hl[-1] // 2)
                wl.append(wl[-1] // 2)
            Sl = [hl[l] * wl[l] for l in range(L)]
            # volumes in tiles under tile-ordered rows, row strides an odd number of 128-byte lines (network.py _tile / _run)
            VT = self._tile()
            Nl = [VT.count(hl[l], wl[l]) for l in range(L)] if VT is not None else Sl
            Pv, TILE = Nl[0], (VT.code if VT is not None else 0)
            ldl = [(s + 31) // 32 * 32 for s in Nl]
            ldl = [n if (n // 32) % 2 else n + 32 for n in ldl]
            keys = None
            if frame_keys is not None:
                if len(frame_keys) != B + 1:
                    raise ValueError("frame_keys must hold one id per frame")
                keys = [("memflow", k, H, W, L, self._precision(), TILE, self._packed_serial) for k in frame_keys]
            feats = self._frame_features(src, list(range(B + 1)), keys, H, W, P, dev, L, hl, wl, Sl, vt=VT)
            ctx = self._frame_context_plain(src, B, H, W, P, dev, Pn, AF)       # cnet on the B previous frames
            pyrs = []
            for k in range(B):       # pair k: queries = frame k, targets = frame k+1
                pyr = [self._buf(f"mpyr{k}_{l}", Pv * ldl[l], dev) for l in range(L)]
                for l in range(L):
                    hip.conv2d(feats[k][0], D, D, 1, 1, Pv, feats[k + 1][1][l], None, Nl[l], 1, 1, pyr[l], ldl[l],
                               out_scale=1.0 / float(D) ** 0.5 / self.FMAP_ROW_SCALE, in_fmt=AF)   # query rows carry x16
                pyrs.append(pyr)

            GLD, Z, RH, HH, INP, MF, MT = 768, 0, 128, 256, 384, 512, 640
            G = self._buf("gru_state", MP * GLD, dev)
            G.view(MP, GLD)[:, HH:HH + 256].copy_(ctx.view(MP, 256))
            # context parts of the GRU gates (+ bias), once per field
            gate_add = {}
            for k, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
                for g, co in (("zr", 256), ("q", 128)):
                    wgt, b = P[f"{ub}.gru.conv{g}{k}.ctx"]
                    a = self._buf(f"gate_add_{g}{k}", MP * co, dev)
                    hip.conv2d(G, 128, GLD, B, h, w, wgt, b, co, kh, kw, a, co, in0_off=INP, pad_h=kh // 2,
                               pad_w=kw // 2, in_fmt=AF)
                    gate_add[g + k] = a

            # memory read-out operator: attn = softmax(q k^T / sqrt(d)), P x P per pair, once per field
            P8 = (Pn + 7) // 8 * 8
            # V as plain f16 in the read-out (one MFMA per product, 64-channel steps of hi halves: the kernel's NM 5) unless
            # VFML_ATT_V_SPLIT=1 keeps its split rows (two MFMAs per product): row stride a multiple of 64 then
            v_f16 = not os.environ.get("VFML_ATT_V_SPLIT")
            ldA = (P8 + 63) // 64 * 64 if v_f16 else (P8 + 31) // 32 * 32
            qmap = self._buf("att_q", MP * AD, dev)
            kmap = self._buf("att_k", MP * AD, dev)
            wgt, b = P["query"]
            hip.conv2d(G, 128, GLD, B, h, w, wgt, b, AD, 1, 1, qmap, AD, in0_off=INP, in_fmt=AF, out_fmt=AF)
            wgt, b = P["key"]
            hip.conv2d(G, 128, GLD, B, h, w, wgt, b, AD, 1, 1, kmap, AD, in0_off=INP, in_fmt=AF)
            scores = self._buf("att_scores", Pn * ldA, dev)
            # probabilities as ONE f16 each (round to nearest, times ATT_SCALE): the read-out streams the matrix 12 times
            # per field and is bound by its bytes; 2^-12 relative per probability, unbiased - the 1080p EPE does not move
            # (tests/test_gpu_memflow.py).  The matrix is then the WEIGHT plane of the read-out GEMM (out^T = V^T . A^T).
            plain = ATT_PLAIN and Pn >= 1024 and Pn % 4 == 0 and Pn * ldA * 2 <= 0x7ffffff0 and ldA <= 32768
            attn = []
            for k in range(B):
                kw_ = hip.SplitWeight(Pn, AD, dev).fill(kmap, src_off=k * Pn * AD, scale=16.0)
                hip.conv2d(qmap, AD, AD, 1, 1, Pn, kw_, None, Pn, 1, 1, scores, ldA, in0_off=k * Pn * AD,
                           out_scale=1.0 / float(AD) ** 0.5, in_fmt=AF)
                if plain:
                    a = self._att_planes.get(k)
                    if a is None or a.rows != Pn or a.kp != ldA or a.hi.device != dev:
                        a = self._att_planes[k] = hip.PlainWeight(Pn, ldA, dev, scale=ATT_SCALE)
                    hip.softmax_rows_f16(scores, Pn, Pn, ldA, a)
                else:
                    a = self._buf(f"att_probs{k}", Pn * ldA, dev)
                    hip.softmax_rows_s16(scores, Pn, Pn, ldA, a, ldA, scale=ATT_SCALE)
                attn.append(a)

            corr = self._buf("mcorr", MP * cor_p, dev, zero=True)
            c1 = self._buf("c1", MP * 256, dev)
            cf = self._buf("cf", MP * 256, dev)
            f1 = self._buf("f1", MP * 128, dev)
            fh = self._buf("fh", MP * 256, dev)
            val = self._buf("att_v", MP * AD, dev)
            flow4 = self._buf("flow4", MP * 4, dev)
            delta = self._buf("mdelta", MP * 4, dev, zero=True)     # channels 2,3 stay zero: no backward flow here
            coords1 = self._buf("coords1", MP * 4, dev)
            if plain:
                vrows = self._buf("att_vt", AD * ldA, dev)            # V^T as split rows [AD][ldA]
                ro = self._buf("att_ro", AD * ldA, dev)               # the GEMM's primary output [AD][ldA] (unused)
                ro_t = self._buf("att_ro_t", Pn * AD, dev)            # its transpose: attn . v, [Pn][AD]
                ro_ws = self._buf("att_ro_ws", AD * ldA + Pn * AD, dev)   # second half of the K axis (vfml.h ksplit_ws)
            else:
                vt = hip.SplitWeight(AD, P8, dev)

            hip.coords_init(coords1, B, h, w)
            hip.coords_update(coords1, None, B, h, w, flow_a=flow4, ld_a=4, flow_b=G, ld_b=GLD, flow_b_off=MF + 124,
                              fmt_b=AF)
            for it in range(cfg.decoder_depth):
                hip.corr_lookup(pyrs, hl, wl, ldl, R, Pn, coords1, 0, 4, corr, 0, cor_p, out_fmt=AF, vol_tile=TILE)
                wgt, b = P[f"{ub}.encoder.convc1"]
                hip.conv2d(corr, cor_p, cor_p, B, h, w, wgt, b, 256, 1, 1, c1, 256, epilogue=hip.EPI_RELU,
                           in_fmt=AF, out_fmt=AF)
                wgt, b = P[f"{ub}.encoder.convc2"]
                hip.conv2d(c1, 256, 256, B, h, w, wgt, b, 192, 3, 3, cf, 256, pad_h=1, pad_w=1, epilogue=hip.EPI_RELU,
                           in_fmt=AF, out_fmt=AF)
                wgt, b = P[f"{ub}.encoder.convf1"]
                hip.conv2d(flow4, 4, 4, B, h, w, wgt, b, 128, 7, 7, f1, 128, pad_h=3, pad_w=3, epilogue=hip.EPI_RELU,
                           out_fmt=AF)
                wgt, b = P[f"{ub}.encoder.convf2"]
                hip.conv2d(f1, 128, 128, B, h, w, wgt, b, 64, 3, 3, cf, 256, out_off=192, pad_h=1, pad_w=1,
                           epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF)
                # motion features [conv out (124) | fx, fy, 0, 0]: the flow quad is written by coords_update
                wgt, b = P[f"{ub}.encoder.conv"]
                hip.conv2d(cf, 256, 256, B, h, w, wgt, b, 124, 3, 3, G, GLD, out_off=MF, pad_h=1, pad_w=1,
                           epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF)
                # value map and memory read-out  m_global = m + gamma * attn . v  (per pair: its own attention)
                wgt, b = P[f"{ub}.value"]
                hip.conv2d(G, 128, GLD, B, h, w, wgt, b, AD, 1, 1, val, AD, in0_off=MF, in_fmt=AF)
                for k in range(B if plain else 0):
                    hip.transpose_to_s16(val, Pn, AD, AD, vrows, ldA, scale=16.0, src_off=k * Pn * AD)
                    hip.conv2d(vrows, ldA, ldA, AD, 1, 1, attn[k], None, Pn, 1, 1, ro, ldA, out_scale=1.0 / 16.0,
                               in_fmt=AF, out_t=ro_t, ld_out_t=AD, mfma=1 if v_f16 else 3, ksplit_ws=ro_ws)
                    hip.add_to_s16(ro_t, AD, G, GLD, G, GLD, Pn, AD, scale=gamma, aux_off=k * Pn * GLD + MF,
                                   out_off=k * Pn * GLD + MT)
                for k in range(0 if plain else B):
                    vt.fill_transposed(val, Pn, ld=AD, scale=16.0, src_off=k * Pn * AD)
                    # rows as the batch axis (1x1 "images"): one GEMM over the whole 4.2 GB attention matrix -
                    # the LDS-DMA kernel bases its source descriptor at each tile's first row
                    hip.conv2d(attn[k], P8, ldA, Pn, 1, 1, vt, None, AD, 1, 1, G, GLD, out_off=k * Pn * GLD + MT,
                               out_scale=gamma / ATT_SCALE, epilogue=hip.EPI_ADD_AUX, aux0=G, ld_aux0=GLD, aux0_off=k * Pn * GLD + MF,
                               in_fmt=AF, out_fmt=AF, aux_fmt=AF)
                for k, (kh, kw) in (("1", (1, 5)), ("2", (5, 1))):
                    wgt, _ = P[f"{ub}.gru.convzr{k}.iter"]
                    hip.conv2d(G, 128, GLD, B, h, w, wgt, None, 256, kh, kw, G, GLD, in0_off=HH, out_off=Z,
                               in1=G, c1=256, ld1=GLD, in1_off=MF, pad_h=kh // 2, pad_w=kw // 2,
                               epilogue=hip.EPI_GRU_ZR, split=128, aux0=G, ld_aux0=GLD, aux0_off=HH,
                               addend=gate_add["zr" + k], ld_addend=256, in_fmt=AF, out_fmt=AF, aux_fmt=AF)
                    wgt, _ = P[f"{ub}.gru.convq{k}.iter"]
                    hip.conv2d(G, 128, GLD, B, h, w, wgt, None, 128, kh, kw, G, GLD, in0_off=RH, out_off=HH,
                               in1=G, c1=256, ld1=GLD, in1_off=MF, pad_h=kh // 2, pad_w=kw // 2,
                               epilogue=hip.EPI_GRU_Q, aux0=G, ld_aux0=GLD, aux0_off=Z,
                               aux1=G, ld_aux1=GLD, aux1_off=HH, addend=gate_add["q" + k], ld_addend=128,
                               in_fmt=AF, out_fmt=AF, aux_fmt=AF)
                wgt, b = P[f"{ub}.flow_head.conv1"]
                hip.conv2d(G, 128, GLD, B, h, w, wgt, b, 256, 3, 3, fh, 256, in0_off=HH, pad_h=1, pad_w=1,
                           epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF)
                wgt, b = P[f"{ub}.flow_head.conv2"]
                hip.conv2d(fh, 256, 256, B, h, w, wgt, b, 2, 3, 3, delta, 4, pad_h=1, pad_w=1, in_fmt=AF)
                hip.coords_update(coords1, delta, B, h, w, flow_a=flow4, ld_a=4, flow_b=G, ld_b=GLD,
                                  flow_b_off=MF + 124, fmt_b=AF)

            mask = self._buf("mmask", MP * 576, dev)
            wgt, b = P[f"{ub}.mask.0"]
            hip.conv2d(G, 128, GLD, B, h, w, wgt, b, 256, 3, 3, fh, 256, in0_off=HH, pad_h=1, pad_w=1,
                       epilogue=hip.EPI_RELU, in_fmt=AF, out_fmt=AF)
            wgt, b = P[f"{ub}.mask.2"]
            hip.conv2d(fh, 256, 256, B, h, w, wgt, b, 576, 1, 1, mask, 576, out_scale=0.25, in_fmt=AF)
            up = torch.empty(B, H, W, 2, device=dev, dtype=torch.float32)
            for k in range(B):
                hip.convex_upsample(coords1, k * Pn * 4, 0, mask, k * Pn * 576, 576, h, w, up.view(-1), out_off=k * H * W * 2)
            low = flow4.view(B, h, w, 4)[..., :2].permute(0, 3, 1, 2).clone()
        return low, up.permute(0, 3, 1, 2)

    def _frame_context_plain(self, src, B, H, W, P, dev, Pn, AF):
        """cnet on the previous frame of every pair (frames 0..B-1): [B*Pn*256] = tanh | relu halves."""
        frames = self._buf("frames", B * H * W * 4, dev)
        hip.frames_to_nhwc4(src[0:B].contiguous(), B, H, W, float(self.cfg.input_scale), float(self.cfg.input_shift),
                            frames)
        ctx = torch.empty(B * Pn * 256, device=dev)
        self._encoder("cnet", frames, B, H, W, P, dev, ctx, 256, 0, hip.EPI_TANH_RELU, self.hidden_dim, out_fmt=AF)
        return ctx


def build_memflow_network(cfg):
    return MemFlowNetHIP(cfg)
