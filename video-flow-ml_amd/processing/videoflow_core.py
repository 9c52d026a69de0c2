"""VideoFlowCore — tensor-in / tensor-out flow engine.

API mirror of reference processing/videoflow_core.py:33-242 (constructor :52, load_model :73-128,
compute_flow_from_tensor :130-198, info/memory helpers :200-242).  The three names the reference
imports from its VideoFlow submodule (:28-30) come from the in-repo HIP engine instead.
"""
import os

import torch

from vfml import InputPadder, build_network, get_cfg
from vfml.weights import checkpoint_name, load_checked


def _is_gpu_name(dev: str) -> bool:
    return dev.startswith("cuda")


class VideoFlowCore:
    def __init__(self, device, fast_mode=False, dataset='sintel', architecture='mof', variant='standard'):
        self.device = device
        self.fast_mode = fast_mode
        self.dataset = dataset
        self.architecture = architecture.lower()
        self.variant = variant
        self.model = None
        self.cfg = None
        # Arithmetic of the engine (vfml/cfg.py `precision`; an extension - the reference has no such switch, its
        # submodule decides on autocast by itself): VFML_PRECISION = f16x3 | f16x2 | f16 | mixed | f32, and for
        # 'mixed' VFML_MFMA_PLAN = JSON {layer prefix: 1|2|"2a"|3}.  Unset: 'mixed' with the shipped per-layer plan for the
        # multi-frame network - the arithmetic bench.py's headline is measured in, held to mean EPE <= 1e-4 px against the
        # fp32 oracle at 1080p by tests/test_gpu_e2e.py (a tenth of the drop-in's 1e-3 px contract; upstream itself runs
        # autocast f16); the same plan for the tri-frame network (4.5e-5 px against its all-3 field at 720p seq 9).
        # VFML_MFMA_PLAN may also NAME a shipped plan (vfml/cfg.py NAMED_PLANS): "bof-f16" is BASELINE config 5's fp16-grade
        # arithmetic (2e-4 px at 720p, inside the 1e-3 contract - which plain 'f16' everywhere is not).
        self.precision = os.environ.get("VFML_PRECISION") or "mixed"
        self.mfma_plan = None
        if os.environ.get("VFML_MFMA_PLAN"):
            import json
            from vfml.cfg import NAMED_PLANS
            spec = os.environ["VFML_MFMA_PLAN"]
            self.mfma_plan = dict(NAMED_PLANS[spec]) if spec in NAMED_PLANS else json.loads(spec)

    # -- model ------------------------------------------------------------------------------
    def load_model(self):
        """Resolve `VideoFlow_ckpt/{ARCH}_{dataset}[_288960noise].pth` (relative to the CWD, as the
        reference does at :79-85), build the network, strict-load the weights, move to the device."""
        cfg = get_cfg()
        model_path = f"VideoFlow_ckpt/{checkpoint_name(self.architecture, self.dataset, self.variant)}"
        cfg.model = model_path
        if self.fast_mode:  # reference :91-94
            cfg.decoder_depth, cfg.corr_levels, cfg.corr_radius = 6, 3, 3
        # The reference passes the same cfg whatever the architecture (its comment at :100 expects the
        # network to be "detected from the weights"); here 'bof' selects the tri-frame network explicitly.
        cfg.network = "BOFNet" if self.architecture == "bof" else "MOFNetStack"
        if self.precision:
            cfg.precision = self.precision
            if self.precision == "mixed":
                from vfml.cfg import DEFAULT_MIXED_PLAN
                cfg.mfma_plan = dict(self.mfma_plan if self.mfma_plan is not None else DEFAULT_MIXED_PLAN)
        if os.environ.get("VFML_CORR_VOLUME"):     # 'f16' / 'f16@k': correlation pyramids (from level k up) as f16 (vfml/cfg.py)
            cfg.corr_volume = os.environ["VFML_CORR_VOLUME"]
        elif self.precision == "mixed" and self.mfma_plan is None:
            from vfml.cfg import DEFAULT_MIXED_CORR_VOLUME      # (part of the shipped plan's measured EPE budget)
            cfg.corr_volume = DEFAULT_MIXED_CORR_VOLUME
        if not os.path.exists(model_path):
            raise FileNotFoundError(f"VideoFlow model weights not found: {model_path}")
        self.cfg = cfg
        model = build_network(cfg)
        # weights_only: a third-party .pth never gets to run pickle code (torch >= 2.6 defaults to it, older did not)
        state = torch.load(model_path, map_location=self.device, weights_only=True)
        if any(k.startswith('module.') for k in state):   # DataParallel checkpoints (:106-108)
            state = {k.replace('module.', ''): v for k, v in state.items()}
        load_checked(model, state, model_path)
        model.to(self.device)
        model.eval()
        self.model = model
        arch = self.architecture.upper()
        print("[Model] VideoFlow model loaded successfully:")
        print(f"  Path: {model_path}\n  Architecture: {arch} ({self.architecture})\n  Dataset: {self.dataset}")
        print(f"  Variant: {self.variant}\n  Device: {self.device}")
        if self.fast_mode:
            print(f"  Fast Mode: Enabled (decoder depth {cfg.decoder_depth}, corr levels {cfg.corr_levels}, "
                  f"corr radius {cfg.corr_radius})")
        return model_path

    # -- inference --------------------------------------------------------------------------
    def _validate(self, x):
        if self.model is None:
            raise RuntimeError("VideoFlow model not loaded. Call load_model() first.")
        if not isinstance(x, torch.Tensor):
            raise ValueError("Input must be a torch.Tensor")
        have, want = str(x.device), str(self.device)
        # 'cuda' and 'cuda:0' name the same device (reference :160-170)
        same = have == want or (_is_gpu_name(have) and want == 'cuda') or (_is_gpu_name(want) and have == 'cuda')
        if not same:
            raise ValueError(f"Input tensor device ({x.device}) doesn't match model device ({self.device})")
        if x.dim() != 5:
            raise ValueError(f"Input tensor must have 5 dimensions [B,T,C,H,W], got {x.dim()}")
        if x.shape[0] != 1:
            raise ValueError(f"Batch size must be 1, got {x.shape[0]}")
        if x.shape[2] != 3:
            raise ValueError(f"Must have 3 color channels, got {x.shape[2]}")

    def compute_flow_from_tensor(self, frame_batch_tensor):
        """[1,T,3,H,W] float in [0,1] on the model's device -> flow tensor [2,H,W]
        (pad -> model -> unpad -> index shape[1]//2; reference :181-198)."""
        self._validate(frame_batch_tensor)
        padder = InputPadder(frame_batch_tensor.shape[-2:])
        with torch.no_grad():
            flows, _ = self.model(padder.pad(frame_batch_tensor), {})
            flows = padder.unpad(flows)
            return flows[0, flows.shape[1] // 2]

    # -- introspection ----------------------------------------------------------------------
    def is_model_loaded(self):
        return self.model is not None

    def get_model_info(self):
        if self.model is None:
            return {"status": "not_loaded"}
        cfg = self.cfg
        return {
            "status": "loaded",
            "model_path": getattr(cfg, 'model', 'unknown') if cfg else 'unknown',
            "dataset": self.dataset,
            "architecture": self.architecture.upper(),
            "variant": self.variant,
            "config": {k: getattr(cfg, k, 'default') for k in ("decoder_depth", "corr_levels", "corr_radius")},
            "fast_mode": self.fast_mode,
            "device": str(self.device),
        }

    def get_device(self):
        return self.device

    def set_eval_mode(self):
        if self.model is not None:
            self.model.eval()

    def get_memory_usage(self):
        """MB allocated / cached / peak on the GPU.  Accepts the plain-string device the CLI passes
        (the reference's `self.device.type` at :235 raises on a str)."""
        dev = torch.device(self.device) if isinstance(self.device, str) else self.device
        if dev.type != 'cuda':
            return {"message": "Memory tracking only available for CUDA devices"}
        mb = 1024 ** 2
        return {"allocated": torch.cuda.memory_allocated(dev) / mb,
                "cached": torch.cuda.memory_reserved(dev) / mb,
                "max_allocated": torch.cuda.max_memory_allocated(dev) / mb}
