"""Model configuration bag.

Stands in for `configs.multiframes_sintel_submission.get_cfg` of the VideoFlow submodule
(imported at reference processing/videoflow_core.py:30; mutated at :88 `cfg.model` and
:92-94 `decoder_depth / corr_levels / corr_radius` for --fast).  A plain mutable attribute bag,
like yacs' CfgNode as the reference uses it.
"""
import os


class Cfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "Cfg(" + ", ".join(f"{k}={v!r}" for k, v in sorted(self.__dict__.items())) + ")"

    def clone(self):
        return Cfg(**self.__dict__)


# precision='mixed' without an explicit plan: MFMAs per product by layer-name prefix (vfml/network.py `_nm`), chosen
# with tools/precision_plan.py under an end-point-error budget (DESIGN.md "Mixed plan"); 3 where nothing matches.
DEFAULT_MIXED_PLAN = {
    # layer-name prefix: terms of the split product (1 = both operands as plain f16).  Measured at 1080p, seq 5, one
    # layer at a time against the all-3 field (tools/precision_plan.py, profiles/r02_precision_plan.md): these are the
    # layers whose rounding barely reaches the flow - the two GRU [z | r] gate convolutions (sigmoid gates: 2.0e-5 /
    # 3.5e-5 px), the correlation volume (both operands are activations: 3.1e-5 px), the flow branch of the motion
    # encoder and the mask head (~1e-6 px each).  Everything that feeds the flow linearly (motion encoder output,
    # q gates, flow head: 2e-4 .. 8e-4 px from the weights' rounding alone) stays at 3.
    "update_block.gru.convzr1.iter": 1,
    "update_block.gru.convzr2.iter": 1,
    "corr": 1,
    "update_block.encoder.convf1": 1,
    "update_block.encoder.convf2": 1,
    "update_block.mask.0": 1,
    "update_block.mask.2": 1,
    # activations as plain f16, weights still hi + lo ("2a": two MFMAs per product): the first q gate and the temporal
    # fusion, 3.1e-5 / 2.5e-5 px on their own - the weights' rounding is what costs, not the activations'
    "update_block.gru.convq1.iter": "2a",
    "update_block.tprop": "2a",
}
# BASELINE config 5 ("BOF_things seq_len=9 1280x720 fp16"): an fp16-GRADE plan that stays inside the drop-in's 1e-3 px contract,
# which plain f16 everywhere does not (2.1e-3 .. 2.9e-3 px mean EPE against the fp32-grade field at 720p over three weight
# seeds).  One MFMA per product ("": 1) wherever a WEIGHT's rounding does not reach the flow linearly; what runs once per frame
# keeps all three terms (the context encoder alone costs ~1.3e-3 px at one term, the feature encoder and the context parts of
# the gates ~1.5e-4); the layers on the linear flow path - both correlation layers and the output of the motion encoder, the
# second q gate, both flow-head layers - take their ACTIVATIONS as plain f16 and keep the weights' lo half ("2a").
# Measured at 1280x720 seq 9 on seeds 0 / 1 / 2 (tools/precision_plan.py --arch bof --candidates: "P3";
# profiles/r03_bof_plan.md): 1.5e-4 / 1.9e-4 / 2.1e-4 px against the all-3 field at 9.2-9.4 ms per field in the sliding state
# (all-1: 8.0-8.2 ms; all-3: 10.3-10.5 ms; the 1e-4-grade DEFAULT_MIXED_PLAN: 4.5e-5 .. 6.6e-5 px, 9.3-9.6 ms).  With fewer
# layers at "2a" (the round's first candidate: without convc1 / q2) seed 2 came out at 7.5e-4 px - inside the contract, but
# three times its own seed-0 figure; the plan shipped is the one that is flat over the seeds.
_UB = "update_block"
BOF_F16_PLAN = {
    "": 1,
    "cnet": 3, "fnet": 3,
    f"{_UB}.gru.convzr1.ctx": 3, f"{_UB}.gru.convq1.ctx": 3, f"{_UB}.gru.convzr2.ctx": 3, f"{_UB}.gru.convq2.ctx": 3,
    f"{_UB}.encoder.convc1": "2a", f"{_UB}.encoder.convc2": "2a", f"{_UB}.encoder.conv": "2a",
    f"{_UB}.gru.convq2.iter": "2a",
    f"{_UB}.flow_head.conv1": "2a", f"{_UB}.flow_head.conv2": "2a",
}
# Pyramid storage that goes with the shipped mixed plan: the coarsest level as f16 (profiles/r03_corr_volume_levels.md: worst mean
# EPE over three seeds x T in {3, 5} at 1080p 8.8e-5 px against 7.7e-5 with f32 volumes, budget 1e-4; +0.6 % fields/s).  Levels
# 2-3 (1.00e-4) and beyond do not fit the budget.
DEFAULT_MIXED_CORR_VOLUME = "f16@3"
NAMED_PLANS = {"default": DEFAULT_MIXED_PLAN, "bof-f16": BOF_F16_PLAN}     # VFML_MFMA_PLAN may name one

# (A/B switch: VFML_PLAN_EXCLUDE="layer,layer" takes entries out of the default plan - those layers run all three terms)
for _k in filter(None, os.environ.get("VFML_PLAN_EXCLUDE", "").split(",")):
    DEFAULT_MIXED_PLAN.pop(_k, None)


def get_cfg():
    return Cfg(
        model="",                 # checkpoint path, set by VideoFlowCore.load_model
        network="MOFNetStack",
        feat_dim=256,             # encoder output channels; hidden = context = feat_dim // 2
        down_ratio=8,
        corr_levels=4,
        corr_radius=4,
        decoder_depth=12,         # update iterations ("default 12", reference videoflow_core.py:92)
        # Network input = input_scale * x + input_shift.  The reference hands over x in [0,1]
        # (processing/videoflow_processor.py:154), (2, -1) maps that onto [-1, 1].  The literal
        # upstream pair for 0..255 inputs would be (2/255, -1); see DESIGN.md "Input range".
        input_scale=2.0,
        input_shift=-1.0,
        # arithmetic of the conv / correlation GEMMs:
        #   'f16x3' split-f16 on the f16 matrix cores (3 MFMAs per product, ~22 mantissa bits)
        #   'f16x2' the same with every weight as one round-to-nearest f16 (2 MFMAs per product)
        #   'f16'   plain f16 operands, f32 accumulate (1 MFMA per product): the "fp16" arithmetic of BASELINE
        #           config 5 (BOF_things 720p fp16)
        #   'mixed' per layer: mfma_plan {layer-name prefix: 1 | 2 | 3}, 3 where nothing matches
        #   'f32'   exact fp32 on the f32 matrix cores (v_mfma_f32_32x32x2_f32), 5.3x slower peak
        precision="f16x3",
        mfma_plan=None,
        # storage of the correlation pyramids: 'f32', or 'f16' (opt-in; VFML_CORR_VOLUME=f16): one round-to-nearest f16 per
        # correlation value - half the store bytes of the volume GEMMs, half the pyramid memory (17 instead of 33 GB at
        # 1080p seq 5), fewer sectors per gathered window; the values the motion encoder sees then carry 11 bits (1080p:
        # mean EPE ~1e-4 px against the oracle instead of ~1e-5: inside the 1e-3 contract, outside the mixed plan's 1e-4
        # budget, hence not a default)
        corr_volume="f32",
        # the update iterations of a field as one replayed HIP graph (None: on unless VFML_GRAPH=0; results identical)
        use_graph=None,
    )
