"""The reference's YAML -> constructor map for the networks of the hot path.

The reference builds its flow model from `<weights>/config.yaml` (`nnunet/run/run_training.py:191`): `read_config_video`
(`nnunet/lib/training_utils.py:478-485`) or `read_config` (`:459-476`) load the file, and one `build_*` function per model family reads
every constructor argument as `config['key']` -- no defaults, so a missing key is a `KeyError` (`build_seg_flow_gaussian_model`
`:1460-1537`, `build_flow_model_successive` `:1256-1286`, `build_2d_model` `:1938-1996`).  This module mirrors those functions by name
and by key: the same YAML files (`nnunet/raft_config.yaml`, `video.yaml`, `successive.yaml`, `adversarial_acdc.yaml`) can be handed to
it unchanged.

What is mirrored exactly
  * the key lists: every key a reference builder reads is read here, in the same order, so the FIRST missing key raises the same
    `KeyError` -- including `prediction`, which `raft_config.yaml` forgets (`training_utils.py:1481`): the shipped
    `raft_config.yaml` fails in the reference's builder and fails here.  `with_defaults(config, prediction=False)` is the documented way
    to build from it (SURVEY.md section 6 measured the reference with exactly that value supplied);
  * the assertions of the two readers;
  * `device` / `log_function` / dropout-type keys are read and ignored (inference only, weights live where they are loaded to).

What is narrower: the cineflow networks implement the dispatches the four YAML files select (SegFlowGaussian: `motion_appearance` or
the cost-volume transformer, optional RAFT loop; OpticalFlowModelSuccessive + ModelWrap; MTLmodel without Swin encoder stages).  A value
that selects another generation of the reference's code (`label_input: true`, `skip_co_type: past`, `norm: batch`, ...) raises
`NotImplementedError` naming the key, instead of silently building something else.

ruamel.yaml is absent from this image; pyyaml's safe loader reads the same documents (plain scalars, lists and comments only).
`file:line` citations are relative to /root/reference.
"""
import copy

import yaml

from .models import ModelWrap, OpticalFlowModelSuccessive, SegFlowGaussian


# ------------------------------------------------------------------------------------------------ readers
def _load(filename):
    with open(filename) as f:
        config = yaml.safe_load(f)
    if not isinstance(config, dict):
        raise ValueError("%s does not hold a YAML mapping" % filename)
    return config


def read_config(filename, middle=False, video=False):
    """nnunet/lib/training_utils.py:459-476 (the 2-D / MTLmodel configs, e.g. adversarial_acdc.yaml), assertion for assertion."""
    config = _load(filename)
    if config['bottleneck'] == 'swin_3d' or config['bottleneck'] == 'vit_3d' or config['bottleneck'] == 'factorized':
        assert config['nb_frames'] > 1, "bottleneck mode 'swin_3d', 'vit_3d' and 'factorized' require nb_frames to be more than 1"
    if config['bottleneck'] == 'factorized':
        assert len(config['patch_size']) == 2, "bottleneck mode 'factorized' require len(patch_size) to be 2"
    if filename == 'lib_config.yaml':
        assert config['semi_supervised'] == False, "can not run in a semi supervised manner with the lib dataset"  # noqa: E712
    if config['semi_supervised'] == True:  # noqa: E712
        assert config['use_spatial_transformer'] == False, "Semi supervised model can not be used with spatial transformer"  # noqa: E712
    assert len(config['transformer_depth']) == len(config['num_heads']), "transformer_depth and num_heads must have the same size"
    return config


def read_config_video(filename):
    """nnunet/lib/training_utils.py:478-485 (raft_config.yaml, video.yaml, successive.yaml)."""
    config = _load(filename)
    if config['only_first']:
        assert not config['split']
    return config


def with_defaults(config, **defaults):
    """A copy of `config` with the given keys filled in where the file forgot them (the file's own values always win)."""
    out = copy.deepcopy(dict(config))
    for k, v in defaults.items():
        out.setdefault(k, v)
    return out


# ------------------------------------------------------------------------------------------------ helpers
def _need(config, key, allowed, what):
    v = config[key]
    if v not in allowed:
        raise NotImplementedError("%s: %s = %r selects a generation of the reference's code that is outside the hot path "
                                  "(built: %s)" % (what, key, v, ", ".join(repr(a) for a in allowed)))
    return v


# keyword -> config key, in the order build_seg_flow_gaussian_model reads them (training_utils.py:1466-1532)
_SEGFLOW_KEYS = [
    ("deep_supervision", "deep_supervision"), ("no_residual", "no_residual"), ("memory_attn", "memory_attn"),
    ("motion_appearance", "motion_appearance"), ("dim_feedforward", "dim_feedforward"), ("label_pretrained", "label_input"),
    ("cross_attn_before_corr", "cross_attn_before_corr"), ("correlation_value", "correlation_value"), ("downsample_conv", "downsample_conv"),
    ("use_context_encoder", "use_context_encoder"), ("append_cat", "append_cat"), ("match_first", "match_first"), ("raft_iters", "raft_iters"),
    ("cat_correlation", "cat_correlation"), ("stride", "stride"), ("prediction", "prediction"), ("radius", "radius"), ("remove_GRU", "remove_GRU"),
    ("warp", "warp"), ("memory_read", "memory_read"), ("small_memory", "small_memory"), ("cost_volume", "cost_volume"),
    ("conv_bottleneck", "conv_bottleneck"), ("raft", "raft"), ("skip_co_depth", "skip_co_depth"), ("d_model", "d_model"), ("mamba", "mamba"),
    ("memory_length", "video_length"), ("nb_conv", "nb_conv"), ("residual", "residual"), ("query_type", "query_type"),
    ("extra_block", "extra_block"), ("nb_merging_block", "nb_merging_blocks"), ("no_skip_co", "no_skip_co"), ("P", "P"), ("no_label", "no_label"),
    ("logits_input", "logits_input"), ("nb_inputs", "nb_inputs"), ("nb_inputs_memory", "nb_inputs_memory"), ("backward_flow", "backward_flow"),
    ("gaussian", "gaussian"), ("timesformer", "timesformer"), ("supervise_iterations", "supervise_iterations"), ("deformable", "deformable"),
    ("skip_co_type", "skip_co_type"), ("shrink_select", "shrink_select"), ("bottleneck_type", "bottleneck_type"), ("marginal", "marginal"),
    ("topk", "topk"), ("pos_1d", "pos_1d"), ("norm", "norm"), ("legacy", "legacy"), ("motion_from_ed", "motion_from_ed"),
    ("one_to_all", "one_to_all"), ("all_to_all", "all_to_all"), ("final_stride", "final_stride"), ("out_encoder_dims", "out_encoder_dims"),
    ("inference_mode", "inference_mode"), ("in_dims", "in_encoder_dims"), ("nb_layers", "nb_layers"), ("conv_depth", "conv_depth"),
    ("bottleneck_heads", "bottleneck_heads"), ("drop_path_rate", "drop_path_rate"), ("only_first", "only_first"),
]


def seg_flow_gaussian_kwargs(config, image_size):
    """The reference's keyword set (name -> value) for SegFlowGaussian.__init__, read key by key like training_utils.py:1466-1532."""
    kw = {}
    for name, key in _SEGFLOW_KEYS:
        kw[name] = config[key]                       # KeyError on the first missing key, like the reference
    config['device']                                 # read at :1534 (model.to(config['device']))
    kw["image_size"] = image_size
    return kw


def build_seg_flow_gaussian_model(config, image_size, log_function=None):
    """nnunet/lib/training_utils.py:1460-1537 -> cineflow.models.SegFlowGaussian (weights are loaded afterwards, onto the device they are
    loaded to: `config['device']` is read and otherwise ignored)."""
    kw = seg_flow_gaussian_kwargs(config, image_size)
    what = "build_seg_flow_gaussian_model"
    _need(config, "label_input", (False,), what)               # forward(): label_pretrained -> forward_label_input_no_context (SegFlowGaussian.py:380-382)
    _need(config, "prediction", (False,), what)                # :386 the *_cat_prediction dispatch and its 7-channel memory encoder (:239-240)
    _need(config, "skip_co_type", ("both",), what)             # :287-296 the other skip reductions
    _need(config, "correlation_value", (False,), what)         # :262-264 an extra bottleneck-level cost volume
    _need(config, "remove_GRU", (False,), what)                # :338
    _need(config, "norm", ("group",), what)
    _need(config, "legacy", (True,), what)
    _need(config, "deep_supervision", (False,), what)          # extra decoder heads (decoder_alt.py:860-889)
    _need(config, "nb_conv", (1, 2), what)
    # keys the reference's constructor / forward also branch on and that this build does not parametrise: pinned to the value BOTH shipped
    # configurations (video.yaml, raft_config.yaml) carry, so that a file differing in one of them fails loudly instead of silently building
    # the default network.  memory_read = False, e.g., adds a skip reduction block at the bottleneck level
    # (SegFlowGaussian.py skip_co_reduction loop); no_skip_co / conv_bottleneck / final_stride / cost_volume / small_memory / P change
    # the layer lists; backward_flow / gaussian / timesformer / shrink_select / topk / marginal / pos_1d select other forward branches.
    for key, value in (("memory_read", True), ("no_skip_co", False), ("conv_bottleneck", False), ("final_stride", 1), ("cost_volume", True),
                       ("small_memory", False), ("P", 0), ("backward_flow", True), ("gaussian", False), ("timesformer", False),
                       ("shrink_select", False), ("topk", False), ("marginal", True), ("pos_1d", "sin")):
        _need(config, key, (value,), what)
    return SegFlowGaussian(image_size=image_size, in_dims=kw["in_dims"], out_encoder_dims=kw["out_encoder_dims"], d_model=kw["d_model"],
                           conv_depth=kw["conv_depth"], skip_co_depth=kw["skip_co_depth"], bottleneck_heads=kw["bottleneck_heads"],
                           nb_layers=kw["nb_layers"], dim_feedforward=kw["dim_feedforward"], motion_appearance=bool(kw["motion_appearance"]),
                           radius=kw["radius"], stride=kw["stride"], nb_conv=kw["nb_conv"], residual=bool(kw["residual"]),
                           extra_block=bool(kw["extra_block"]), downsample_conv=kw["downsample_conv"], raft=bool(kw["raft"]),
                           raft_iters=kw["raft_iters"])


# keyword -> config key of build_flow_model_successive (training_utils.py:1258-1282); nb_channels / backward / segmentation are arguments
_SUCCESSIVE_KEYS = [
    ("deep_supervision", "deep_supervision"), ("nb_conv", "nb_conv"), ("norm", "norm"), ("legacy", "legacy"), ("downsample_conv", "downsample_conv"),
    ("motion_from_ed", "motion_from_ed"), ("one_to_all", "one_to_all"), ("all_to_all", "all_to_all"), ("final_stride", "final_stride"),
    ("use_sfb", "use_sfb"), ("conv_bottleneck", "conv_bottleneck"), ("out_encoder_dims", "out_encoder_dims"), ("inference_mode", "inference_mode"),
    ("in_dims", "in_encoder_dims"), ("nb_layers", "nb_layers"), ("conv_depth", "conv_depth"), ("bottleneck_heads", "bottleneck_heads"),
    ("drop_path_rate", "drop_path_rate"), ("only_first", "only_first"),
]


def build_flow_model_successive(config, image_size, log_function=None, nb_channels=1, backward=False, segmentation=False):
    """nnunet/lib/training_utils.py:1256-1286 -> cineflow.models.OpticalFlowModelSuccessive."""
    kw = {name: config[key] for name, key in _SUCCESSIVE_KEYS}
    config['device']
    what = "build_flow_model_successive"
    _need(config, "norm", ("group",), what)
    _need(config, "legacy", (True,), what)
    _need(config, "use_sfb", (False,), what)
    _need(config, "conv_bottleneck", (False,), what)
    _need(config, "deep_supervision", (False,), what)
    if segmentation or backward:
        raise NotImplementedError("%s: segmentation / backward heads are outside the hot path" % what)
    # the reference assigns in_dims[0] = nb_channels IN the config's own list (Optical_flow_model_successive.py:249); a copy here
    return OpticalFlowModelSuccessive(image_size=image_size, nb_channels=nb_channels, in_dims=list(kw["in_dims"]),
                                      out_encoder_dims=list(kw["out_encoder_dims"]), conv_depth=list(kw["conv_depth"]),
                                      bottleneck_heads=kw["bottleneck_heads"], nb_layers=kw["nb_layers"], nb_conv=kw["nb_conv"],
                                      downsample_conv=kw["downsample_conv"])


def build_successive_model_wrap(config, image_size, log_function=None):
    """nnunet/training/network_training/nnMTLTrainerV2FlowSuccessive.py:490-496: model1 on single frames, model2 on the 6-channel
    refinement input (4 channels with `no_error`, which is not built), wrapped by ModelWrap."""
    if config['no_error']:
        raise NotImplementedError("build_successive_model_wrap: no_error = True (4-channel refinement input) is outside the hot path")
    if config['video_length'] <= 2:
        raise NotImplementedError("build_successive_model_wrap: video_length <= 2 builds no refinement network (model2 = None)")
    m1 = build_flow_model_successive(config, image_size, log_function, nb_channels=1, backward=False, segmentation=False)
    m2 = build_flow_model_successive(config, image_size, log_function, nb_channels=6, backward=False, segmentation=False)
    return ModelWrap(m1, m2)


# keyword -> config key of build_2d_model's MTLmodel branch (training_utils.py:1946-1989)
_MTL_KEYS = [
    ("device", "device"), ("transformer_bottleneck", "transformer_bottleneck"), ("separability", "separability"),
    ("adversarial_loss", "adversarial_loss"), ("asymmetric_unet", "asymmetric_unet"), ("affinity", "affinity"),
    ("add_extra_bottleneck_blocks", "add_extra_bottleneck_blocks"), ("filter_skip_co_segmentation", "filter_skip_co_segmentation"),
    ("directional_field", "directional_field"), ("classification", "classification"), ("batch_size", "batch_size"),
    ("uncertainty_weighting", "uncertainty_weighting"), ("reconstruction", "reconstruction"), ("reconstruction_skip", "reconstruction_skip"),
    ("proj", "proj"), ("shortcut", "shortcut"), ("use_conv_mlp", "use_conv_mlp"), ("similarity_down_scale", "similarity_down_scale"),
    ("concat_spatial_cross_attention", "concat_spatial_cross_attention"), ("encoder_attention_type", "encoder_attention_type"),
    ("spatial_cross_attention_num_heads", "spatial_cross_attention_num_heads"), ("merge", "merge"), ("out_encoder_dims", "out_encoder_dims"),
    ("swin_abs_pos", "swin_abs_pos"), ("patch_size", "patch_size"), ("in_dims", "in_encoder_dims"), ("deep_supervision", "deep_supervision"),
    ("bottleneck", "bottleneck"), ("drop_path_rate", "drop_path_rate"), ("conv_depth", "conv_depth"), ("transformer_depth", "transformer_depth"),
    ("num_heads", "num_heads"), ("bottleneck_heads", "bottleneck_heads"), ("num_bottleneck_layers", "num_bottleneck_layers"),
    ("rpe_mode", "rpe_mode"), ("rpe_contextual_tensor", "rpe_contextual_tensors"),
]


def build_2d_model(config, conv_layer=None, norm=None, log_function=None, image_size=224, window_size=7, middle=False, num_classes=4, processor=None):
    """nnunet/lib/training_utils.py:1938-1996 (`model: swin` branch) -> cineflow.mtl.MTLmodel.  `conv_layer` / `norm` are accepted for the
    reference's call shape (voxelmorph_saver_Lib.py:343) and must describe what the YAML says (BatchNorm2d); `load_weights` is read and
    must be false (weights are loaded by the caller)."""
    from .mtl import MTLmodel
    what = "build_2d_model"
    _need(config, "model", ("swin",), what)
    kw = {name: config[key] for name, key in _MTL_KEYS}
    _need(config, "norm", ("BatchNorm2d",), what)
    for key in ("separability", "adversarial_loss", "affinity", "directional_field", "classification", "reconstruction", "reconstruction_skip",
                "uncertainty_weighting", "shortcut", "swin_abs_pos"):
        _need(config, key, (False,), what)
    _need(config, "transformer_bottleneck", (True,), what)
    _need(config, "add_extra_bottleneck_blocks", (True,), what)
    _need(config, "rpe_mode", ("bias",), what)
    if list(kw["transformer_depth"]) or list(kw["num_heads"]):
        raise NotImplementedError("%s: Swin stages in the encoder (transformer_depth %r) are outside the hot path" % (what, kw["transformer_depth"]))
    if middle:
        raise NotImplementedError("%s: middle = True is outside the hot path" % what)
    if config['load_weights']:
        raise NotImplementedError("%s: load_weights = True reads a pretrained file at a hard-coded path; load the state dict yourself" % what)
    return MTLmodel(image_size=image_size, window_size=window_size, num_classes=num_classes, in_dims=list(kw["in_dims"]),
                    out_encoder_dims=list(kw["out_encoder_dims"]), conv_depth=list(kw["conv_depth"]),
                    spatial_cross_attention_num_heads=list(kw["spatial_cross_attention_num_heads"]), bottleneck_heads=kw["bottleneck_heads"],
                    num_bottleneck_layers=kw["num_bottleneck_layers"], asymmetric_unet=bool(kw["asymmetric_unet"]),
                    filter_skip_co_segmentation=bool(kw["filter_skip_co_segmentation"]), deep_supervision=bool(kw["deep_supervision"]),
                    processor=processor or None)


def build_flow_net(config, image_size, log_function=None):
    """The dispatch run_training.py makes through the trainer class, from the config alone: a config with `d_model` and `raft` is a
    SegFlowGaussian one (raft_config.yaml / video.yaml), one with `no_error` is the successive pair (successive.yaml)."""
    if "no_error" in config and "d_model" not in config:
        return build_successive_model_wrap(config, image_size, log_function)
    return build_seg_flow_gaussian_model(config, image_size, log_function)
