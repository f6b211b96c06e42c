"""cineflow.config: the reference's YAML -> constructor map (nnunet/lib/training_utils.py:459-485, :1256-1286, :1460-1537, :1938-1996).

tests/golden/configs.json holds the VALUES of the reference's four YAML files on the hot path and, per file, the parameter count /
state-dict digest of the model the reference's own classes build from them (tests/golden/make_config_fixtures.py, run against
/root/reference in the build container).  CPU only: models are built, never run."""
import hashlib
import json
import os

import numpy as np
import pytest
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fixtures():
    with open(os.path.join(HERE, "golden", "configs.json")) as f:
        return json.load(f)


def write_yaml(tmp_path, name, values):
    p = tmp_path / (name + ".yaml")
    with open(p, "w") as f:
        yaml.safe_dump(values, f)
    return str(p)


def digest(shapes, skip=("grid",)):
    items = sorted((k, tuple(int(v) for v in s)) for k, s in shapes.items() if not k.endswith(skip))
    n = sum(int(np.prod(s)) if len(s) else 1 for _, s in items)
    return {"param_count": n, "entries": len(items), "names_shapes_sha1": hashlib.sha1(json.dumps(items).encode()).hexdigest()}


def test_video_yaml_builds_the_reference_model(fixtures, tmp_path):
    from cineflow import config as C
    rec = fixtures["video"]
    cfg = C.read_config_video(write_yaml(tmp_path, "video", rec["values"]))
    net = C.build_seg_flow_gaussian_model(cfg, image_size=rec["image_size"], log_function=print)
    assert not net.motion_appearance and not net.raft and hasattr(net, "cost_volume_encoder_list")
    assert digest(net.state_shapes()) == rec["reference"]              # 25 357 698 parameters, same names and shapes
    assert rec["reference"]["param_count"] == 25357698
    assert type(C.build_flow_net(cfg, rec["image_size"])).__name__ == "SegFlowGaussian"


def test_raft_config_yaml_misses_prediction_like_the_reference(fixtures, tmp_path):
    """raft_config.yaml lacks the `prediction` key build_seg_flow_gaussian_model reads (training_utils.py:1481): KeyError there, KeyError
    here, on that key; with the value the SURVEY measured with (False) it is the 25 267 906-parameter motion_appearance model."""
    from cineflow import config as C
    rec = fixtures["raft_config"]
    assert rec["missing_key"] == "prediction"
    cfg = C.read_config_video(write_yaml(tmp_path, "raft_config", rec["values"]))
    with pytest.raises(KeyError) as e:
        C.build_seg_flow_gaussian_model(cfg, image_size=256)
    assert e.value.args[0] == "prediction"
    net = C.build_seg_flow_gaussian_model(C.with_defaults(cfg, prediction=False), image_size=rec["image_size"])
    assert net.motion_appearance and net.raft_iters == 12
    assert digest(net.state_shapes()) == rec["reference"]
    assert rec["reference"]["param_count"] == 25267906
    assert "prediction" not in cfg                                      # with_defaults copies


def test_successive_yaml_builds_the_model_pair(fixtures, tmp_path):
    from cineflow import config as C
    rec = fixtures["successive"]
    cfg = C.read_config_video(write_yaml(tmp_path, "successive", rec["values"]))
    wrap = C.build_flow_net(cfg, rec["image_size"])
    assert type(wrap).__name__ == "ModelWrap" and wrap.model1.d_model == 512
    assert digest(wrap.state_shapes()) == rec["reference"]              # 2 x 12.6 M parameters (model1: 1 input channel, model2: 6)
    assert cfg["in_encoder_dims"] == rec["values"]["in_encoder_dims"]   # the builder does not write nb_channels into the config's list


def test_adversarial_acdc_yaml_builds_the_cropping_network(fixtures, tmp_path):
    """voxelmorph_saver_Lib.py:340-348: read_config(adversarial_acdc.yaml) -> build_2d_model(..., image_size=224, window_size=7, num_classes=2)"""
    from cineflow import config as C
    rec = fixtures["adversarial_acdc"]
    cfg = C.read_config(write_yaml(tmp_path, "adversarial_acdc", rec["values"]), False, False)
    net = C.build_2d_model(cfg, conv_layer=None, norm=None, log_function=None, image_size=rec["image_size"], window_size=rec["window_size"],
                           middle=False, num_classes=rec["num_classes"], processor=None)
    skip = ("grid", "num_batches_tracked", "relative_position_index", "attn_mask")
    assert digest(net.state_shapes(), skip) == rec["reference"]


def test_readers_keep_the_reference_assertions_and_key_errors(fixtures, tmp_path):
    from cineflow import config as C
    v = dict(fixtures["video"]["values"], only_first=True, split=True)
    with pytest.raises(AssertionError):
        C.read_config_video(write_yaml(tmp_path, "bad_video", v))
    a = dict(fixtures["adversarial_acdc"]["values"])
    a["num_heads"] = [3]
    with pytest.raises(AssertionError, match="transformer_depth and num_heads"):
        C.read_config(write_yaml(tmp_path, "bad_acdc", a), False, False)
    for gone in ("d_model", "stride", "only_first"):                     # no defaults anywhere: a missing key is a KeyError naming it
        cfg = {k: x for k, x in fixtures["video"]["values"].items() if k != gone}
        with pytest.raises(KeyError) as e:
            C.build_seg_flow_gaussian_model(cfg, 256)
        assert e.value.args[0] == gone
    cfg = {k: x for k, x in fixtures["successive"]["values"].items() if k != "use_sfb"}
    with pytest.raises(KeyError):
        C.build_flow_model_successive(cfg, 256, None, nb_channels=1)


def test_values_outside_the_hot_path_are_refused_by_name(fixtures):
    from cineflow import config as C
    for key, val in (("label_input", True), ("skip_co_type", "past"), ("norm", "batch"), ("prediction", True), ("remove_GRU", True),
                     # keys the reference's constructor / forward branch on and the build does not parametrise (pinned to the shipped value)
                     ("memory_read", False), ("no_skip_co", True), ("conv_bottleneck", True), ("final_stride", 2), ("cost_volume", False),
                     ("backward_flow", False), ("gaussian", True), ("timesformer", True), ("small_memory", True), ("P", 1), ("pos_1d", "learned")):
        cfg = dict(fixtures["video"]["values"], **{key: val})
        with pytest.raises(NotImplementedError, match=key):
            C.build_seg_flow_gaussian_model(cfg, 256)
    with pytest.raises(NotImplementedError, match="no_error"):
        C.build_successive_model_wrap(dict(fixtures["successive"]["values"], no_error=True), 256)
    with pytest.raises(NotImplementedError, match="transformer_depth"):
        C.build_2d_model(dict(fixtures["adversarial_acdc"]["values"], transformer_depth=[2], num_heads=[3]), image_size=224, window_size=7, num_classes=2)


def test_voxelmorph_saver_builds_the_reference_cropping_network(fixtures, tmp_path):
    """voxelmorph_saver_Lib.py:340-348: the Processor of the saver carries MTLmodel(num_classes=2) built from adversarial_acdc.yaml"""
    from cineflow import voxelmorph_saver as VS
    from cineflow.mtl import MTLmodel
    import torch
    p = write_yaml(tmp_path, "adversarial_acdc", fixtures["adversarial_acdc"]["values"])
    net = VS.build_cropping_network(p, image_size=224, window_size=7)
    assert isinstance(net, MTLmodel) and net.num_classes == 2
    s = VS.Saver({"transpose_forward": [0, 1, 2], "transpose_backward": [0, 1, 2]}, 224, 128, device=torch.device("cpu"), cropping_network=net)
    assert s.processor.cropping_network is net and s.processor.crop_size == 128
