"""ams_amd/spec.py vs the reference's MetaGraphDef (tests/golden/student_graph_*.json)."""
import json

import pytest

from ams_amd import spec


@pytest.fixture(scope="module", params=["cityscapes", "pascalvoc2012"])
def graph(request, golden_dir):
    g = json.loads((golden_dir / ("student_graph_%s.json" % request.param)).read_text())
    g["num_classes"] = 19 if request.param == "cityscapes" else 21
    return g


def test_variables_names_shapes_order(graph):
    s = spec.build_spec(graph["num_classes"])
    assert [v.name for v in s.trainable] == graph["trainable_variables"]
    shapes = {v["name"]: tuple(v["shape"]) for v in graph["variables"]}
    assert len(shapes) == len(s.trainable) + len(s.stats) == 272
    for v in s.trainable + s.stats:
        assert shapes[v.name] == v.shape, v.name
    assert s.all_variable_names() == [v["name"] for v in graph["variables"]]
    if graph["num_classes"] == 19:
        assert s.n_trainable == 2113043 and s.n_stats == 33088
    # arena offsets are dense and ordered
    off = 0
    for v in s.trainable:
        assert v.offset == off
        off += v.size


def test_conv_attributes(graph):
    s = spec.build_spec(graph["num_classes"])
    nodes = {n["name"]: n for n in graph["compute_nodes"]}
    rate2 = {n["name"].rsplit("/depthwise/SpaceToBatchND", 1)[0] for n in graph["compute_nodes"]
             if n["op"] == "SpaceToBatchND"}
    for l in s.layers:
        name = l.scope + ("/depthwise" if l.kind == "dw" else "/Conv2D")
        node = nodes[name]
        assert node["op"] == ("DepthwiseConv2dNative" if l.kind == "dw" else "Conv2D")
        assert node["strides"] == [1, l.stride, l.stride, 1], name
        if l.rate == 2:
            assert l.scope in rate2 and node["padding"] == "VALID" and node["dilations"] == [1, 1, 1, 1]
        else:
            assert l.scope not in rate2 and node["padding"] == "SAME"
        if l.bn_eps is not None:
            bn = nodes[l.scope + "/BatchNorm/FusedBatchNormV3"]
            assert bn["is_training"] is True
            assert bn["epsilon"] == pytest.approx(l.bn_eps, rel=1e-7)
            assert bn["inputs"][0] in (name, l.scope + "/depthwise/BatchToSpaceND")
        else:
            assert nodes[l.scope + "/BiasAdd"]["inputs"][0] == name
        act = {"relu6": "Relu6", "relu": "Relu"}.get(l.act)
        if act:
            assert nodes[l.scope + "/" + act]["op"] == act
        else:
            assert l.scope + "/Relu6" not in nodes and l.scope + "/Relu" not in nodes
    assert len(rate2) == 3
    assert sum(1 for n in graph["compute_nodes"] if n["op"] in ("Conv2D", "DepthwiseConv2dNative")) == len(s.layers)


def test_residuals_and_head_wiring(graph):
    s = spec.build_spec(graph["num_classes"])
    nodes = {n["name"]: n for n in graph["compute_nodes"]}
    adds = sorted(n["name"] for n in graph["compute_nodes"] if n["op"] == "AddV2" and n["name"].endswith("/add")
                  and "paddings" not in n["name"])
    want = sorted("MobilenetV2/expanded_conv_%d/add" % l.block for l in s.layers if l.residual_from is not None)
    assert adds == want and len(want) == 10
    for l in s.layers:
        if l.residual_from is not None:
            add = nodes["MobilenetV2/expanded_conv_%d/add" % l.block]
            assert add["inputs"] == [l.scope + "/Identity", l.scope.rsplit("/", 1)[0] + "/input"]
    # head: Mean(axes 1,2, keep) -> image_pooling; concat_2 = [pool branch (resized), aspp0] on channels
    mean = nodes["Mean"]
    assert mean["axes"] == [1, 2] and mean["keep_dims"] is True
    assert nodes["image_pooling/Conv2D"]["inputs"][0] == "Mean"
    cat = "concat_2" if graph["num_classes"] == 19 else "concat"   # VOC graph pads with PadV2, so no concat/concat_1
    assert nodes[cat]["inputs"][:2] == ["ResizeBilinear", "aspp0/Relu"] and nodes[cat]["const"][2] == [3]
    assert nodes["concat_projection/Conv2D"]["inputs"][0] == cat
    for r in ("ResizeBilinear", "ResizeBilinear_1", "ResizeBilinear_2"):
        assert nodes[r]["align_corners"] is True and not nodes[r]["half_pixel_centers"]


def test_preprocess_constants_and_ema(graph):
    nodes = {n["name"]: n for n in graph["compute_nodes"]}
    scale_node = "mul_4" if graph["num_classes"] == 19 else "mul_2"
    assert nodes[scale_node]["const"][0] == [pytest.approx(spec.PIXEL_SCALE, rel=1e-9)]
    assert nodes["sub_2"]["const"][1] == [1.0]
    if graph["num_classes"] == 19:      # VOC graph pads with one PadV2 instead of two concats
        assert nodes["mul"]["const"][0] == [spec.PAD_VALUE] and nodes["mul_1"]["const"][0] == [spec.PAD_VALUE]
        assert nodes["concat"]["const"][2] == [1] and nodes["concat_1"]["const"][2] == [2]
    assert len(graph["ema"]) == 108 == len(graph["update_ops"])
    decays = sorted({e["decay"][0] for e in graph["ema"].values()})
    if graph["num_classes"] == 19:
        assert decays == [pytest.approx(spec.BN_DECAY, rel=1e-9)]
    assert all(e["one"] == [1.0] for e in graph["ema"].values())
    for name, e in graph["ema"].items():
        stat_out = e["stat"].rsplit(":", 1)[1]
        assert (stat_out == "1") == e["var"].endswith("moving_mean")
        assert (stat_out == "2") == e["var"].endswith("moving_variance")
    assert nodes["fifo_queue"]["capacity"] == 200


def test_derived_counts():
    assert spec.macs_per_frame(512, 1024) == 5410863296
    assert spec.macs_per_frame(256, 512) == 1404795584
    el = spec.activation_elements(512, 1024)
    assert el["total"] == 171922446 and el["conv_in"] == 85239779 and el["conv_out"] == 83153875
    sizes = spec.feature_sizes(512, 1024)
    assert sizes[0] == (257, 513) and sizes[4] == (129, 257) and sizes[-1] == (33, 65)
    assert spec.same_pad(513, 3, 2, 1) == (257, 1, 1)
    assert spec.same_pad(33, 3, 1, 2) == (33, 2, 2)
    assert spec.same_pad(300, 3, 2, 1) == (150, 0, 1)     # even size: asymmetric, extra pad goes after
