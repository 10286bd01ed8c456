"""Every TensorFlow 1.15 op rule of SURVEY.md Appendix C, as hand-derived vectors (tests/golden/tf_semantics.json, generator
make_tf_semantics.py), against the three CPU checkers: oracle/student_torch.py, oracle/student_np.py and the graph executor
oracle/graph_interp.py.  tests/test_gpu_tf_semantics.py runs the HIP kernels against the same file."""
import json

import numpy as np
import pytest
import torch

from ams_amd import spec as S
from oracle import graph_interp as GI
from oracle import student_np as ON
from oracle import student_torch as OT


@pytest.fixture(scope="module")
def V(golden_dir):
    return json.loads((golden_dir / "tf_semantics.json").read_text())


def _mini(nodes, variables, frames, fetch, mode="train"):
    prog = {"nodes": [{"name": "features", "op": "Identity", "inputs": []}] + nodes, "output": fetch, "feed": "features"}
    return GI.GraphExecutor(prog, variables, np.float64).run(frames, mode)


def test_same_padding(V):
    for c in V["same_pad"]["cases"]:
        want = (c["out"], c["before"], c["after"])
        assert S.same_pad(c["in"], c["k"], c["stride"], c["rate"]) == want, c        # what both oracles and the engine's host side use
        assert GI._same_pads(c["in"], (c["k"] - 1) * c["rate"] + 1, c["stride"]) == want, c


def test_depthwise_on_even_sizes(V):
    for c in V["depthwise"]["cases"]:
        x = np.asarray(c["x"], np.float64)[None, :, :, None]
        w = np.asarray(c["w"], np.float64)[:, :, None, None]
        want = np.asarray(c["y"])
        got_np = ON.depthwise3x3(x, w, c["stride"], c["rate"])[0, :, :, 0]
        got_t = OT.conv_same(torch.as_tensor(x).permute(0, 3, 1, 2), torch.as_tensor(w), "dw", c["stride"], c["rate"])[0, 0].numpy()
        np.testing.assert_allclose(got_np, want, rtol=0, atol=1e-12)
        np.testing.assert_allclose(got_t, want, rtol=0, atol=1e-12)
        # the graph's own form: stride via `strides`, rate 2 via SpaceToBatchND -> VALID -> BatchToSpaceND with the paddings
        # tf.required_space_to_batch_paddings yields for base paddings [[2,2],[2,2]]
        nodes = [{"name": "w", "op": "VariableV2", "shape": [3, 3, 1, 1]}]
        if c["rate"] == 1:
            nodes.append({"name": "dw", "op": "DepthwiseConv2dNative", "inputs": [["features", 0], ["w", 0]], "padding": "SAME",
                          "strides": [1, c["stride"], c["stride"], 1], "dilations": [1, 1, 1, 1]})
            fetch = "dw"
        else:
            h, wd = x.shape[1], x.shape[2]
            extra = [(2 - (h + 4) % 2) % 2, (2 - (wd + 4) % 2) % 2]
            consts = {"bs": [2, 2], "pads": [[2, 2 + extra[0]], [2, 2 + extra[1]]], "crops": [[0, extra[0]], [0, extra[1]]]}
            for k, val in consts.items():
                a = np.asarray(val)
                nodes.append({"name": k, "op": "Const", "dtype": 3, "shape": list(a.shape), "value": a.reshape(-1).tolist(), "inputs": []})
            nodes += [{"name": "s2b", "op": "SpaceToBatchND", "inputs": [["features", 0], ["bs", 0], ["pads", 0]]},
                      {"name": "dw", "op": "DepthwiseConv2dNative", "inputs": [["s2b", 0], ["w", 0]], "padding": "VALID",
                       "strides": [1, 1, 1, 1], "dilations": [1, 1, 1, 1]},
                      {"name": "b2s", "op": "BatchToSpaceND", "inputs": [["dw", 0], ["bs", 0], ["crops", 0]]}]
            fetch = "b2s"
        got_g = _mini(nodes, {"w:0": w}, x, fetch)[0, :, :, 0]
        np.testing.assert_allclose(got_g, want, rtol=0, atol=1e-12)


def test_stem_pads_then_normalises_then_convolves(V):
    c = V["stem"]
    frame = np.asarray(c["frame_u8"], np.float64)[None]
    w = np.asarray(c["w_hwio"], np.float64)
    want = np.asarray(c["y"])
    assert c["pixel_scale"] == S.PIXEL_SCALE
    got_np = ON.conv3x3_dense(ON.preprocess(frame), w, 2)[0]
    got_t = OT.conv_same(OT.preprocess(torch.as_tensor(frame)), torch.as_tensor(w), "conv", 2, 1)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(got_np, want, rtol=0, atol=1e-6)
    np.testing.assert_allclose(got_t, want, rtol=0, atol=1e-6)
    # the reference graph's own prefix (fifo dequeue -> concat -> concat_1 -> mul_4 -> sub_2 -> MobilenetV2/Conv/Conv2D)
    ex = GI.GraphExecutor(GI.load_program("cityscapes"), {"MobilenetV2/Conv/weights:0": w}, np.float64)
    got_g = ex.run(frame, "train", fetch="MobilenetV2/Conv/Conv2D")[0]
    np.testing.assert_allclose(got_g, want, rtol=0, atol=1e-6)


def test_batch_norm_training_and_moving_averages(V):
    c = V["batch_norm_train"]
    x = np.asarray(c["x"], np.float64).reshape(1, 2, 2, 1)
    g, b = np.asarray([c["gamma"]]), np.asarray([c["beta"]])
    y, mu, var_u = ON.batch_norm_train(x, g, b, c["eps"])
    np.testing.assert_allclose(y.reshape(-1), c["y"], rtol=1e-12)
    assert mu[0] == c["batch_mean"] and var_u[0] == pytest.approx(c["batch_var_unbiased"], rel=1e-14)
    yt, mut, vart = OT.batch_norm_train(torch.as_tensor(x).permute(0, 3, 1, 2), torch.as_tensor(g), torch.as_tensor(b), c["eps"])
    np.testing.assert_allclose(yt.numpy().reshape(-1), c["y"], rtol=1e-12)
    assert float(mut) == c["batch_mean"] and float(vart) == pytest.approx(c["batch_var_unbiased"], rel=1e-14)
    taps = {}
    nodes = [{"name": "g", "op": "VariableV2", "shape": [1]}, {"name": "b", "op": "VariableV2", "shape": [1]},
             {"name": "bn/FusedBatchNormV3", "op": "FusedBatchNormV3", "inputs": [["features", 0], ["g", 0], ["b", 0]], "epsilon": c["eps"],
              "is_training": True}]
    prog = {"nodes": [{"name": "features", "op": "Identity", "inputs": []}] + nodes, "output": "bn/FusedBatchNormV3", "feed": "features"}
    yg = GI.GraphExecutor(prog, {"g:0": g, "b:0": b}, np.float64).run(x, "train", taps=taps)
    np.testing.assert_allclose(yg.reshape(-1), c["y"], rtol=1e-12)
    assert taps["bn/FusedBatchNormV3"][1][0] == pytest.approx(c["batch_var_unbiased"], rel=1e-14)
    # AssignMovingAvg with the UNBIASED variance
    for before, stat, after in ((c["moving_mean_before"], c["batch_mean"], c["moving_mean_after"]),
                                (c["moving_var_before"], c["batch_var_unbiased"], c["moving_var_after"])):
        assert float(ON.ema_update(np.float64(before), np.float64(stat))) == pytest.approx(after, rel=1e-12)
        assert float(OT.ema_update(torch.tensor(before, dtype=torch.float64), torch.tensor(stat, dtype=torch.float64))) == pytest.approx(after, rel=1e-12)
    f = V["batch_norm_frozen"]
    got = ON.batch_norm(np.asarray(f["x"]), f["gamma"], f["beta"], f["moving_mean"], f["moving_var"], S.BN_EPS_FROZEN)
    np.testing.assert_allclose(got, f["y"], rtol=1e-12)
    t = lambda a: torch.tensor([a], dtype=torch.float64)  # noqa: E731
    got_t = OT.batch_norm_frozen(torch.as_tensor(f["x"], dtype=torch.float64).view(1, 1, 2, 2), t(f["gamma"]), t(f["beta"]), t(f["moving_mean"]),
                                 t(f["moving_var"]))
    np.testing.assert_allclose(got_t.numpy().reshape(-1), f["y"], rtol=1e-12)


def test_resize_bilinear_align_corners(V):
    for c in V["resize_bilinear"]["cases"]:
        x = np.asarray(c["x"], np.float64)[None, :, :, None]
        want = np.asarray(c["y"])
        for got in (ON.resize_bilinear_align_corners(x, c["oh"], c["ow"]), OT.resize_bilinear_align_corners(torch.as_tensor(x), c["oh"], c["ow"]).numpy(),
                    GI._resize_bilinear(x, (c["oh"], c["ow"]), True, False)):
            np.testing.assert_allclose(got[0, :, :, 0], want, rtol=0, atol=1e-12)


def test_gather_argmax_labels_ce_confusion(V):
    c = V["head"]
    ci, nc = c["class_indices"], c["num_classes"]
    logits = np.asarray(c["logits"], np.float64)
    assert ON.gather_argmax(logits, ci).tolist() == c["argmax_in_subset"]
    o = OT.StudentOracle.__new__(OT.StudentOracle)        # op-level methods only: no weights needed
    o.class_indices, o.K, o.spec = torch.as_tensor(ci), len(ci), S.build_spec(nc)
    assert torch.argmax(o.reduced_logits(torch.as_tensor(logits)[None, None]), dim=-1).reshape(-1).tolist() == c["argmax_in_subset"]
    tgt, w, _ = ON.label_targets(np.asarray(c["teacher_ids"]), ci, nc)
    assert tgt.tolist() == c["target_in_subset"] and w.tolist() == c["weight"]
    tgt_t, w_t = o.label_targets(np.asarray(c["teacher_ids"]))
    assert tgt_t.tolist() == c["target_in_subset"] and w_t.tolist() == c["weight"]
    z = np.asarray(c["ce_logits"], np.float64)
    onehot = np.eye(3)[c["ce_target"]]
    assert float(ON.softmax_ce(z, onehot)) == pytest.approx(c["ce"], rel=1e-12)
    loss = o.loss_from_reduced(torch.as_tensor(z)[None], torch.tensor([c["ce_target"]]), torch.tensor([1]))
    assert float(loss) == pytest.approx(c["ce"], rel=1e-12)
    cf = c["confusion"]
    cm = ON.confusion(cf["teacher_in_subset"], cf["pred_in_subset"], cf["weight"], cf["k"])
    want = np.zeros((cf["k"], cf["k"]))
    for r, col, n in cf["nonzero"]:
        want[r, col] = n
    assert np.array_equal(cm, want) and cm.dtype == np.float64


def test_adam_tf1_form(V):
    a = V["adam_tf1"]
    for c in a["cases"]:
        b1p, b2p = a["beta1"] ** c["t"], a["beta2"] ** c["t"]
        w, m, v = ON.adam_step(np.float64(c["w"]), np.float64(c["g"]), np.float64(c["m"]), np.float64(c["v"]), c["lr"], b1p, b2p)
        assert (float(w), float(m), float(v)) == pytest.approx((c["w_after"], c["m_after"], c["v_after"]), rel=1e-12)
        t = lambda x: torch.tensor(x, dtype=torch.float64)  # noqa: E731
        wt, mt, vt = OT.adam_update(t(c["w"]), t(c["g"]), t(c["m"]), t(c["v"]), c["lr"], b1p, b2p)
        assert (float(wt), float(mt), float(vt)) == pytest.approx((c["w_after"], c["m_after"], c["v_after"]), rel=1e-12)


def test_relu6(V):
    x = np.asarray(V["relu6"]["x"])
    assert np.clip(x, 0, 6).tolist() == V["relu6"]["y"]
    assert OT.StudentOracle._act(torch.as_tensor(x), "relu6").tolist() == V["relu6"]["y"]
