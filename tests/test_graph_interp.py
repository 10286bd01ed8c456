"""The hand-written oracles against an executor of the reference's OWN graph.

oracle/graph_interp.py runs the node list decoded from checkpoints/*/model.meta (tests/golden/student_program_*.json);
its wiring is the reference's file.  oracle/student_torch.py and oracle/student_np.py are built from ams_amd/spec.py —
the table the HIP engine is also built from — so agreement here is what stands between a wiring error in spec.py and a
green HIP-vs-oracle suite."""
import numpy as np
import pytest
import torch

from ams_amd import spec as S, synth, weights as Wt
from oracle import graph_interp as GI
from oracle import student_np as ON
from oracle import student_torch as OT

CI = [0, 1, 2, 10, 11, 13]


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("tag,nc", [("cityscapes", 19), ("pascalvoc2012", 21)])
@pytest.mark.parametrize("size", [(32, 64), (24, 40)])          # 24x40: even feature sizes -> asymmetric SAME padding
def test_oracles_follow_the_reference_graph(tag, nc, size):
    W0 = Wt.synthetic_weights(S.build_spec(nc), seed=1)
    frames = synth.SyntheticVideo(32, 2, CI, seed=4).clip()[0][:, :size[0], :size[1]].astype(np.float64)
    ex = GI.GraphExecutor(GI.load_program(tag), W0, np.float64)
    o = OT.StudentOracle(W0, CI, num_classes=nc, dtype=torch.float64)
    for mode in ("frozen", "train"):
        taps = {}
        want = ex.run(frames, mode, taps=taps)
        assert want.shape == (2, size[0], size[1], nc)
        with torch.no_grad():
            got = o.logits_full(frames, mode).numpy()
        assert rel(got, want) < 1e-7, (mode, rel(got, want))
        low = ex.run(frames, mode, fetch="logits/semantic/BiasAdd")
        got_np = ON.forward_lowres(W0, frames, mode, num_classes=nc, dtype=np.float64)
        assert low.shape == got_np.shape and rel(got_np, low) < 1e-7, mode
        if mode == "train":
            # FusedBatchNormV3 outputs 1 / 2 feed the moving averages: mean and UNBIASED variance
            assert len(taps) == 54
            for scope, (mu, var) in o.last_batch_stats.items():
                t_mu, t_var = taps[scope + "/BatchNorm/FusedBatchNormV3"]
                np.testing.assert_allclose(mu.numpy(), t_mu, rtol=1e-6, atol=1e-9)
                np.testing.assert_allclose(var.numpy(), t_var, rtol=1e-6, atol=1e-9)


def test_program_reads_every_variable_once_and_only_model_variables():
    for tag, nc in (("cityscapes", 19), ("pascalvoc2012", 21)):
        prog = GI.load_program(tag)
        names = [n["name"] + ":0" for n in prog["nodes"] if n["op"] == "VariableV2"]
        s = S.build_spec(nc)
        assert sorted(names) == sorted(v.name for v in s.trainable)     # the forward reads weights, gamma, beta (+ biases)
        ops = {n["op"] for n in prog["nodes"]}
        assert "FusedBatchNormV3" in ops and prog["output"] == "ResizeBilinear_2"


def test_f32_executor_is_as_close_to_f64_as_the_f32_oracle():
    W0 = Wt.synthetic_weights(S.build_spec(), seed=0)
    frames = synth.SyntheticVideo(32, 1, CI, seed=6).clip()[0]
    prog = GI.load_program("cityscapes")
    ref = GI.GraphExecutor(prog, W0, np.float64).run(frames, "frozen")
    f32 = GI.GraphExecutor(prog, W0, np.float32).run(frames, "frozen")
    with torch.no_grad():
        o32 = OT.StudentOracle(W0, CI).logits_full(frames.astype(np.float32), "frozen").numpy()
    assert f32.dtype == np.float32
    assert rel(f32, ref) < 1e-3 and rel(o32, ref) < 1e-3
