"""Scheduler scenarios shared by the CPU suite (oracle-backed class behind the SemanticNetwork boundary: host logic only) and
the GPU suite (the HIP-backed product class).  The assertions are about control flow — event times, sample counts, the ASR / ATR
trajectory, which models get published — and hold for either class."""
import random

import numpy as np
import pytest

from ams_amd import run as R


def _main(network_cls, argv):
    return R.main(argv, network_cls=network_cls)


def _results(out, suffix):
    import glob
    hits = glob.glob(out + "*_results*" + suffix)
    assert len(hits) == 1, (suffix, hits)
    return np.load(hits[0])


def case_asr_atr_control_loop(tmp_path, network_cls=None, sampling="reference"):
    """--enable_ASR --enable_ATR (run.py:279-307): the phi-score of the newly uploaded teacher labels moves the sampling rate by
    -0.2*tanh((phi-0.6)*20), clipped to [0.1, 1]; a low recent rate hibernates training (train period +2 s per event, up to 6x).
    The logged trajectory must follow exactly those formulas, and the publishing times must follow the logged period."""
    out = str(tmp_path / "out") + "/"
    np.random.seed(1)
    random.seed(1)
    summary = _main(network_cls, ["--input_video", "synthetic:25-synth:seconds=200:fps=1", "--student_checkpoint", "synthetic:0", "--output_dir", out,
                      "--gpu", "0", "--mode", "simple", "--height", "32", "--batch_size", "2", "--iter", "1", "--send_period", "1",
                      "--train_period", "10", "--first_train_time", "10", "--memory_len", "50", "--enable_ASR", "--enable_ATR", "--sampling", sampling])
    assert summary["frames"] == 200
    ctl = _results(out, "_control.npy")                  # second, phi, send_rate, train_period_current, hibernating
    times = _results(out, "_model_update_times.npy")
    assert ctl.shape[1] == 5 and len(ctl) == len(times) - 1 and np.array_equal(ctl[:, 0], times[1:])
    # send_rate starts at sampling_period / fps = 1 / 1 (reference run.py:115) = fps / sampling_period (per_second): at 1 fps the two
    # settings upload the same frames, so the trajectory below holds for both
    rate, deq, period, hib = 1.0, [], 10, False
    for sec, phi, got_rate, got_period, got_hib in ctl:
        if not np.isnan(phi):
            rate = float(np.clip(rate - 0.2 * np.tanh((phi - 0.6) * 20), 0.1, 1))
            deq = (deq + [rate])[-5:]
        if deq:
            if np.mean(deq) < 0.25:
                hib = True
            if np.mean(deq) > 0.35 and hib:
                hib, period = False, 10
            if hib:
                period = min(period + 2, 60)
        assert got_rate == pytest.approx(rate, abs=1e-12) and got_period == period and bool(got_hib) == hib, (sec, phi)
    assert np.all((ctl[:, 2] >= 0.1) & (ctl[:, 2] <= 1.0))
    phis = ctl[~np.isnan(ctl[:, 1]), 1]
    assert len(phis) >= 3 and np.all((phis > 0) & (phis <= 1))
    # consecutive synthetic teacher maps overlap strongly (phi > 0.6): the rate falls by 0.2 per event to its floor, the mean of
    # the last five rates drops under 0.25 at the 7th event, training hibernates and every later gap between published models
    # is 2 s longer than the one before
    assert ctl[-1, 2] == pytest.approx(0.1) and ctl[-1, 4] == 1 and ctl[5, 4] == 0 and ctl[6, 4] == 1
    gaps = np.diff(times[1:])
    assert gaps[0] == 10 and gaps.max() >= 20 and np.all(np.diff(gaps) >= 0) and np.all(np.diff(gaps[6:]) == 2)
    samples = _results(out, "_fps_client.npy")
    assert samples[:6].tolist() == [10, 8, 6, 4, 2, 1] and samples[-1] == 1     # send_rate frames per second x 10 s, then the 0.1 floor


def case_upload_period_is_the_train_period(tmp_path, network_cls=None, sampling="reference"):
    """The last argument of train_model is FLAGS.train_period (run.py:600-601): samples arrive every train_period seconds even when
    --send_period (the frame sampling period, 1 sample per send_period frames) is a different number."""
    out = str(tmp_path / "out") + "/"
    _main(network_cls, ["--input_video", "synthetic:25-synth:seconds=9:fps=6", "--student_checkpoint", "synthetic:0", "--output_dir", out, "--gpu", "0",
            "--mode", "simple", "--height", "32", "--batch_size", "2", "--iter", "1", "--send_period", "3", "--train_period", "2",
            "--first_train_time", "4", "--memory_len", "6", "--sampling", sampling])
    samples = _results(out, "_fps_client.npy")
    if sampling == "per_second":
        assert samples.tolist() == [4, 4, 4, 4]           # uploads at 2, 4, 6, 8 s: fps / send_period = 2 frames per second x 2 s
    else:
        assert samples.tolist() == [6, 6, 6, 6]           # reference run.py:115,175: the fraction send_period / fps = 0.5 of each 12-frame bucket
    assert _results(out, "_model_update_times.npy").tolist() == [0.0, 4.0, 6.0, 8.0]


def case_other_scheduler_modes(tmp_path, mode, network_cls=None):
    """run.py:593-659: `early` trains until the cut-off and serves the rest of the video with the last model, `pretrained` never
    trains, `horizon` retrains on [t-k1, t) and evaluates on [t, t+k2) for every (t, k1) and runs the pretrained pass first."""
    import glob
    out = str(tmp_path / "out") + "/"
    common = ["--input_video", "synthetic:25-synth:seconds=12:fps=4", "--student_checkpoint", "synthetic:0", "--output_dir", out, "--gpu", "0",
              "--height", "32", "--batch_size", "2", "--iter", "1", "--send_period", "4", "--train_period", "2", "--memory_len", "4",
              "--sampling", "per_second"]
    if mode == "early":
        summary = _main(network_cls, common + ["--mode", "early", "--early_cutoff_time", "4"])
        assert summary["frames"] == 48
        # events [0, 4]: the last frame before the cut-off completes second 4, which trains on what was uploaded and publishes
        assert _results(out, "_model_update_times.npy").tolist() == [0.0, 4.0]
        assert len(glob.glob(out + "early4_f4_4_*_final.pb")) == 1 and len(glob.glob(out + "early4_f4_0_*_final.pb")) == 1
        assert _results(out, "_fps_client.npy").tolist() == [2, 2]             # nothing is uploaded after the cut-off
    elif mode == "pretrained":
        summary = _main(network_cls, common + ["--mode", "pretrained"])
        assert summary["frames"] == 48 and _results(out, "_model_update_times.npy").tolist() == [0.0]
        assert len(_results(out, "_bw_downlink.npy")) == 0
    else:
        summary = _main(network_cls, common + ["--mode", "horizon", "--horizon_k1s", "2,4", "--horizon_k2", "3", "--horizon_points", "2"])
        # points t = 4 and t = 4 + (12 - 3 - 4) // 1 = 9; windows k1 = 2, 4 each; plus the pretrained pass
        labels = sorted({p.split("/")[-1].split("_results")[0] for p in glob.glob(out + "*_results*_mious.npy")})
        assert labels == sorted(["pretrained", "2__4__7_f4", "0__4__7_f4", "7__9__12_f4", "5__9__12_f4"])
        for lab, n in (("pretrained", 48), ("0__4__7_f4", 12), ("5__9__12_f4", 12)):
            assert len(np.load(glob.glob(out + lab + "_results*_mious.npy")[0])) == n
        # a window shorter than the upload period still publishes a model for its event time (the edge loads it)
        assert summary["frames"] == 12


def case_reference_sampling_default(tmp_path, network_cls=None):
    """Default uplink sampling = the reference's (run.py:115, :136-137, :175): send_rate = send_period / fps is the fraction of the
    bucket handed to choose_frames, the replay memory holds int(memory_len / send_period * fps) entries.  At send_period == fps every
    bucketed frame is uploaded; the per_second setting uploads one frame per second from the same buckets."""
    import glob
    outs = {}
    for tag, extra in (("ref", []), ("sec", ["--sampling", "per_second"])):
        out = str(tmp_path / tag) + "/"
        np.random.seed(3)
        random.seed(3)
        _main(network_cls, ["--input_video", "synthetic:25-synth:seconds=6:fps=4", "--student_checkpoint", "synthetic:0", "--output_dir", out,
                            "--gpu", "0", "--mode", "simple", "--height", "32", "--batch_size", "2", "--iter", "1", "--send_period", "4",
                            "--train_period", "2", "--first_train_time", "2", "--memory_len", "4"] + extra)
        outs[tag] = out
    assert _results(outs["ref"], "_fps_client.npy").tolist() == [8, 8, 8]      # fraction 4 / 4 = 1: all 8 frames of each 2-second bucket
    assert _results(outs["sec"], "_fps_client.npy").tolist() == [2, 2, 2]      # 1 frame per second
    for tag, total in (("ref", 24), ("sec", 6)):
        txt = open(glob.glob(outs[tag] + "*_results*_update.txt")[0]).read().split()
        assert int(txt[4]) == total


def case_edge_pipeline_equals_synchronous_loop(tmp_path, network_cls=None):
    """--edge_pipeline n (frame t + 1 submitted before frame t is collected; model reloads drain the pipeline first): every per-frame
    output file is identical to the synchronous loop's."""
    import glob
    outs = {}
    for tag, extra in (("sync", []), ("pipe", ["--edge_pipeline", "2"])):
        out = str(tmp_path / tag) + "/"
        np.random.seed(4)
        random.seed(4)
        s = _main(network_cls, ["--input_video", "synthetic:25-synth:seconds=6:fps=3", "--student_checkpoint", "synthetic:0", "--output_dir", out,
                                "--gpu", "0", "--mode", "simple", "--height", "64", "--batch_size", "2", "--iter", "1", "--send_period", "3",
                                "--train_period", "2", "--first_train_time", "2", "--memory_len", "4"] + extra)
        # (height 64: from 17 low-resolution pixels per frame on, a frame's arithmetic does not depend on how many frames share the pass;
        # below that the one-row kernel of the image-pooling branch also takes the 1x1 layers of a single frame)
        assert s["frames"] == 18
        outs[tag] = out
    for suffix in ("_loss.npy", "_mioucats.npy", "_mious.npy", "_mioumems.npy", "_model_update_times.npy"):
        a, b = _results(outs["sync"], suffix), _results(outs["pipe"], suffix)
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), suffix
