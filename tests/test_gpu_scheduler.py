"""BASELINE.json configs[0] as plumbing: 64 synthetic 256x512 frames through the run.py-style scheduler (server
fine-tune phases + edge inference with metric), on the HIP path."""
import numpy as np
import pytest

from ams_amd import run as R

pytestmark = pytest.mark.gpu


def test_simple_mode_on_synthetic_clip(tmp_path):
    out = str(tmp_path / "out") + "/"
    summary = R.main(["--input_video", "synthetic:25-synth:seconds=8:fps=8", "--student_checkpoint", "synthetic:0",
                      "--output_dir", out, "--gpu", "0", "--mode", "simple", "--height", "256", "--batch_size", "4",
                      "--iter", "3", "--send_period", "1", "--train_period", "2", "--first_train_time", "2",
                      "--memory_len", "4", "--train_strategy", "coord_desc_rand"])
    assert summary["frames"] == 64 and np.isfinite(summary["mean_miou"]) and summary["frames_per_sec"] > 30
    import glob
    files = {f.split("_results_")[-1].split("_256_")[-1] if "_256_" in f else f for f in glob.glob(out + "*_results*")}
    names = " ".join(sorted(glob.glob(out + "*_results*")))
    for suffix in ("_fps_client.npy", "_bw_uplink.npy", "_bw_downlink.npy", "_model_update_times.npy", "_update.txt",
                   "_loss.npy", "_mioucats.npy", "_mious.npy", "_mioumems.npy"):
        assert suffix in names, suffix
    res = [f for f in glob.glob(out + "*_results*_model_update_times.npy")][0]
    assert np.load(res).tolist() == [0.0, 2.0, 4.0, 6.0]
    cats = np.load(res.replace("_model_update_times", "_mioucats"))
    assert cats.shape == (64, 6, 6) and cats.sum() > 0
    down = np.load(res.replace("_model_update_times", "_bw_downlink"))
    assert len(down) == 3 and all(d > 0 for d in down)
    # a 10 % coordinate-descent delta (mask bits + fp16 values, gzipped) is far smaller than the 8.45 MB model
    assert max(down) / 8 < 1.5e6
    txt = open(res.replace("_model_update_times.npy", "_update.txt")).read().split()
    assert int(txt[2]) == 3 and int(txt[4]) == 64


def test_event_times_follow_reference_formula():
    flags = R.build_parser().parse_args(["--input_video", "synthetic:25-x", "--student_checkpoint", "synthetic", "--output_dir", "o",
                                         "--mode", "simple"])
    ev = R.event_times(flags, 150)
    assert ev == [0, 100, 110, 120, 130, 140]
    flags.train_period = 30
    assert R.event_times(flags, 200) == [0, 120, 150, 180]
    flags.initial_fill = True
    flags.memory_len = 160
    assert R.event_times(flags, 200) == [0, 180]
