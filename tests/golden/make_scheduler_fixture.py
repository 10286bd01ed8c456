#!/usr/bin/env python3
"""Generate tests/golden/scheduler_oracle_run.json: the ams_amd.run scheduler executed once on the CPU oracle.

Runs in the build container (CPU only).  ``ams_amd.run.main`` takes the class behind the SemanticNetwork boundary as an argument;
here it is oracle/oracle_network.OracleSemanticNetwork (PyTorch-CPU f32 restatement), so every per-frame output below was
produced WITHOUT the HIP path.  tests/test_gpu_scheduler.py runs the same command line on the HIP-backed class and compares:
the event times, sample counts and file set must be identical; per-frame losses / mIoUs / confusion matrices agree within the
f32 error class of two fine-tune phases ("the loop itself is pinned by the build's own CPU-oracle run").
"""
import glob
import json
import random
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

from ams_amd import run as R  # noqa: E402
from oracle.oracle_network import OracleSemanticNetwork  # noqa: E402

ARGS = ["--input_video", "synthetic:25-synth:seconds=6:fps=4", "--student_checkpoint", "synthetic:0", "--gpu", "0", "--mode", "simple",
        "--height", "64", "--batch_size", "4", "--iter", "1", "--send_period", "2", "--train_period", "2", "--first_train_time", "2",
        "--memory_len", "4", "--train_strategy", "full_model"]
SEED = 5


def run(network_cls, out_dir):
    np.random.seed(SEED)
    random.seed(SEED)
    summary = R.main(ARGS + ["--output_dir", out_dir], network_cls=network_cls)
    get = lambda suffix: np.load(glob.glob(out_dir + "*_results*" + suffix)[0])  # noqa: E731
    return {"args": ARGS, "seed": SEED, "frames": summary["frames"], "mean_miou": summary["mean_miou"],
            "model_update_times": get("_model_update_times.npy").tolist(), "fps_client": get("_fps_client.npy").tolist(),
            "loss": get("_loss.npy").astype(np.float64).tolist(), "mious": get("_mious.npy").astype(np.float64).tolist(),
            "mioucats": get("_mioucats.npy").astype(np.int64).tolist(),
            "files": sorted(f.split("_64_")[-1] if "_64_" in f else f for f in (Path(p).name for p in glob.glob(out_dir + "*")))}


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        data = run(OracleSemanticNetwork, tmp + "/")
    data["produced_by"] = "oracle/oracle_network.OracleSemanticNetwork (PyTorch-CPU f32) through ams_amd.run.main"
    out = Path(__file__).resolve().parent / "scheduler_oracle_run.json"
    out.write_text(json.dumps(data, indent=0, sort_keys=True))
    print(out, out.stat().st_size, "bytes;", data["frames"], "frames, updates at", data["model_update_times"], "mean mIoU %.4f" % data["mean_miou"])
