#!/usr/bin/env python3
"""Benchmark of the AMS student hot path on MI355X (contract: see README / DESIGN.md §measurement).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts N fresh children (one rank per GPU, RCCL
rendezvous on 127.0.0.1) before making any GPU call itself, waits for them and prints rank 0's JSON line.  Under torchrun
(WORLD_SIZE set) it is a rank.

Workload (BASELINE.json configs[1]): student inference only, 512x1024 synthetic video, frames resident in HBM as uint8 when the
timed region starts.  One step = one pass of the hot path (frozen student forward + fused upsample/argmax -> int32 label maps) over
one batch of --batch frames per GPU.  value = frames/s over all GPUs (replicas: no collective on this path).
Extra legs on the same JSON line (none of them is `value`):
  distill        configs[2]: the 8-frame fine-tune step (BN batch statistics, backward, Adam); N > 1: weak scaling (8 frames per GPU,
                 SyncBN + one gradient all-reduce through RCCL inside the engine) and `distill_strong` = configs[4]'s shape: the
                 global 8-frame batch split N ways
  stream         configs[2] as a workload / configs[3] at N > 1: per second of video 30 single-frame inferences with metric on the
                 edge model + one 8-frame fine-tune step + server->edge hand-off, sustained; every rank runs its own video
  roofline       dominant kernel of the inference step + whole-step fractions, HIP events on the launch stream over a profiled replay
  bf16_variant   the opt-in AMS_MATMUL_BF16 plan: speed, label-mismatch fraction and mIoU delta against the default plan
  parity         HIP vs the CPU oracle on two 512x1024 frames: exact label-match fraction, logits error, mIoU against the teacher
  cpu_baseline   the CPU oracle (PyTorch-CPU restatement) on bounded samples, rank 0 / N = 1
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from ams_amd import hip, spec as S, synth, weights as Wt  # noqa: E402
from ams_amd.engine import StudentEngine  # noqa: E402

CI = [0, 1, 2, 10, 11, 13]          # exp 25 (Cityscapes) class subset, reference exp_configs.py:86-89
BF16_MFMA_PEAK_TF = 2500.0          # dense bf16 matrix pipe (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TF = 157.3            # exact-f32 matrix pipe (v_mfma_f32_16x16x4_f32), MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; value = the median window (spread reported)")
    ap.add_argument("--settle", type=float, default=0.3, help="seconds of untimed inference before the warm-up steps (plan choice + clocks)")
    ap.add_argument("--batch", type=int, default=32, help="frames per step per GPU (throughput saturates at ~24)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--train-batch", type=int, default=8)
    ap.add_argument("--stream-seconds", type=int, default=8, help="seconds of video in the interleaved infer + fine-tune leg")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-stream", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the SemanticNetwork-level legs (infer_api, distill_api)")
    ap.add_argument("--api-iterations", type=int, default=200, help="iterations of the distill_api leg (reference run.py default: 200)")
    ap.add_argument("--api-batch", type=int, default=10, help="mini_batch_size of the distill_api leg (reference run.py default: 10)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-bf16", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--only-timed", action="store_true", help="warm-up + timed loop only (what the rocprofv3 --pmc passes wrap)")
    ap.add_argument("--dump-layers", action="store_true", help="print the per-launch profile of one step to stderr")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------- launcher
def launch_ranks(args) -> int:
    """Parent of `python bench.py --gpus N` (N > 1, not under torchrun).  Makes NO GPU call: torch.cuda.device_count() does not
    initialise the runtime on this image, and a process that has touched the GPU must never re-exec.  Each child is a fresh
    interpreter with torchrun's environment contract."""
    n = args.gpus
    visible = torch.cuda.device_count()
    assert visible >= 1, "bench.py needs an MI355X"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    share = visible < n          # fewer GPUs than ranks: ranks share devices and talk over gloo — a plumbing check, not a measurement
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r % visible), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if share:
            env["AMS_BENCH_SHARED_GPUS"] = str(visible)
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if any(rcs) or not lines:
        sys.stderr.write("bench.py: rank exit codes %s\n%s\n" % (rcs, out.decode(errors="replace")[-2000:]))
        return 1
    print(lines[-1], flush=True)
    return 0


# ------------------------------------------------------------------------------------------------------------- helpers
def read_profile(eng):
    need = C.c_size_t(0)
    hip.check(eng.lib.ams_student_profile_read(eng._h, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value + 16)
    hip.check(eng.lib.ams_student_profile_read(eng._h, buf, len(buf), C.byref(need)))
    rows = []
    for line in buf.value.decode().splitlines():
        name, layer, ms, nbytes, flops, flops_x6 = line.split("\t")
        rows.append((name, int(layer), float(ms), float(nbytes), float(flops), float(flops_x6)))
    return rows


def pmc_traffic(kernel, batch, height):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (tools/pmc_traffic.sh wraps
    `bench.py --only-timed`, FETCH_SIZE and WRITE_SIZE in separate passes).  Counter units are KiB.  Correction per
    MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled; WRITE_SIZE is
    exact.  Both were re-checked on kernels of known size in this library's access patterns (profiles/*_pmc_calib*:
    1 GiB copy 0.500 / 1.000, depthwise 0.500 / 1.000).  The passes are valid for the workload they were collected at."""
    tag = "_b%d_pmc_traffic.json" % batch
    files = sorted(f for f in os.listdir(ROOT / "profiles") if f.endswith(tag))
    if not files or height != 512:
        return {"traffic": None}
    table = json.load(open(ROOT / "profiles" / files[-1]))
    for name, v in table.items():
        if kernel + "(" in name and v.get("fetch_kb_raw_avg") is not None and v.get("write_kb_raw_avg") is not None:
            fetch, write = 2.0 * v["fetch_kb_raw_avg"] * 1024, v["write_kb_raw_avg"] * 1024
            return {"traffic": round(fetch + write),
                    "traffic_detail": {"source": "profiles/" + files[-1], "fetch_bytes": round(fetch), "write_bytes": round(write),
                                       "fetch_correction": 2.0, "launches_averaged": v["launches"]}}
    return {"traffic": None}


def algorithmic_bytes(H):
    """SURVEY.md §8 d4 per-frame byte counts at f32 storage: layer-wise convention (every conv reads its input and writes its
    output once) and block-fused convention (only block inputs / outputs, the frame and the label map touch HBM)."""
    spec = S.build_spec()
    W = 2 * H
    el = S.activation_elements(H, W)
    frame, labels = H * W * 3, H * W * 4
    layerwise = 4.0 * (el["total"] + spec.n_trainable) + frame + labels
    sizes = S.feature_sizes(H, W, spec.layers)
    fused_el = 0
    block_in = None
    for l, (h, w) in zip(spec.layers, sizes):
        if l.idx == 1:
            fused_el += h * w * l.cout                       # stem output = first block's input
            block_in = h * w * l.cout
        elif l.scope.endswith("/project"):
            fused_el += block_in + h * w * l.cout + (h * w * l.cout if l.residual_from else 0)
            block_in = h * w * l.cout
    hl, wl = sizes[-1]
    fused_el += 2 * block_in + hl * wl * 19                  # head: features read by the pool and by aspp0, low-res logits out
    blockfused = 4.0 * (fused_el + spec.n_trainable) + frame + labels
    return layerwise, blockfused


def miou_of(conf):
    from ams_amd.utils import calculate_miou
    return float(np.nanmean(calculate_miou(np.asarray(conf, dtype=np.float64), nan=True)))


# ------------------------------------------------------------------------------------------------------------- rank
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    shared = os.environ.get("AMS_BENCH_SHARED_GPUS")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = None
    if world > 1 or os.environ.get("AMS_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = "gloo" if shared else "nccl"          # RCCL needs one GPU per rank
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    n_gpus = world

    H, B = args.height, args.batch
    spec = S.build_spec()
    W0 = Wt.synthetic_weights(spec, seed=0)
    video = synth.SyntheticVideo(H, max(B, args.train_batch), CI, seed=rank)
    frames_np, labels_np = video.clip()
    frames = torch.from_numpy(frames_np[:B]).to(dev)          # uint8 [B,H,2H,3] resident in HBM

    eng = StudentEngine(CI, H, 2 * H, max_batch=B, trainable=False, device=dev)
    eng.load_variables(W0)
    eng.freeze()
    def tuning(e):
        if os.environ.get("AMS_MATMUL"):               # tuning knob: 0 exact f32, 1 two-part bf16, 2 three-part bf16, 4 two-part fp16 (default)
            e.set_matmul_mode(int(os.environ["AMS_MATMUL"]))
        if os.environ.get("AMS_FUSE_DW_PROJECT"):      # tuning knob: the optional depthwise+project kernel
            e.set_fuse_dw_project(os.environ["AMS_FUSE_DW_PROJECT"] == "1")

    tuning(eng)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # Set-up, before the W warm-up steps: the first call with this batch size times the engine's two plans (one stream / two half-batches
    # on two streams) and keeps the faster; then the clocks settle under load (a cold chip measures ~3 % low over the first 20 steps:
    # 9.15 k vs 9.44 k frames/s steady state).  Both are part of bringing the engine up, like weight upload and freeze; `settle_s` is in the line.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        eng.predict(frames)
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        eng.predict(frames)
    # `windows` timed windows of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and maxed over the ranks;
    # value = the MEDIAN window (one 67 ms window is at the mercy of a clock step or a host hiccup; the spread is in the line)
    win_s = []
    for _w in range(max(1, args.windows)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = eng.predict(frames)
        barrier()
        win_s.append(max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(win_s))
    fps = n_gpus * B * args.steps / elapsed
    spread = {"windows": len(win_s), "steps_per_window": args.steps, "frames_per_sec_min": round(n_gpus * B * args.steps / max(win_s), 2),
              "frames_per_sec_max": round(n_gpus * B * args.steps / min(win_s), 2),
              "rel_spread": round((max(win_s) - min(win_s)) / elapsed, 4), "value_is": "median window"}
    checksum = int(out.sum().item())

    if args.only_timed:
        if dist is not None:
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"value": round(fps, 2), "unit": "frames/s", "ms_per_step": round(1e3 * elapsed / args.steps, 4),
                              "labels_checksum": checksum, "note": "--only-timed: no other leg was run"}), flush=True)
        return

    # ---- latency mode + hipGraph replay (single GPU only: they say nothing about scaling) -------------------------
    fps_b1 = graph_fps = graph_fps_b1 = None
    b1_micro = {}
    if n_gpus == 1:
        one = frames[:1].contiguous()
        for _ in range(3):
            eng.predict(one)
        torch.cuda.synchronize(dev)
        n1 = max(10, args.steps)

        def window(fn):
            """median of three timed windows of n1 calls (a window is ~10 ms: one host hiccup would otherwise decide the number)"""
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(n1):
                    fn()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t1)
            return sorted(ts)[1]

        fps_b1 = n1 / window(lambda: eng.predict(one))
        # the same one-frame RESULTS, several frames per pass: bit-identical per frame to the one-frame call (tests/test_gpu_fullsize.py)
        for mb in (2, 3, 4):                              # n frames in ONE pass with per-frame metrics (ams_student_predict_frames)
            pe = StudentEngine(CI, H, 2 * H, max_batch=mb, trainable=False, device=dev)
            pe.load_variables(W0)
            pe.freeze()
            tuning(pe)
            fmb = frames[:mb].contiguous()
            for _ in range(3):
                pe.predict_frames(fmb)
            b1_micro[mb] = mb * n1 / window(lambda: pe.predict_frames(fmb))
            lab_mb, _c, _l = pe.predict_frames(fmb)
            ablated = any(os.environ.get(k, "0") not in ("", "0") for k in ("AMS_XWR_ABL", "AMS_PWH_ABL", "AMS_FB_ABL"))      # measurement build, tools/ only
            assert ablated or torch.equal(lab_mb[:1], eng.predict(one)), "a frame of a %d-frame pass differs from its one-frame call" % mb
            pe.close()
            del pe
        try:
            gp = eng.graphed_predict(B)
            gp1 = eng.graphed_predict(1)
            for g_, fr_, n_, key in ((gp, frames, args.steps, "b"), (gp1, one, n1, "b1")):
                for _ in range(3):
                    g_(fr_)
                torch.cuda.synchronize(dev)
                tg = time.perf_counter()
                for _ in range(n_):
                    lab_g = g_(fr_)
                torch.cuda.synchronize(dev)
                rate = n_ * fr_.shape[0] / (time.perf_counter() - tg)
                if key == "b":
                    graph_fps = rate
                    assert int(lab_g.sum().item()) == checksum, "graph replay changed the label maps"
                else:
                    graph_fps_b1 = rate
        except Exception as e:  # noqa: BLE001
            print("hipGraph leg skipped:", e, file=sys.stderr)

    # ---- roofline leg: profiled replay of the same step ----------------------------------------------------------
    roofline = None
    kernels = {}
    layerwise_b, blockfused_b = algorithmic_bytes(H)
    if not args.no_profile and rank == 0:
        hip.check(eng.lib.ams_student_profile(eng._h, 1))
        n_prof = min(args.steps, 5)
        for _ in range(n_prof):
            eng.predict(frames)
        rows = read_profile(eng)
        hip.check(eng.lib.ams_student_profile(eng._h, 0))
        if args.dump_layers:
            per = len(rows) // n_prof
            for name, layer, ms, nbytes, flops, fx6 in rows[-per:]:
                print("%3d %-28s %8.1f us %8.1f GB/s %10.0f KB %7.1f TFLOP/s" % (layer, name, 1e3 * ms, nbytes / ms / 1e6, nbytes / 1e3, (flops + fx6) / ms / 1e9),
                      file=sys.stderr)
        agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
        for name, layer, ms, nbytes, flops, fx6 in rows:
            a = agg[name]
            a[0] += 1
            a[1] += ms
            a[2] += nbytes
            a[3] += flops
            a[4] += fx6
        total_ms = sum(a[1] for a in agg.values())
        total_bytes = sum(a[2] for a in agg.values())
        for name, (cnt, ms, nbytes, flops, fx6) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            kernels[name] = {"launches": cnt, "avg_us": round(1e3 * ms / cnt, 2), "share": round(ms / total_ms, 4),
                             "alg_GBps": round(nbytes / ms / 1e6, 1)}
            if flops + fx6:
                kernels[name]["alg_TFLOPs"] = round((flops + fx6) / ms / 1e9, 1)
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        cnt, ms, nbytes, flops, fx6 = dom[1]
        achieved = nbytes / ms / 1e6          # bytes / ms -> GB/s
        step_ms = 1e3 * elapsed / args.steps
        hbm_view = {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "alg_bytes_per_launch": round(nbytes / cnt)}
        if dom[0].startswith("block_kernel"):
            # The whole-block kernels move only block inputs and outputs; what bounds them is the EXACT-f32 matrix pipe
            # (v_mfma_f32_16x16x4_f32: 157.3 TFLOP/s = 1/16 of the bf16 rate, MI355X_MICROARCH.md), so that is the roof they are priced
            # against; the HBM view of the same launches sits beside it
            # FLOPs formed as exact-f32 MFMAs are priced at 157.3 TFLOP/s, those formed as six bf16 MFMAs on three-part splits at a
            # sixth of the dense bf16 peak: `peak` is the blend for this kernel's mix (alg. FLOPs / matrix-pipe floor time)
            tf = (flops + fx6) / ms / 1e9
            floor_ms = 1e3 * (flops / (F32_MFMA_PEAK_TF * 1e12) + 6.0 * fx6 / (BF16_MFMA_PEAK_TF * 1e12))
            peak_tf = (flops + fx6) / floor_ms / 1e9
            roofline = {"bound": "mfma", "kernel": dom[0], "achieved": round(tf, 1), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                        "frac": round(tf / peak_tf, 4), "traffic": None,
                        "alg_flops_per_launch": round((flops + fx6) / cnt),
                        "pipe_mix": {"exact_f32_mfma_flops": round(flops / cnt), "split_bf16_x6_flops": round(fx6 / cnt),
                                     "f32_peak": F32_MFMA_PEAK_TF, "bf16_peak": BF16_MFMA_PEAK_TF},
                        "hbm": hbm_view}
        else:
            roofline = {"bound": "hbm", "kernel": dom[0], "traffic": None}
            roofline.update(hbm_view)
        roofline.update({"avg_launch_us": round(1e3 * ms / cnt, 2), "launches": cnt, "step_kernel_ms": round(total_ms / n_prof, 3),
                         "note": "HIP events on the launch stream around every kernel of a profiled replay of the timed step on ONE stream (under the profiler the engine "
                                 "does not split the batch over two streams: the headline may run the same kernels as two half-size launches side by side, "
                                 "AMS_OPT_DUAL_STREAM; profiles/*_infer_kernel_stats.csv is taken with AMS_DUAL_STREAM=0 likewise); algorithmic bytes = "
                                 "f32 operands read once + results written once, algorithmic FLOPs = 2 x MACs of the block without halo or "
                                 "padding (DESIGN.md); launches of one kernel symbol are pooled (sum of work / sum of time)"})
        parts_timed = 1 if B < 8 else 3 if B in (12, 24, 48) else 2          # dual_parts_static (engine_forward.hip)
        if os.environ.get("AMS_DUAL_STREAM") == "0":
            parts_timed = 1
        roofline["plan"] = "one stream (the engine's per-launch profiler replays the step on ONE stream)"
        roofline["timed_plan"] = ("%d parts of the batch on %d streams (AMS_OPT_DUAL_STREAM static rule); rocprofv3 trace of this plan: "
                                  "profiles/r06_infer_kernel_stats_dual.csv" % (parts_timed, parts_timed)) if parts_timed > 1 else "one stream"
        roofline.update(pmc_traffic(dom[0], B, H))
        # the whole step against the three denominators of SURVEY 8 d4, over the TIMED step (not the profiled replay)
        as_built = total_bytes / n_prof
        roofline["whole_step"] = {
            "ms_per_step": round(step_ms, 4), "frames": B,
            "as_built": {"alg_bytes": round(as_built), "achieved_GBps": round(as_built / step_ms / 1e6, 1),
                         "frac": round(as_built / step_ms / 1e6 / HBM_PEAK_GBS, 4),
                         "note": "sum over the launches of this plan: fused kernels count only what they must move"},
            "layerwise_f32": {"alg_bytes": round(B * layerwise_b), "achieved_GBps": round(B * layerwise_b / step_ms / 1e6, 1),
                              "frac": round(B * layerwise_b / step_ms / 1e6 / HBM_PEAK_GBS, 4),
                              "note": "SURVEY 8 d4: every conv reads its input and writes its output once, f32 (%.0f MB/frame)" % (layerwise_b / 1e6)},
            "blockfused_f32": {"alg_bytes": round(B * blockfused_b), "achieved_GBps": round(B * blockfused_b / step_ms / 1e6, 1),
                               "frac": round(B * blockfused_b / step_ms / 1e6 / HBM_PEAK_GBS, 4),
                               "note": "SURVEY 8 d4 stretch: only block inputs / outputs, frame and labels touch HBM, f32 (%.0f MB/frame)" % (blockfused_b / 1e6)}}
        roofline["whole_step"]["headline"] = "blockfused_f32"
        roofline["whole_step_frac"] = roofline["whole_step"]["blockfused_f32"]["frac"]
        # one frame per call (the reference's call shape): as-built bytes of the B = 1 plan / time per call / HBM peak
        if fps_b1:
            hip.check(eng.lib.ams_student_profile(eng._h, 1))
            eng.predict(frames[:1].contiguous())
            rows1 = read_profile(eng)
            hip.check(eng.lib.ams_student_profile(eng._h, 0))
            bytes1 = sum(r[3] for r in rows1)
            ms1 = 1e3 / fps_b1
            roofline["batch1"] = {"launches": len(rows1), "as_built_bytes": round(bytes1), "ms_per_call": round(ms1, 4),
                                  "achieved_GBps": round(bytes1 / ms1 / 1e6, 1), "frac": round(bytes1 / ms1 / 1e6 / HBM_PEAK_GBS, 4),
                                  "kernel_ms": round(sum(r[2] for r in rows1), 4),
                                  "note": "one 512x1024 frame per call: %d dependent launches of 30-70 blocks; latency-bound, not bandwidth-bound" % len(rows1)}
        if dom[0].startswith("xdw_wreg_kernel"):
            # The fused expand+depthwise of the 160 -> 960 blocks moves 12 % of the bytes of the two kernels it replaces; what
            # bounds it is the matrix pipe (every f32 product is 6, or 3, bf16 MFMAs) next to the depthwise VALU work.  Reported
            # beside the HBM figure, not instead of it.
            hl, wl = eng.lowres
            nmma = 6 if os.environ.get("AMS_MATMUL") == "2" else 3      # default: two fp16 parts = 3 MFMAs per 32 k; AMS_MATMUL=2: three bf16 parts = 6
            flops = 2.0 * B * hl * wl * 160 * 960 * nmma
            tf_s = flops / (1e3 * ms / cnt * 1e-6) / 1e12
            roofline["matrix_pipe"] = {"bound": "mfma", "achieved": round(tf_s, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf_s / 2500.0, 4),
                                       "note": "16-bit MFMA FLOPs issued for the split products of the expand GEMM (halo rows excluded); "
                                               "f32-equivalent rate = achieved / %d" % nmma}

    # ---- parity leg (rank 0, N = 1): HIP vs the CPU oracle at the benchmark's own size -----------------------------------
    parity = None
    if not args.no_parity and rank == 0 and n_gpus == 1:
        from oracle.student_torch import StudentOracle
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        nf = 2
        oracle = StudentOracle(W0, CI)
        fr32 = frames_np[:nf].astype(np.float32)
        with torch.no_grad():
            low_o = oracle.forward_lowres(fr32, "frozen").numpy()
        lab_o, cm_o, loss_o = oracle.predict_with_metric(fr32, labels_np[:nf], "frozen")
        lab_g, cm_g, loss_g = eng.predict_with_metric(frames[:nf], torch.from_numpy(labels_np[:nf]).to(dev))
        hl, wl = eng.lowres
        low_g = eng.logits_lowres.view(-1, hl, wl, 32)[:nf, :, :, :19].cpu().numpy()
        lab_g = lab_g.cpu().numpy()
        lg = loss_g.cpu().numpy()
        # the TIMED call itself (B frames, the plan the headline ran): one frame out of each half of the batch against the oracle
        pick = [0, B - 1] if B > 1 else [0]
        lab_full = eng.predict(frames)
        low_full = eng.logits_lowres.view(-1, hl, wl, 32)[:B, :, :, :19]
        fr_pick = frames_np[pick].astype(np.float32)
        with torch.no_grad():
            low_po = oracle.forward_lowres(fr_pick, "frozen").numpy()
        lab_po = oracle.predict(fr_pick, "frozen")
        lab_pg = lab_full[pick].cpu().numpy()
        timed = {"batch": B, "plan": roofline["timed_plan"] if roofline else None, "frames_checked": pick,
                 "label_exact_match_fraction": round(float((lab_pg == lab_po).mean()), 7),
                 "label_mismatch_pixels": int((lab_pg != lab_po).sum()),
                 "logits_max_rel_err": float("%.3e" % (np.abs(low_full[pick].cpu().numpy() - low_po).max() / np.abs(low_po).max()))}
        parity = {"frames": nf, "batch": nf, "plan": "one stream (2-frame call)", "timed_call": timed,
                  "size": "%dx%d" % (H, 2 * H), "oracle": "oracle/student_torch.py, PyTorch-CPU f32 (parity unpinned: no TensorFlow here, DESIGN.md §2)",
                  "label_exact_match_fraction": round(float((lab_g == lab_o).mean()), 7),
                  "label_mismatch_pixels": int((lab_g != lab_o).sum()),
                  "logits_max_rel_err": float("%.3e" % (np.abs(low_g - low_o).max() / np.abs(low_o).max())),
                  "loss": {"hip": round(float(lg[0] / lg[1]), 6), "oracle": round(float(loss_o), 6)},
                  "miou_vs_teacher": {"hip": round(miou_of(cm_g.cpu().numpy()), 6), "oracle": round(miou_of(cm_o), 6)},
                  "note": "synthetic weights and a procedural teacher: the mIoU values are small and only their AGREEMENT is meaningful"}
    # ---- bf16 variant (BASELINE.json configs[1] says "bf16"; SURVEY 8 d6: "bf16 path: label mismatch fraction + mIoU delta") ------
    # NOT the headline: one bf16 part per operand in the late 1x1 layers (plain bf16 products, f32 accumulate, f32 storage)
    bf16_leg = None
    if not args.no_bf16 and rank == 0 and n_gpus == 1:
        lab_ref, cm_ref, _ = eng.predict_with_metric(frames, torch.from_numpy(labels_np[:B]).to(dev))
        hl, wl = eng.lowres
        low_ref = eng.logits_lowres.view(-1, hl, wl, 32)[:B, :, :, :19].clone()
        eng.set_matmul_mode(hip.MATMUL_BF16)
        for _ in range(args.warmup):
            eng.predict(frames)
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        for _ in range(args.steps):
            eng.predict(frames)
        torch.cuda.synchronize(dev)
        tb = time.perf_counter() - tb
        lab_b, cm_b, _ = eng.predict_with_metric(frames, torch.from_numpy(labels_np[:B]).to(dev))
        low_b = eng.logits_lowres.view(-1, hl, wl, 32)[:B, :, :, :19]
        m_ref, m_b = miou_of(cm_ref.cpu().numpy()), miou_of(cm_b.cpu().numpy())
        bf16_leg = {"frames_per_sec": round(B * args.steps / tb, 1), "ms_per_step": round(1e3 * tb / args.steps, 4),
                    "label_mismatch_fraction_vs_default": float("%.3e" % (lab_b != lab_ref).float().mean().item()),
                    "logits_max_rel_dev_vs_default": float("%.3e" % ((low_b - low_ref).abs().max() / low_ref.abs().max()).item()),
                    "miou_vs_teacher": {"default": round(m_ref, 6), "bf16": round(m_b, 6), "delta_points": round(100 * (m_b - m_ref), 4)},
                    "what": "AMS_MATMUL_BF16: output-stride-16 layers and head with ONE bf16 part per operand (1 MFMA per 32 k instead of 6); early "
                            "blocks exact f32; activations stored f32",
                    "note": "opt-in, outside the 1e-3 logits tolerance with the synthetic weights (random weights amplify rounding; a trained "
                            "checkpoint would sit lower): reported beside the f32-level headline, never instead of it"}
        eng.set_matmul_mode(hip.MATMUL_SPLIT_F16)
        tuning(eng)
    eng.close()
    del eng
    torch.cuda.empty_cache()

    # ---- fine-tune legs -----------------------------------------------------------------------------------------------------
    def make_sync(engine):
        """-> kwargs for StudentEngine.train_step: RCCL inside the engine (one GPU per rank), gloo callback when ranks share GPUs."""
        if dist is None:
            return {}, None
        if backend == "nccl":
            from ams_amd.dist import RcclComm
            comm = RcclComm(rank, world, dev)
            return {"comm": comm}, comm
        from ams_amd.dist import ArenaAllReduce

        class _HostReduce(ArenaAllReduce):           # gloo reduces host tensors: stage through the CPU (plumbing check only)
            def __call__(self, _user, offset, count, dtype_code):
                try:
                    t = self.view(int(offset), int(count), int(dtype_code))
                    torch.cuda.synchronize(t.device)
                    h = t.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    t.copy_(h)
                    self.calls += 1
                    self.bytes += t.numel() * t.element_size()
                    return 0
                except BaseException as e:  # noqa: BLE001
                    self.error = e
                    return 1
        red = _HostReduce(engine.arena)
        return {"allreduce": red}, red

    def time_steps(engine, tf, tl, kw, global_batch, n_steps):
        for _ in range(2):
            engine.train_step(tf, tl, 1e-3, global_batch=global_batch, **kw)
        barrier()
        ts = time.perf_counter()
        for _ in range(n_steps):
            loss = engine.train_step(tf, tl, 1e-3, global_batch=global_batch, **kw)
        barrier()
        tt = max_over_ranks(time.perf_counter() - ts)
        ls = loss.cpu().numpy()
        return tt, float(ls[0] / max(ls[1], 1))

    distill = distill_strong = None
    rccl_seen = None
    TB = args.train_batch
    n_train = max(3, min(args.steps, 10))
    if not args.no_train:
        teng = StudentEngine(CI, H, 2 * H, max_batch=TB, trainable=True, device=dev)
        teng.load_variables(W0)
        tf = torch.from_numpy(frames_np[:TB]).to(dev)
        tl = torch.from_numpy(labels_np[:TB]).to(dev)
        kw, sync = make_sync(teng)
        if sync is not None and hasattr(sync, "rank_world"):
            rccl_seen = sync.rank_world()                  # what RCCL itself reports, not WORLD_SIZE
            if rccl_seen != (rank, world):
                print("bench: RCCL communicator reports rank %d of %d, launcher says %d of %d" % (rccl_seen + (rank, world)), file=sys.stderr)
                sys.exit(3)
        tt, loss_v = time_steps(teng, tf, tl, kw, TB * n_gpus, n_train)
        step_bytes = 3.0 * TB * layerwise_b + 7 * 4.0 * spec.n_trainable + 4.0 * 2 * spec.n_stats
        ms = 1e3 * tt / n_train
        distill = {"steps_per_sec": round(n_train / tt, 3), "ms_per_step": round(ms, 3), "batch_per_gpu": TB,
                   "global_batch": TB * n_gpus, "scaling": "weak",
                   "mode": ("syncbn + gradient all-reduce, %s" % ("RCCL issued by the engine on the launch stream" if backend == "nccl" else "gloo host callback (ranks share GPUs)"))
                   if dist is not None else "single",
                   "loss": round(loss_v, 5),
                   "roofline": {"bound": "hbm", "alg_bytes_per_step": round(step_bytes), "achieved": round(step_bytes / ms / 1e6, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(step_bytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                                "note": "SURVEY 8 d4: 3 x B x layer-wise f32 inference bytes + optimizer (7 x 4 B x params) + BN moving averages, per GPU"}}
        # HBM bytes of one step from the committed counter passes (tools/profile_r03.sh `pmc`; valid for 8 frames of 512x1024 on one GPU)
        tfile = sorted(f for f in os.listdir(ROOT / "profiles") if f.endswith("_train_pmc_traffic.json"))
        if tfile and TB == 8 and H == 512 and dist is None:
            tj = json.load(open(ROOT / "profiles" / tfile[-1]))
            distill["roofline"]["traffic"] = round(tj["total_GB_per_step"] * 1e9)
            distill["roofline"]["traffic_detail"] = {"source": "profiles/" + tfile[-1], "fetch_bytes": round(tj["fetch_GB_per_step"] * 1e9),
                                                     "write_bytes": round(tj["write_GB_per_step"] * 1e9), "fetch_correction": 2.0,
                                                     "hbm_GBps_at_this_step_time": round(tj["total_GB_per_step"] * 1e3 / ms, 1)}
        else:
            distill["roofline"]["traffic"] = None
        if sync is not None and hasattr(sync, "stats"):
            calls, nbytes = sync.stats()
            distill["collectives_per_step"] = round(calls / (n_train + 2), 1)
            distill["bytes_reduced_per_step"] = round(nbytes / (n_train + 2))

        def collective_split(engine, tf_, tl_, global_batch, step_ms):
            """Diagnostic, after the timed steps: three more steps with a HIP event pair around every collective (ams_comm_set_timing) ->
            how much of a step the launch stream spends INSIDE ncclAllReduce (waiting for the peers included), per rank."""
            if sync is None or not hasattr(sync, "set_timing"):
                return None
            sync.set_timing(True)
            for _ in range(3):
                engine.train_step(tf_, tl_, 1e-3, global_batch=global_batch, **kw)
            tot, mx, n = sync.timing()
            sync.set_timing(False)
            per = tot / 3.0
            t = torch.tensor([per], dtype=torch.float64, device=dev)
            if dist is not None and backend == "nccl":
                gathered = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(gathered, t)
                per_rank = [round(float(x.item()), 3) for x in gathered]
            else:
                per_rank = [round(per, 3)]
            return {"collective_ms_per_step": round(max(per_rank), 3), "collective_ms_per_step_by_rank": per_rank,
                    "collectives_timed_per_step": round(n / 3.0, 1), "longest_collective_ms": round(mx, 4),
                    "compute_ms_per_step_estimate": round(step_ms - max(per_rank), 3),
                    "note": "HIP events around each ncclAllReduce on the launch stream over three extra (untimed) steps; a span includes the wait "
                            "for the slowest peer, so compute = step - collective is a lower bound of the pure kernel time"}
        cs = collective_split(teng, tf, tl, TB * n_gpus, ms)
        if cs:
            distill.update(cs)
        # configs[4]'s shape: ONE 8-frame batch split over the ranks (strong scaling; SyncBN keeps full-batch semantics)
        if dist is not None and TB % n_gpus == 0:
            per = TB // n_gpus
            lo = rank * per
            tfs, tls = tf[lo:lo + per].contiguous(), tl[lo:lo + per].contiguous()
            tt2, loss2 = time_steps(teng, tfs, tls, kw, TB, n_train)
            cs2 = collective_split(teng, tfs, tls, TB, 1e3 * tt2 / n_train) or {}
            distill_strong = {**cs2, "steps_per_sec": round(n_train / tt2, 3), "ms_per_step": round(1e3 * tt2 / n_train, 3), "batch_per_gpu": per,
                              "global_batch": TB, "scaling": "strong", "loss": round(loss2, 5),
                              "note": "BASELINE.json configs[4] shape: one stream's 8-frame fine-tune batch sharded over the GPUs; 110 latency-bound "
                                      "collectives per step (SURVEY 8 e4: communication-latency-bound by construction)"}
        if sync is not None and hasattr(sync, "close"):
            sync.close()
        teng.close()
        del teng
        torch.cuda.empty_cache()

    # ---- stream leg: configs[2] as a workload (N = 1) / configs[3] (N > 1: one video and one student pair per GPU, no collective) ----
    stream = None
    if not args.no_stream and not args.no_train:
        fps_video = 30
        secs = args.stream_seconds
        clip_f, clip_l = synth.SyntheticVideo(H, 40, CI, seed=100 + rank).clip()
        vf = torch.from_numpy(clip_f).to(dev)                # 40 distinct frames resident in HBM, cycled
        vl = torch.from_numpy(clip_l).to(dev)
        server = StudentEngine(CI, H, 2 * H, max_batch=TB, trainable=True, device=dev)
        server.load_variables(W0)
        PASS = 3                                              # frames per inference pass of the edge (the edge holds two frames: 67 ms at 30 fps)
        edge = StudentEngine(CI, H, 2 * H, max_batch=PASS, trainable=False, device=dev)
        edge.load_variables(W0)
        edge.freeze()

        train_stream = torch.cuda.Stream(device=dev)

        def one_second(sec, overlap, PASS=PASS):
            # The server's step of this second trains on frames it already holds and does not depend on the edge's inferences of the same
            # second, which use the model of the previous hand-off — in the deployed system the two run on different machines at the same
            # time.  overlap: the step goes to a second stream beside the edge's 30 single-frame calls and joins before the hand-off.
            conf_sum = torch.zeros(len(CI), len(CI), dtype=torch.int64, device=dev)
            main = torch.cuda.current_stream(dev)
            if overlap:
                train_stream.wait_stream(main)
                with torch.cuda.stream(train_stream):         # (enqueuing it from a second host thread measured the same: 40.5x)
                    idx_s = torch.arange(TB, device=dev) * 5 % vf.shape[0]
                    server.train_step(vf[idx_s], vl[idx_s], 1e-3)
            for k in range(0, fps_video, PASS):               # edge: every frame of this second with its own metrics, PASS frames per pass
                idx_e = (torch.arange(k, min(k + PASS, fps_video), device=dev) + sec * fps_video) % vf.shape[0]
                _lab, confs, _loss = edge.predict_frames(vf[idx_e], vl[idx_e])
                conf_sum += confs.sum(0)
            if overlap:
                main.wait_stream(train_stream)
            else:
                idx = torch.arange(TB, device=dev) * 5 % vf.shape[0]
                server.train_step(vf[idx], vl[idx], 1e-3)     # server: one 8-frame fine-tune step per second (configs[2])
            edge.params.copy_(server.params)                  # hand-off: trained variables -> edge, BN folded again
            edge.stats.copy_(server.stats)
            edge.freeze()
            return conf_sum

        def run_seconds(overlap, frames_per_pass):
            one_second(0, overlap, frames_per_pass)
            barrier()
            t_start = time.perf_counter()
            total = torch.zeros(len(CI), len(CI), dtype=torch.int64, device=dev)
            for sec in range(secs):
                total += one_second(sec + 1, overlap, frames_per_pass)
            barrier()
            return max_over_ranks(time.perf_counter() - t_start), total
        tt_seq, _ = run_seconds(False, 1)
        tt, conf_total = run_seconds(True, 1)            # the reference's call shape: ONE frame per edge call (run.py:422-423) — the tracked figure
        tt_pipe, _ = run_seconds(True, PASS)
        stream = {"videos": n_gpus, "video_seconds_each": secs, "wall_s": round(tt, 4),
                  "sustained_frames_per_sec": round(n_gpus * secs * fps_video / tt, 1),
                  "realtime_factor_per_video": round(secs / tt, 2),
                  "realtime_factor_sequential": round(secs / tt_seq, 2),
                  "realtime_factor_pipelined": round(secs / tt_pipe, 2),
                  "sustained_frames_per_sec_pipelined": round(n_gpus * secs * fps_video / tt_pipe, 1),
                  "rccl_ranks": (rccl_seen[1] if rccl_seen else (1 if dist is None else None)),
                  "overlap": "the server's fine-tune step of a second runs on a second stream beside the edge's 30 inferences of that second (they are "
                             "independent until the hand-off, as in the deployed system); realtime_factor_sequential is the same work one after the other",
                  "per_video_second": "30 frames labelled with per-frame metrics on the edge model, ONE frame per call (realtime_factor_per_video, _sequential: the reference's "
                                      "synchronous per-frame shape) or %d frames per pass (realtime_factor_pipelined: ams_student_predict_frames, the edge holds %d frames) "
                                      "+ 1 x %d-frame fine-tune step + server->edge hand-off (device copy + BN fold)" % (PASS, PASS - 1, TB),
                  "miou_vs_teacher_rank0": round(miou_of(conf_total.cpu().numpy()), 5),
                  "target": ">= 30 frames/s of inference + one 8-frame step per second on one GPU (BASELINE.json north star): realtime_factor >= 1",
                  "note": "configs[2] interleaved on one GPU; with N > 1 configs[3]: every rank serves its own video with its own student pair, no collective"}
        server.close()
        edge.close()
        del server, edge
        torch.cuda.empty_cache()

    # ---- API-level legs: what run.py would see (reference run.py:32-39 defaults, :309-313 per-phase timing, :422-423 the per-frame loop) ----
    infer_api = distill_api = None
    if not args.no_api and rank == 0 and n_gpus == 1:
        from collections import deque
        from ams_amd.semantic_network import SemanticNetwork, FrozenGraph
        cw = np.zeros((19, 1)); cw[CI] = 1
        n_api = 200
        # infer_api: the edge loop — 200 x predict_with_metric on HOST uint8 frames [1,H,2H,3] with host labels, results back on the host
        edge_net = SemanticNetwork("<bench>", class_weights_exp=cw, height=H, frozen=True,
                                   frozen_graph=FrozenGraph(W0, np.array(CI), H, 19))
        hf = [frames_np[k % len(frames_np)][None] for k in range(8)]
        hl_ = [labels_np[k % len(labels_np)][None] for k in range(8)]
        for k in range(5):
            edge_net.predict_with_metric(hf[k % 8], hl_[k % 8])
        ta = time.perf_counter()
        for k in range(n_api):
            edge_net.predict_with_metric(hf[k % 8], hl_[k % 8])
        ta = time.perf_counter() - ta
        # the engine-only rate of the same call: frames and labels resident in HBM, nothing fetched
        df, dl = torch.from_numpy(hf[0]).to(dev), torch.from_numpy(hl_[0]).to(dev)
        torch.cuda.synchronize(dev)
        te = time.perf_counter()
        for k in range(n_api):
            edge_net.engine.predict_with_metric(df, dl)
        torch.cuda.synchronize(dev)
        te = time.perf_counter() - te
        infer_api = {"calls": n_api, "frames_per_sec": round(n_api / ta, 1), "ms_per_call": round(1e3 * ta / n_api, 4),
                     "engine_only_frames_per_sec": round(n_api / te, 1), "engine_only_ms_per_call": round(1e3 * te / n_api, 4),
                     "host_overhead_ms_per_call": round(1e3 * (ta - te) / n_api, 4),
                     "what": "SemanticNetwork.predict_with_metric(frame [1,%d,%d,3] uint8 ndarray, labels [1,%d,%d] uint8 ndarray) -> labels, confusion matrix, "
                             "IoU, mIoU, loss on the host: H2D of 1.6 MB + the one-frame pass + ONE D2H of labels and metrics + calculate_miou, "
                             "synchronous, as the reference's run.py:422-423 loop calls it" % (H, 2 * H, H, 2 * H)}
        edge_net.close_model()
        del edge_net
        torch.cuda.empty_cache()
        if not args.no_train:
            # distill_api: the server's phase at the reference's defaults — train_with_deque over a 250-frame replay memory, 200 iterations
            # of mini_batch_size 10 (run.py:32-39), timed as run.py:309-313 times it (sampling thread + staging thread + steps + the
            # variable read-back at the end of _train)
            iters, mb = args.api_iterations, args.api_batch
            net = SemanticNetwork("<bench>", class_weights_exp=cw, height=H, frozen=False, scale=[1], mini_batch_size=mb, lr=1e-3,
                                  initial_variables=W0)
            replay_f = deque(frames_np[k % len(frames_np)] for k in range(250))
            replay_l = deque(labels_np[k % len(labels_np)] for k in range(250))
            net.train_with_deque(replay_f, replay_l, 5)                           # warm-up (streams, pinned staging ring, first launches)
            torch.cuda.synchronize(dev)
            td = time.perf_counter()
            net.train_with_deque(replay_f, replay_l, iters)
            torch.cuda.synchronize(dev)
            td = time.perf_counter() - td
            loop_ms = net._last_train_ms
            # engine-only rate at the same batch size: frames resident in HBM
            ef = torch.from_numpy(np.stack([frames_np[k % len(frames_np)] for k in range(mb)])).to(dev)
            el = torch.from_numpy(np.stack([labels_np[k % len(labels_np)] for k in range(mb)])).to(dev)
            for _ in range(3):
                net.engine.train_step(ef, el, 1e-3)
            torch.cuda.synchronize(dev)
            te = time.perf_counter()
            n_e = 30
            for _ in range(n_e):
                net.engine.train_step(ef, el, 1e-3)
            torch.cuda.synchronize(dev)
            te = (time.perf_counter() - te) / n_e
            distill_api = {"iterations": iters, "mini_batch_size": mb, "replay_frames": 250,
                           "iterations_per_sec": round(iters / td, 2), "ms_per_iteration": round(1e3 * td / iters, 3),
                           "train_loop_ms_per_iteration": round(loop_ms / iters, 3),
                           "engine_only_iterations_per_sec": round(1.0 / te, 2), "engine_only_ms_per_iteration": round(1e3 * te, 3),
                           "api_over_engine": round((iters / td) * te, 4),
                           "what": "SemanticNetwork.train_with_deque(deque of 250 uint8 frames, deque of labels, %d iterations) at mini_batch_size %d, %dx%d: "
                                   "host sampling (np.random.choice per frame, the reference's draw order) into a pinned staging ring, H2D on a copy stream, "
                                   "the fine-tune steps, and the read-back of all variables at the end of the phase; train_loop_ms = the loop alone "
                                   "(what the reference prints per phase, run.py:311-313)" % (iters, mb, H, 2 * H)}
            net.close_model()
            del net
            torch.cuda.empty_cache()

    # ---- CPU baseline legs: the oracle on the host cores (rank 0, N = 1 only) --------------------------------------------------
    cpu = None
    if not args.no_cpu and rank == 0 and n_gpus == 1:
        from oracle.student_torch import StudentOracle
        # PyTorch-CPU oversubscribes badly on very wide hosts (256 hardware threads: 55 s per frame); 32 threads is the
        # fastest setting measured on the GPU box, and `cores` reports the threads actually used
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        oracle = StudentOracle(W0, CI)
        oracle.predict(frames_np[:1].astype(np.float32))                           # warm-up
        rates = []
        for rep in range(3):                                                       # three bounded samples: the spread is part of the figure
            tc = time.perf_counter()
            n_cpu = 0
            while time.perf_counter() - tc < 4.0 and n_cpu < 16:
                oracle.predict(frames_np[n_cpu % len(frames_np)][None].astype(np.float32))
                n_cpu += 1
            rates.append(n_cpu / (time.perf_counter() - tc))
        legs = {}
        # configs[0]: 64 frames of 256x512 (bounded to 6 s)
        small_f = synth.SyntheticVideo(256, 8, CI, seed=3).clip()[0].astype(np.float32)
        oracle.predict(small_f[:1])
        tc, n_small = time.perf_counter(), 0
        while n_small < 64 and time.perf_counter() - tc < 6.0:
            oracle.predict(small_f[n_small % 8][None])
            n_small += 1
        legs["cfg0_infer_256x512"] = {"value": round(n_small / (time.perf_counter() - tc), 3), "unit": "frames/s", "frames": n_small}
        # configs[2]: one 8-frame fine-tune step at 512x1024 (forward with batch statistics, autograd backward, Adam), once
        if not args.no_train:
            tc = time.perf_counter()
            oracle.train_step(frames_np[:TB].astype(np.float32), labels_np[:TB], 1e-3)
            dt = time.perf_counter() - tc
            legs["cfg2_train_step_%dx512x1024" % TB] = {"value": round(1.0 / dt, 4), "unit": "steps/s", "seconds": round(dt, 2), "steps": 1}
        cpu = {"value": round(float(np.median(rates)), 3), "unit": "frames/s", "cores": cores, "kind": "port",
               "spread": {"min": round(min(rates), 3), "max": round(max(rates), 3), "samples": 3},
               "legs": legs,
               "sample": "3 x <= 4 s of single 512x1024 frames of the same synthetic clip, PyTorch-CPU f32 restatement "
                         "(oracle/student_torch.py) with %d threads, median reported; stand-in for the reference's TF1 CPU path, which "
                         "cannot run here (TF 1.15 absent)" % torch.get_num_threads()}

    if rank == 0:
        result = {
            "metric": "frames/sec student infer (DeeplabV3+MobileNetV2, 512x1024) + distill-steps/sec",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "settle_s": args.settle,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "spread": spread, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (2xfp16 split products)", "data": "synthetic",
            "rccl_ranks": (rccl_seen[1] if rccl_seen else (1 if dist is None else None)),
            "rccl_ranks_source": ("ams_comm_stats (the library's RCCL communicator)" if rccl_seen else
                                  "no RCCL communicator in this run" + ("" if dist is None else " (ranks share GPUs: gloo host callback)")),
            "precision_note": "f32 accumulation everywhere, f32 storage of every tensor the network defines; the products of the 1x1 layers (whole-block "
                              "kernels of the early section, output-stride-16 section, head) are formed as 3 fp16 MFMAs on two-part splits of the f32 "
                              "operands, x ~ hi + lo 2^-11 (22 significand bits; AMS_MATMUL_SPLIT_F16): f32-level — 512x1024 logits 3-4e-5 from the f64 "
                              "oracle, as exact f32 MFMA (3.7e-5), the three-part bf16 split of rounds 1-4 (3.7e-5) and the f32 CPU oracle (4.2e-5): "
                              "tools/logit_error.py; the depthwise result of a stride-16 block travels to its project GEMM as those fp16 pairs (4 bytes per "
                              "value, like f32); the first block's stem and project run on the same two fp16 parts (an fp16 table of the 256 byte values; first_block_walk_kernel), the 64-wide project of block 6 on exact f32 MFMA; "
                              "the fine-tune step forms every split product on three bf16 parts (6 MFMAs)",
            "kernels": kernels,
            "config": {"workload": "student infer only, %dx%d synthetic clip, frozen BN, uint8 frames resident in HBM, "
                                   "int32 label maps out (BASELINE.json configs[1])" % (H, 2 * H),
                       "frames_per_step_per_gpu": B, "class_subset": CI, "weights": "synthetic seed 0",
                       "parallelism": "replicas x%d (no collective on the inference path)" % n_gpus},
            "frames_per_sec_batch1": round(fps_b1, 2) if fps_b1 else None,
            "frames_per_sec_batch1_pipelined": ({**{("frames_per_pass_%d" % k): round(v, 2) for k, v in b1_micro.items()},
                                                 "note": "one-frame results, several frames per pass: frames_per_pass_n = n consecutive frames labelled in ONE pass "
                                                         "with per-frame metrics (ams_student_predict_frames; the edge holds n - 1 frames: 33 ms each at 30 fps); "
                                                         "per-frame results are bit-identical to the one-frame call"}
                                                if b1_micro else None),
            "hipgraph": {"frames_per_sec": round(graph_fps, 2) if graph_fps else None,
                         "frames_per_sec_batch1": round(graph_fps_b1, 2) if graph_fps_b1 else None,
                         "note": "same step captured once with torch.cuda.graph and replayed; single GPU, not the headline"},
            "distill": distill,
            "distill_api": distill_api,
            "infer_api": infer_api,
            "distill_strong": distill_strong,
            "stream": stream,
            "roofline": roofline,
            "parity": parity,
            "bf16_variant": bf16_leg,
            "cpu_baseline": cpu,
            "labels_checksum": checksum,
        }
        # The driver's record keeps the line's TAIL: the numbers a reader needs first go last, in one compact object
        def _get(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        result["summary"] = {
            "frames_per_sec": result["value"], "rel_spread": _get(spread, "rel_spread"), "frames_per_sec_batch1": result["frames_per_sec_batch1"],
            "roofline_frac": _get(roofline, "frac"), "roofline_kernel": _get(roofline, "kernel"), "whole_step_frac": _get(roofline, "whole_step_frac"),
            "batch1_frac": _get(roofline, "batch1", "frac"),
            "distill_ms_per_step": _get(distill, "ms_per_step"), "distill_roofline_frac": _get(distill, "roofline", "frac"),
            "distill_api_iterations_per_sec": _get(distill_api, "iterations_per_sec"), "infer_api_ms_per_call": _get(infer_api, "ms_per_call"),
            "stream_realtime_factor_per_video": _get(stream, "realtime_factor_per_video"),
            "stream_realtime_factor_pipelined": _get(stream, "realtime_factor_pipelined"),
            "parity_logits_rel": _get(parity, "timed_call", "logits_max_rel_err") or _get(parity, "logits_max_rel_err"),
            "cpu_baseline_frames_per_sec": _get(cpu, "value"),
            "rccl_ranks": result["rccl_ranks"], "collective_ms_per_step": _get(distill, "collective_ms_per_step"),
            "distill_strong_ms_per_step": _get(distill_strong, "ms_per_step"),
            "distill_strong_collective_ms_per_step": _get(distill_strong, "collective_ms_per_step")}
        if shared:
            result["gpu_sharing"] = {"ranks": world, "gpus": int(shared), "backend": backend,
                                     "note": "fewer GPUs than ranks: a plumbing check of the multi-rank path, NOT a scaling measurement"}
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the last thing on stdout: RCCL's banner sits in the C stdio buffer until it is flushed
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
