"""Edge/server scheduler around the hot path — the *intended* semantics of the reference's run.py.

Same flags (run.py:18-69), same call order on ``SemanticNetwork`` (construct -> save initial frozen model; per
training event: [phi-score / ASR / ATR] -> ``restore_initial`` -> ``train_with_deque`` -> delta accounting ->
``save_to_frozen_graph``; edge: reload at every event time, ``predict_with_metric`` per frame), same output files
(``*_fps_client.npy``, ``*_bw_uplink.npy``, ``*_bw_downlink.npy``, ``*_model_update_times.npy``, ``*_update.txt``,
``*_loss.npy``, ``*_mioucats.npy``, ``*_mious.npy``, ``*_mioumems.npy``).  The reference file does not run as
committed; the defects listed in SURVEY.md Appendix D are resolved towards their evident intent:
  * sampling and training fire ONCE per matching second (reference: once per frame of that second);
  * ``label_memory.append`` (reference ``extend`` pushes rows);
  * first training at ceil(100 / train_period) * train_period seconds, then every ``train_period`` (int range);
  * uplink sampling follows the reference by default (``--sampling reference``): ``send_rate = sampling_period / fps``
    (run.py:115; 1.0 at the defaults 30 / 30) is handed to ``choose_frames`` as the FRACTION of the bucket (run.py:175), so at
    the defaults every bucketed frame is uploaded and the replay memory of ``memory_len / sampling_period * fps`` entries
    spans a few seconds; ASR moves it inside [0.1, 1] (run.py:287-288).  ``--sampling per_second`` is the evident intent of
    the flag names instead: ``send_rate`` counts frames per SECOND (``fps / sampling_period`` at the start, started inside
    ASR's [0.1, 1] when ASR is on), ``choose_frames`` receives ``send_rate / fps``, and the replay memory spans
    ``memory_len`` seconds.  ``*_fps_client.npy``, the uplink byte counts, ``*_update.txt`` totals and the replay-memory
    contents (hence the fine-tuned models) differ between the two settings; INTEGRATION.md lists them;
  * samples are uploaded every ``train_period`` seconds (the reference's last ``train_model`` argument, run.py:600-601);
  * a training event that finds the replay memory empty still publishes the current model for that time, so that the edge
    has a model to load (the reference would fail inside ``mini_batch``).
Out of scope (networking emulation / reporting): H.264 uplink through ffmpeg (``--compress_uplink``
is rejected), PNG-exact uplink byte counts (zlib-deflated frame size is logged instead), the matplotlib plots.
Video comes from a ``FrameSource``: there is no OpenCV here, so ``--input_video`` is either
``synthetic:<NUM>-<name>[:seconds=S][:fps=F]`` (procedural clip) or a directory holding
``frame_%06d.npy`` (RGB uint8) and ``gt_%06d.npy`` files.
"""
from __future__ import annotations

import argparse
import gzip
import os
import sys
import time
import zlib
from collections import deque
from typing import List, Optional, Tuple

import numpy as np

from .exp_configs import class_weights, coco_class_converter, is_coco, test_length
from .semantic_network import SemanticNetwork
from .synth import SyntheticVideo
from .utils import calculate_miou, choose_frames, resize_linear, resize_nearest, string_class_iou


# ----------------------------------------------------------------------------------------------------------- flags
def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="AMS edge/server emulation on MI355X")
    req = dict(required=True)
    p.add_argument("--input_video", **req, help="synthetic:<NUM>-<name>[:seconds=S][:fps=F] or a frame directory NUM-NAME")
    p.add_argument("--gt_video", default="", help="directory of gt_%%06d.npy labels (unused for synthetic input)")
    p.add_argument("--student_checkpoint", **req, help="path prefix of <prefix>.npy, or 'synthetic[:seed]'")
    p.add_argument("--output_dir", **req)
    p.add_argument("--gpu", default="0")
    p.add_argument("--initial_fill", action="store_true")
    p.add_argument("--memory_len", type=int, default=250)
    p.add_argument("--batch_size", type=int, default=10)
    p.add_argument("--iter", type=int, default=200)
    p.add_argument("--height", type=int, default=256)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--send_period", type=int, default=30)
    p.add_argument("--train_period", type=int, default=10)
    p.add_argument("--only_results", action="store_true")
    p.add_argument("--compress_uplink", action="store_true")
    p.add_argument("--no_restore", action="store_true")
    p.add_argument("--save_pic", action="store_true")
    p.add_argument("--gpu_ingest", action="store_true",
                   help="resize / BGR->RGB of frames and labels on the GPU (extra flag; reference: cv2 on the host)")
    p.add_argument("--enable_ASR", action="store_true")
    p.add_argument("--enable_ATR", action="store_true")
    p.add_argument("--train_strategy", default="full_model",
                   choices=["full_model", "coord_desc_auto", "coord_desc_last", "coord_desc_first", "coord_desc_both",
                            "coord_desc_rand"])
    p.add_argument("--coord_fraction", default="0.1", choices=["0.1", "0.05", "0.2", "0.01"])
    p.add_argument("--mode", **req, choices=["simple", "pretrained", "horizon", "early"])
    p.add_argument("--early_cutoff_time", type=int, default=60)
    # additions (not in the reference): make short synthetic runs possible
    p.add_argument("--length", type=int, default=None, help="override exp_configs.test_length (seconds)")
    p.add_argument("--first_train_time", type=int, default=None, help="override ceil(100/train_period)*train_period")
    p.add_argument("--edge_pipeline", type=int, default=1, choices=[1, 2, 3, 4],
                   help="edge: frames per inference pass (extra flag; 1 = the reference's synchronous per-frame call).  With n >= 2 the edge "
                        "holds n frames and labels them in ONE pass (a one-frame pass leaves most of the chip idle), while the previous pass's "
                        "results are being consumed; per-frame outputs are identical")
    p.add_argument("--sampling", default="reference", choices=["reference", "per_second"],
                   help="uplink sampling: 'reference' = run.py:115/175 (send_rate = send_period / fps is the fraction of the bucket "
                        "that is uploaded), 'per_second' = send_rate counts frames per second (fps / send_period)")
    p.add_argument("--horizon_k1s", default="16,32,64,128,256,512", help="horizon mode: training-window lengths in seconds (reference: hard-coded)")
    p.add_argument("--horizon_k2", type=int, default=256, help="horizon mode: evaluation window in seconds (reference: 256)")
    p.add_argument("--horizon_points", type=int, default=3, help="horizon mode: number of evaluation points (reference: 3)")
    return p


# ----------------------------------------------------------------------------------------------------------- video
class FrameSource:
    """fps, number of frames, and (RGB uint8 frame, uint8 teacher label) by absolute frame index."""
    fps: int

    def __len__(self) -> int:  # pragma: no cover
        raise NotImplementedError

    def read(self, i: int) -> Tuple[np.ndarray, np.ndarray]:  # pragma: no cover
        raise NotImplementedError


class SyntheticSource(FrameSource):
    def __init__(self, exp_num: int, height: int, seconds: int, fps: int):
        self.fps = fps
        cw = class_weights(exp_num)
        self.video = SyntheticVideo(height, seconds * fps, np.where(cw.reshape(-1) == 1)[0], num_classes=cw.shape[0], seed=exp_num)
        self.n = seconds * fps

    def __len__(self):
        return self.n

    def read(self, i):
        return self.video.frame(i)


class DirectorySource(FrameSource):
    def __init__(self, frames_dir: str, gt_dir: str, fps: int = 30):
        self.frames_dir, self.gt_dir, self.fps = frames_dir, gt_dir or frames_dir, fps
        self.n = len([f for f in os.listdir(frames_dir) if f.startswith("frame_") and f.endswith(".npy")])

    def __len__(self):
        return self.n

    def read(self, i):
        return (np.load(os.path.join(self.frames_dir, "frame_%06d.npy" % i)),
                np.load(os.path.join(self.gt_dir, "gt_%06d.npy" % i)))


def open_source(flags) -> Tuple[FrameSource, int]:
    spec = flags.input_video
    if spec.startswith("synthetic:"):
        parts = spec.split(":")
        vid_num = int(parts[1].split("-")[0])
        opts = dict(kv.split("=") for kv in parts[2:])
        seconds = int(opts.get("seconds", flags.length or test_length(vid_num)))
        return SyntheticSource(vid_num, flags.height, seconds, int(opts.get("fps", 30))), vid_num
    vid_num = int(os.path.basename(spec.rstrip("/")).split("-")[0])
    return DirectorySource(spec, flags.gt_video), vid_num


def _to_size(frame: np.ndarray, label: np.ndarray, size: List[int], ingest=None):
    """cv2.resize(frame, (2H, H)) [bilinear] and cv2.resize(label, ..., INTER_NEAREST) (run.py:179-183, :415-421).

    ``ingest`` (an ``ams_amd.ingest.FrameIngest``, flag ``--gpu_ingest``) does both on the device: the raw uint8 frame is
    what crosses PCIe and the results stay there (``SemanticNetwork`` takes device tensors); frames that already have the
    network's size pass through untouched either way."""
    if frame.shape[:2] != (size[0], size[1]):
        frame = ingest.frame(frame, size[0], size[1]) if ingest is not None else resize_linear(frame, size[1], size[0])
    if label.shape[:2] != (size[0], size[1]):
        label = ingest.label(label, size[0], size[1]) if ingest is not None else resize_nearest(label, size[1], size[0])
    return frame, label


def _batch1(x):
    """np.expand_dims(x, 0) for host arrays and device tensors alike."""
    return x.unsqueeze(0) if hasattr(x, "unsqueeze") else np.expand_dims(x, axis=0)


def _host(x) -> np.ndarray:
    return x.cpu().numpy() if hasattr(x, "cpu") else x


class Context:
    def __init__(self, flags, network_cls=None):
        self.flags = flags
        # the class behind the SemanticNetwork boundary: the HIP-backed one unless a caller injects another with the same
        # surface (tests/golden/make_scheduler_fixture.py pins this loop with a CPU stand-in)
        self.network_cls = network_cls or SemanticNetwork
        self.size = [flags.height, flags.height * 2]
        self.source, self.vid_num = open_source(flags)
        self.length = flags.length or (len(self.source) // self.source.fps)
        self.ingest = None
        if getattr(flags, "gpu_ingest", False):
            from .ingest import FrameIngest
            self.ingest = FrameIngest("cuda:%s" % flags.gpu)
        ck = flags.student_checkpoint
        self.initial_variables = None
        if ck.startswith("synthetic"):
            from .spec import build_spec
            from .weights import synthetic_weights
            seed = int(ck.split(":")[1]) if ":" in ck else 0
            self.initial_variables = synthetic_weights(build_spec(class_weights(self.vid_num).shape[0]), seed)

    def save_dir(self, prepend: str) -> str:
        video = self.flags.input_video.replace(":", "_").split("/")[-1]
        ck = self.flags.student_checkpoint.replace(":", "_").split("/")
        return os.path.join(self.flags.output_dir, "%s_%s_%s_%d" % (prepend, video, ck[-2] if len(ck) > 1 else ck[-1],
                                                                    self.flags.height))


def print_process(str_log, curr_time):
    print("Process [current time: %d]: " % curr_time, str_log)


# ----------------------------------------------------------------------------------------------------------- server
def train_model(ctx: Context, train_start, train_end, sampling_period, gpu_id, run_label, gt_path, exp_num, save_range,
                sample_send_period):
    """Server side: collect sampled frames in [train_start, train_end), fine-tune at the times in save_range and publish
    a frozen model after each (reference run.py:78-361)."""
    FLAGS = ctx.flags
    assert train_end - train_start != 0, "There should be at least one set of data points"
    assert not FLAGS.compress_uplink, "H.264 uplink emulation (ffmpeg) is out of scope for this build"
    save_range = list(save_range)
    fps = ctx.source.fps
    train_end_frame = min(train_end * fps, len(ctx.source))
    i = train_start * fps
    update_count = 0
    per_second = getattr(FLAGS, "sampling", "reference") == "per_second"
    if per_second:
        send_rate = min(float(fps), fps / float(sampling_period))  # frames per second uploaded (module docstring)
        if FLAGS.enable_ASR:
            send_rate = float(np.clip(send_rate, 0.1, 1))          # start inside ASR's range: its first update must not jump
    else:
        send_rate = sampling_period / fps                          # reference run.py:115: the fraction of the bucket
    sample_per_period, up_bw_per_period, down_bw_per_period = [], [], []
    frame_label_bucket = []
    num_unseen_frames = 0
    model_save_times = [0]
    train_period_reset = train_period_current = (save_range[2] - save_range[1]) if len(save_range) > 2 else FLAGS.train_period
    send_rate_deq = deque(maxlen=5)
    hibernate = False
    map_coco = coco_class_converter() if is_coco(exp_num) else None
    mem = max(1, int(FLAGS.memory_len / sampling_period * fps))
    frame_memory, label_memory = deque(maxlen=mem), deque(maxlen=mem)

    semantic_network = ctx.network_cls(meta_dir=FLAGS.student_checkpoint, class_weights_exp=class_weights(exp_num),
                                       height=FLAGS.height, gpu_id=gpu_id, scale=[1], mini_batch_size=FLAGS.batch_size,
                                       lr=FLAGS.lr, mem_frac=1, coord_frac=float(FLAGS.coord_fraction),
                                       train_biases_only=False, regularize=False,
                                       masked_gradients=FLAGS.train_strategy not in ['full_model'],
                                       cross_miou_compat=FLAGS.enable_ASR, initial_variables=ctx.initial_variables)
    save_dir = ctx.save_dir(run_label + "_%d" % train_start)
    semantic_network.save_to_frozen_graph(save_dir + "_final")
    print_process("Saved model to %s_final.pb" % save_dir, 0)
    train_ms = []
    control_log = []          # per training event: (second, mean phi-score or nan, send_rate, train_period_current, hibernating)

    while i < train_end_frame:
        frame, gt = ctx.source.read(i)
        frame_label_bucket.append((frame, gt))
        i += 1
        if i % fps != 0:
            continue                                   # events are evaluated once per elapsed second
        second = i // fps
        if second % (5) == 0:
            print_process("%d seconds elapsed" % second, second)

        if second % sample_send_period == 0:
            frames_chosen, labels_chosen = choose_frames(frame_label_bucket, min(1.0, send_rate / fps if per_second else send_rate))
            size_images = 0.0
            for fr, label in zip(frames_chosen, labels_chosen):
                fr, label_resized = _to_size(fr, label, ctx.size, ctx.ingest)
                fr, label_resized = _host(fr), _host(label_resized)       # the replay memory lives on the host
                if map_coco is not None:
                    label_resized = map_coco[label_resized]
                frame_memory.append(fr)
                label_memory.append(label_resized)
                size_images += len(zlib.compress(fr.tobytes(), 6)) / 1024     # stand-in for the PNG size
            frame_label_bucket.clear()
            sample_per_period.append(len(frames_chosen))
            num_unseen_frames += len(frames_chosen)
            up_bw_per_period.append(size_images * 8)

        if second in save_range and len(frame_memory) == 0:
            # nothing has been uploaded yet (e.g. a horizon window shorter than the upload period): the model of this event
            # time is the current one, published unchanged
            print_process("No samples in memory at %d s: publishing the current model unchanged" % second, second)
            save_dir = ctx.save_dir(run_label + "_%d" % second)
            semantic_network.save_to_frozen_graph(save_dir + "_final")
            model_save_times.append(float(second))
        elif second in save_range:
            phi = float("nan")
            if FLAGS.enable_ASR and len(label_memory) > 1:
                # phi-score over the frames that arrived since the last update -> sampling rate (run.py:279-290)
                i_start = max(0, len(label_memory) - num_unseen_frames - 1)
                cross = [semantic_network.calc_cross_miou(np.array([label_memory[k], label_memory[k + 1]]))[2]
                         for k in range(i_start, len(label_memory) - 1)]
                if cross:
                    phi = float(np.mean(cross))
                    send_rate = float(np.clip(send_rate - 0.2 * np.tanh((np.mean(cross) - 0.6) * 20), 0.1, 1))
                    send_rate_deq.append(send_rate)
                    print_process("Send rate updated to %.2f" % send_rate, second)
                num_unseen_frames = 0
            if FLAGS.enable_ATR and len(send_rate_deq) > 0:
                if np.mean(list(send_rate_deq)) < 0.25:
                    hibernate = True
                if np.mean(list(send_rate_deq)) > 0.35 and hibernate:
                    hibernate = False
                    train_period_current = train_period_reset
                if hibernate:
                    train_period_current = min(train_period_current + 2, 6 * train_period_reset)
                idx = save_range.index(second)
                save_range = save_range[:idx] + list(range(second, train_end, train_period_current))
            control_log.append((second, phi, send_rate, train_period_current, int(hibernate)))

            if not FLAGS.no_restore:
                semantic_network.restore_initial()
            t1 = time.time()
            semantic_network.train_with_deque(frame_memory, label_memory, FLAGS.iter, FLAGS.train_strategy)
            train_ms.append(1000 * (time.time() - t1))
            print("Training for %d iterations took %d ms!!!" % (FLAGS.iter, train_ms[-1]))
            # model delta on the downlink: packed mask bits + masked parameters as fp16, gzip -9 (run.py:316-336)
            payload = semantic_network.delta_payload()          # value part gathered + cast to fp16 on the device
            full_size = sum(val.size for val in semantic_network.curr_mask)
            with open(save_dir + '_mask.dat', 'wb') as f:
                f.write(payload)
            with gzip.open(save_dir + '_mask.dat.gz', 'wb', compresslevel=9) as f:
                f.write(payload)
            curr_update = os.path.getsize(save_dir + '_mask.dat.gz') * 8
            down_bw_per_period.append(curr_update)
            update_count += 1
            print("Full size of model is %d; update is %.1f Kbit" % (full_size, curr_update / 1024))
            save_dir = ctx.save_dir(run_label + "_%d" % second)
            semantic_network.save_to_frozen_graph(save_dir + "_final")
            print_process("Saved model to %s_final.pb" % save_dir, second)
            model_save_times.append(float(second))

    semantic_network.close_model()
    final_save_dir = ctx.save_dir(run_label + "_results")
    np.save(final_save_dir + '_fps_client.npy', sample_per_period)
    np.save(final_save_dir + '_bw_uplink.npy', up_bw_per_period)
    np.save(final_save_dir + '_bw_downlink.npy', down_bw_per_period)
    np.save(final_save_dir + '_model_update_times.npy', model_save_times)
    np.save(final_save_dir + '_train_ms.npy', train_ms)
    np.save(final_save_dir + '_control.npy', np.asarray(control_log, dtype=np.float64).reshape(-1, 5))
    with open(final_save_dir + '_update.txt', 'w') as f:
        f.write("%d\n%d\n%d\n%d\n%d" % (sum(down_bw_per_period), sum(up_bw_per_period), update_count,
                                        train_end - train_start, sum(sample_per_period)))
    return model_save_times


# ----------------------------------------------------------------------------------------------------------- edge
def infer_output(ctx: Context, inf_start, inf_end, gpu_id, run_label, gt_path, exp_num, load_range):
    """Edge side: label every frame in [inf_start, inf_end) with the newest published model (run.py:364-461)."""
    FLAGS = ctx.flags
    assert inf_end - inf_start != 0, "There should be at least one set of data points"
    fps = ctx.source.fps
    inf_end_frame = min(inf_end * fps, len(ctx.source))
    i = inf_start * fps
    semantic_network = None
    confusion_matrix_memory = deque(maxlen=10 * fps)
    loss_s, miou_cats, miou_s, miou_mem_s = [], [], [], []
    final_save_dir = ctx.save_dir(run_label + "_results")
    load_times = set(float(t) for t in load_range)
    t_infer = 0.0
    depth = int(getattr(FLAGS, "edge_pipeline", 1))
    in_flight = deque()               # tickets of submitted frames (depth >= 2), oldest first

    def record(result, n_done):
        _labels, conf_mat_, _, miou_, loss_ = result
        loss_s.append(loss_)
        miou_cats.append(np.array(conf_mat_))
        miou_s.append(miou_)
        confusion_matrix_memory.append(conf_mat_)
        miou_mem_s.append(np.nanmean(calculate_miou(np.sum(list(confusion_matrix_memory), axis=0), nan=True)))
        if n_done % fps == 0:
            miou = np.nanmean(calculate_miou(np.sum(miou_cats[-fps:], axis=0), nan=True))
            print_process("miou at %03d secs: %.1f%%" % (n_done / fps, float(miou) * 100), n_done / fps)

    done = inf_start * fps
    while i < inf_end_frame:
        if i / fps in load_times:
            while in_flight:                           # the frames still in flight belong to the model that is about to be replaced
                t0 = time.time()
                res = semantic_network.collect(in_flight.popleft())
                t_infer += time.time() - t0
                done += 1
                record(res, done)
            save_dir = ctx.save_dir(run_label + "_%d" % (i // fps))
            if semantic_network is not None:
                semantic_network.close_model()
            kw = {"pipeline_depth": depth} if depth > 1 else {}
            semantic_network = ctx.network_cls(meta_dir=save_dir + "_final", class_weights_exp=class_weights(exp_num),
                                               height=FLAGS.height, gpu_id=gpu_id, mem_frac=1, frozen=True, **kw)
        frame, gt_frame = _to_size(*ctx.source.read(i), ctx.size, ctx.ingest)
        t0 = time.time()
        if depth > 1:
            in_flight.append(semantic_network.predict_with_metric_async(_batch1(frame), _batch1(gt_frame)))
            # up to two passes of `depth` frames in flight: the one on the GPU and the one being filled
            res = semantic_network.collect(in_flight.popleft()) if len(in_flight) >= 2 * depth else None
        else:
            res = semantic_network.predict_with_metric(_batch1(frame), _batch1(gt_frame))
        t_infer += time.time() - t0
        i += 1
        if res is not None:
            done += 1
            record(res, done)
    while in_flight:
        t0 = time.time()
        res = semantic_network.collect(in_flight.popleft())
        t_infer += time.time() - t0
        done += 1
        record(res, done)
    np.save('%s_loss.npy' % final_save_dir, loss_s)
    np.save('%s_mioucats.npy' % final_save_dir, miou_cats)
    np.save('%s_mious.npy' % final_save_dir, miou_s)
    np.save('%s_mioumems.npy' % final_save_dir, miou_mem_s)
    if semantic_network is not None:
        semantic_network.close_model()
    n = max(1, inf_end_frame - inf_start * fps)
    return {"frames": n, "frames_per_sec": n / max(t_infer, 1e-9), "mean_miou": float(np.nanmean(miou_s))}


def event_times(flags, length: int) -> List[int]:
    """[0] + first training at ceil(100/train_period)*train_period, then every train_period (run.py:594-598)."""
    first = flags.first_train_time if flags.first_train_time is not None else int(np.ceil(100 / flags.train_period) * flags.train_period)
    return [0] + [t for t in range(first, length, flags.train_period)
                  if t == 0 or t >= flags.memory_len or not flags.initial_fill]


def main(argv: Optional[List[str]] = None, network_cls=None):
    flags = build_parser().parse_args(argv)
    assert not flags.enable_ATR or flags.enable_ASR, 'ASR must be enabled for ATR to work'
    assert not flags.enable_ASR or flags.mode == 'simple', 'ASR can only be used in simple mode'
    assert not flags.enable_ATR or flags.mode == 'simple', 'ATR can only be used in simple mode'
    os.makedirs(flags.output_dir, exist_ok=True)
    ctx = Context(flags, network_cls)
    vid_num, length = ctx.vid_num, ctx.length
    summary = None
    if flags.mode == 'simple':
        run_label = "%d__%d_tp%d_f%d" % (0, length, flags.train_period, flags.send_period)
        events = event_times(flags, length)
        if not flags.only_results:
            events = train_model(ctx, 0, length, flags.send_period, flags.gpu, run_label, flags.gt_video, vid_num, events,
                                 flags.train_period)          # the times at which a model was actually published
            summary = infer_output(ctx, 0, length, flags.gpu, run_label, flags.gt_video, vid_num, events)
    elif flags.mode == 'early':
        run_label = "early%d_f%d" % (flags.early_cutoff_time, flags.send_period)
        events = [0, flags.early_cutoff_time]
        if not flags.only_results:
            events = train_model(ctx, 0, flags.early_cutoff_time, flags.send_period, flags.gpu, run_label, flags.gt_video,
                                 vid_num, events, flags.train_period)
            summary = infer_output(ctx, 0, length, flags.gpu, run_label, flags.gt_video, vid_num, events)
    elif flags.mode == 'pretrained':
        run_label = "pretrained"
        train_model(ctx, 0, 1, flags.send_period, flags.gpu, run_label, flags.gt_video, vid_num, [0], flags.train_period)
        summary = infer_output(ctx, 0, length, flags.gpu, run_label, flags.gt_video, vid_num, [0])
    else:  # horizon: retrain on [t-k1, t), evaluate on [t, t+k2)
        k1s, k2 = [int(k) for k in flags.horizon_k1s.split(",")], flags.horizon_k2
        number_of_points = flags.horizon_points
        step = (length - k2 - k1s[-1]) // max(number_of_points - 1, 1)
        assert step > 0, "video too short for horizon mode"
        # the un-adapted model over the whole video first, as the reference does (run.py:617-619)
        train_model(ctx, 0, 1, flags.send_period, flags.gpu, "pretrained", flags.gt_video, vid_num, [0], flags.train_period)
        infer_output(ctx, 0, length, flags.gpu, "pretrained", flags.gt_video, vid_num, [0])
        for p in range(number_of_points):
            t = k1s[-1] + p * step
            for k1 in k1s:
                run_label = "%d__%d__%d_f%d" % (t - k1, t, t + k2, flags.send_period)
                train_model(ctx, t - k1, t, flags.send_period, flags.gpu, run_label, flags.gt_video, vid_num, [t], flags.train_period)
                summary = infer_output(ctx, t, t + k2, flags.gpu, run_label, flags.gt_video, vid_num, [t])
    print("Process [Main]:", "Done!!!", summary if summary else "")
    return summary


if __name__ == "__main__":
    main(sys.argv[1:])
