#!/usr/bin/env python3
"""Benchmark of the hot path: ResNet-50 feature extraction, frames/s on synthetic 224x224x3 frames.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: BATCH (256) fp32 NCHW frames per GPU, already
resident in HBM, -> (BATCH, 2048) fp32 features (= backbone(x).flatten(1),
/root/reference/src/preprocess_resnet_features.py:296); for N > 1 each rank owns its own frames
(weak scaling) and the features return to rank 0 with one RCCL gather per step.
Prints ONE JSON line on rank 0 (contract in the task statement), including:
  roofline     — the implicit-GEMM conv kernel class: algorithmic FLOPs / HIP-event time of those launches,
                 measured live on the launch stream.
  cpu_baseline — the CPU restatement of the reference's fp32 path (oracle/) timed on the host cores.
  preheat      — an untimed >= 1 s run of the same step right before the timed region (steady clocks), with its rate.
  checked      — the timed output is finite and frames 0 / B-1 equal the same frames run alone (outside the timed region).
  fp16_b256, fp8_b512 — SECONDARY measurements appended after the headline one (N = 1 only): the mode that meets
                 north_star's 1e-3 and BASELINE configs[4]; each with frames/s, its igemm class against its MFMA peak, and its
                 accuracy against the oracle's emulation and fp64 view (computed inside the cpu_baseline leg).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

BATCH = 256
GFLOP_PER_FRAME = 8.174272512          # 2 x 4,087,136,256 MAC, 53 convs
MFMA_BF16_PEAK_TFLOPS = 2500.0         # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0          # dense fp8 on the scaled K = 128 MFMA, same guide (--precision fp8 prices its igemm class against this)
HBM_PEAK_GBPS = 8000.0                  # HBM3E, same guide
HBM_PEAK_GBS = 8000.0


def host_cores() -> int:
    """CPU threads this process may really use: min(affinity, cgroup quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:                                            # cgroup v2 quota, e.g. "1600000 100000"
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    env = os.environ.get("R50_CPU_THREADS")
    if env:
        n = int(env)
    return max(1, min(n, 64))


def cpu_baseline(sample_frames: int = 64, reps: int = 5, accuracy_of: dict = None) -> dict:
    """Reference's CPU numerics (fp32, autocast off, :239-241) via the oracle restatement, on the host cores.
    The oracle is test infrastructure: this leg is the only place bench.py touches it.  ``accuracy_of``: per secondary
    precision, the device features of the first 2 synthetic frames -- checked here against the oracle's emulation of that
    precision and against its fp64 reference view (per-frame rel-L2)."""
    from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict
    from oracle import resnet50_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synthetic_state_dict(0)
    x = synthetic_frames(sample_frames, seed=1234)
    O.forward_reference(sd, x)                      # warm-up
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        O.forward_reference(sd, x)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    out = {"value": sample_frames / med, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{sample_frames} frames x {reps} timed forwards (median), fp32 torch.nn.functional restatement "
                     f"of torchvision resnet50[:-1], torch {torch.__version__}"}
    if accuracy_of:
        x2 = x[:2]
        ref64 = O.forward_reference(sd, x2, dtype=torch.float64).float()
        acc = {}
        for prec, item in accuracy_of.items():
            feats = item["feats"]
            fmt = item.get("precision", prec)
            if fmt == "fp8":
                emu = O.forward_fp8_emulated(sd, x2, item["scales"])
            else:
                emu = O.forward_bf16_emulated(sd, x2, fused_ds=True, fmt=("fp16" if fmt == "fp16" else "bf16"))
            acc[prec] = {"frames": [0, 1], "metric": "max over the frames of ||device - oracle||_2 / ||oracle||_2 on the 2048-d feature vector",
                         "rel_l2_vs_emulation": float(O.per_row_rel_l2(feats, emu).max()),
                         "rel_l2_vs_fp64_reference": float(O.per_row_rel_l2(feats, ref64).max())}
        out["accuracy"] = acc
    return out


CLASS_LABELS = {
    "igemm": "igemm_bf16_kernel + igemm_ws_kernel + gemm8p_kernel + conv3x3_xres_kernel + conv3x3_s2_kernel: every conv that is a launch of its own",
    "bneck_block2": "bneck_block1_kernel (layer1.0 with its downsample conv, layer1.1, layer1.2) + bneck_block2_kernel (layer2.1-.3): a bottleneck body per launch",
    "bneck_tail3": "bneck_tail3p_kernel: chained layer3 tails (conv3 + identity + ReLU + next conv1; layer3.5: conv3 alone)",
    "bneck_catchain": "bneck_catchain_kernel: layer2.0 conv3 + downsample + ReLU chained with layer2.1.conv1",
    "bneck_tail": "bneck_tail_kernel / bneck_tail2_kernel: fused layer1 / layer2 tails (not on the default path)",
    "conv1": "stem_fused3_kernel: frames -> conv1 7x7 s2 + bn + ReLU + maxpool + layer1.0.conv1",
    "avgpool": "avgpool_kernel",
}
TRAFFIC_KEYS = {"bneck_block2": "bneck_block"}       # class name in the committed PMC summary where it differs


def roofline_object(prof: dict, steps: int, precision: str, traffic: dict = None, traffic_source: str = None) -> dict:
    """The `roofline` object of the bench line from the library's per-class HIP-event profile (one lane, `steps` forwards).

    The object describes the DOMINANT class = the one with the largest ms_per_step; `bound` is whichever of its two roofline fractions
    is larger (these kernels stream and multiply at once, so both are given); every other class rides in `classes` with the same fields.
    `achieved` = algorithmic flops (or layer-wise algorithmic bytes) of the class's launches / their summed event durations."""
    mfma_peak = MFMA_FP8_PEAK_TFLOPS if precision == "fp8" else MFMA_BF16_PEAK_TFLOPS
    traffic = traffic or {}

    def describe(name: str, p: dict) -> dict:
        sec = p["ms"] * 1e-3
        tf = p["flops"] / sec / 1e12
        gb = p["bytes"] / sec / 1e9
        f_m, f_h = tf / mfma_peak, gb / HBM_PEAK_GBPS
        by_mfma = f_m >= f_h
        launches = max(1, p["launches"])
        t = (traffic.get(TRAFFIC_KEYS.get(name, name)) or {}).get("hbm_bytes_per_launch")
        d = {"class": name, "kernel": CLASS_LABELS.get(name, name), "bound": "mfma" if by_mfma else "hbm",
             "achieved": tf if by_mfma else gb, "peak": mfma_peak if by_mfma else HBM_PEAK_GBPS, "unit": "TFLOP/s" if by_mfma else "GB/s",
             "frac": f_m if by_mfma else f_h, "traffic": t, "traffic_source": traffic_source if t is not None else None,
             "tflops": tf, "frac_of_mfma_peak": f_m, "GBps": gb, "frac_of_hbm_peak": f_h,
             "launches_per_step": round(p["launches"] / max(1, steps)), "ms_per_step": p["ms"] / max(1, steps),
             "avg_launch_us": 1e3 * p["ms"] / launches, "flops_per_launch": p["flops"] / launches, "algorithmic_bytes": p["bytes"] / launches,
             "traffic_over_algorithmic": (t / (p["bytes"] / launches)) if (t and p["bytes"]) else None}
        return d

    live = {k: v for k, v in prof.items() if k in CLASS_LABELS and v.get("launches") and v.get("ms", 0) > 0}
    if not live:
        return None
    top = max(live, key=lambda k: live[k]["ms"])
    out = describe(top, live[top])
    out["classes"] = {k: describe(k, v) for k, v in live.items() if k != top}
    out["measured_on"] = "lane 0 alone (one batch at a time), HIP events on the launch stream around every launch of a second pass over the same steps"
    out["traffic_note"] = ("FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per full-batch launch, batch 256 (static: see traffic_source); "
                           "algorithmic_bytes = layer-wise in + out (+ residual) bytes of the same launches, from this run")
    return out


def timed_steps(step, fence, steps: int, warmup: int, preheat_s: float, frames_per_step: int, sync_max=None):
    """W untimed warm-up steps, an untimed pre-heat of the same step lasting >= preheat_s (clocks / power at their
    steady state before the clock starts), then EXACTLY `steps` timed steps between two fences.
    ``sync_max(x)``: MAX over the ranks of a float (multi-rank runs: every rank must run the SAME number of pre-heat steps,
    a step contains a collective)."""
    for _ in range(warmup):
        step()
    fence()
    pre_rate = None
    if preheat_s > 0:
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        fence()
        dt = time.perf_counter() - t0
        if sync_max is not None:
            dt = sync_max(dt)
        more = max(0, int(((preheat_s - dt) / (dt / 10)) + 0.999))
        more = min(more, 100000)
        for _ in range(more):
            step()
        fence()
        pre_rate = (10 + more) * frames_per_step / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0, pre_rate


def secondary_mode(precision: str, batch: int, steps: int, warmup: int, preheat_s: float, dev, sd, n_lanes: int = 1) -> dict:
    """A secondary, clearly labelled measurement after the headline one: the same step in another precision / batch
    (fp16 at batch 256 = the mode inside north_star's 1e-3; fp8 at batch 512 = BASELINE configs[4])."""
    from implementation_phd_lab_vision_amd.backbone import BackboneLanes, ResNet50Backbone
    from implementation_phd_lab_vision_amd.weights import synthetic_frames
    lanes = lanes_obj = None
    if n_lanes > 1:                       # the headline's execution mode: steps dealt round robin over the lanes; lane 0 alone for the checks / profile
        lanes = BackboneLanes(lanes=n_lanes, state_dict=sd, max_batch=batch, precision=precision).to(dev).eval()
        bb = lanes.lane0
    else:
        bb = ResNet50Backbone(state_dict=sd, max_batch=batch, precision=precision).to(dev).eval()
    try:
        x = synthetic_frames(batch, seed=1234).to(dev)
        lanes_mode = "one lane"
        if lanes is not None:
            lanes.tune(x, min_gain=1.01)   # falls back to ONE lane when the lanes cost time (fp8 at batch 512: launches that leave no tail to fill)
            n_lanes, lanes_mode = lanes.active_lanes, lanes.tune_mode
        feats_l = [torch.empty((batch, 2048), dtype=torch.float32, device=dev) for _ in range(max(1, n_lanes))]
        feats = feats_l[0]
        state = {"k": 0}

        def step():
            if lanes is None:
                bb.features(x, out=feats)
            else:
                lanes.submit(x, out=feats_l[state["k"] % n_lanes])
                state["k"] += 1

        def fence():
            torch.cuda.synchronize(dev)

        elapsed, pre = timed_steps(step, fence, steps, warmup, preheat_s, batch)
        lanes_equal = all(bool(torch.equal(feats_l[0], f)) for f in feats_l[1:]) if lanes is not None else None
        single = None
        if lanes is not None:
            fell_back = lanes.active_lanes == 1
            lanes, lanes_obj = None, lanes        # from here on: lane 0 alone (the single-lane figure, the batch-2 check, the event profile)
            if fell_back:                         # the timed region above already ran one batch at a time: it IS the single-lane figure
                single = {"value": batch * steps / elapsed, "unit": "frames/s", "ms_per_step": 1e3 * elapsed / steps, "note": "same measurement as `value` (one lane)"}
            else:
                el1, _ = timed_steps(step, fence, steps, 1, min(preheat_s, 0.3), batch)
                single = {"value": batch * steps / el1, "unit": "frames/s", "ms_per_step": 1e3 * el1 / steps}
        if not bool(torch.isfinite(feats).all()):
            raise SystemExit(f"bench.py: non-finite features in the {precision} secondary run")
        small = bb.features(x[:2].contiguous())
        batch_ok = bool(torch.equal(small, feats[:2]))
        bb.set_option("profile", 1)
        bb.profile_reset()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        prof = bb.profile_collect()
        bb.set_option("profile", 0)
        ig = prof["igemm"]
        peak = MFMA_FP8_PEAK_TFLOPS if precision == "fp8" else MFMA_BF16_PEAK_TFLOPS
        # fp8 mode: layer1's three 3x3 convs run in 16 bits inside the same class; their flops are priced at the fp8 peak too (conservative)
        ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12 if ig["ms"] > 0 else 0.0
        out = {"secondary": True, "precision": precision, "batch": batch, "steps": steps, "lanes": max(1, n_lanes), "lanes_mode": lanes_mode,
               "single_lane": single,
               "lanes_equal": lanes_equal, "value": batch * steps / elapsed,
               "unit": "frames/s", "ms_per_step": 1e3 * elapsed / steps, "preheat_frames_per_s": pre,
               "tflops": batch * steps / elapsed * GFLOP_PER_FRAME / 1e3,
               "frac_of_peak_whole_step": batch * steps / elapsed * GFLOP_PER_FRAME / 1e3 / peak,
               "igemm": {"achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "avg_launch_us": 1e3 * ig["ms"] / max(1, ig["launches"]),
                         "launches_per_step": ig["launches"] / steps},
               "frames_0_1_equal_batch2_run": batch_ok}
        acc_in = {"feats": feats[:2].cpu()}
        if precision == "fp8":
            acc_in["scales"] = list(bb.fp8_scales)
        return out, acc_in
    finally:
        (lanes_obj or lanes or bb).close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="frames per GPU per step (headline config: 256)")
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=2,
                    help="independent batches in flight (backbone.BackboneLanes: one backbone copy + HIP stream per lane, steps dealt round "
                         "robin): the next step's launches fill the CUs that the previous step's launch tails leave idle.  1 = the steps run "
                         "strictly one after the other on one stream (also reported in the line as `single_lane`)")
    ap.add_argument("--streams", type=int, default=0, help="internal streams the batch is split over (0 = library default)")
    ap.add_argument("--precision", choices=["bf16", "fp16", "bf16w2", "fp32x", "fp8"], default="bf16",
                    help="bf16 = headline path; fp32x = fp32-class accuracy mode (bf16 head/tail pairs, 3 products per conv); "
                         "fp8 = BASELINE configs[4] (layer2-4 on the fp8 MFMA; a throughput mode, not the headline)")
    ap.add_argument("--from-host", action="store_true",
                    help="PCIe-inclusive variant (never the headline value): frames start in pinned host memory every step; the "
                         "copy of step k+1 runs on a side stream while step k computes")
    ap.add_argument("--input", choices=["f32", "u8", "video"], default="f32",
                    help="f32 = the reference boundary (normalised fp32 NCHW frames); u8 = resized uint8 crops, normalised in the stem "
                         "kernel; video = decoded 1002x1000 uint8 HWC frames + a 500-pixel crop box: crop + bilinear resize on the device too")
    ap.add_argument("--no-fused-stem", action="store_true", help="A/B: run conv1 / maxpool as separate kernels")
    ap.add_argument("--no-fuse-tail", action="store_true", help="A/B: layer1 conv3 and the next conv1 as two igemm launches")
    ap.add_argument("--no-stem-c1", action="store_true", help="A/B: layer1.0.conv1 as its own igemm launch instead of inside the stem kernel")
    ap.add_argument("--no-ds-cat", action="store_true", help="A/B: downsample conv and conv3 of layer2.0 / 3.0 / 4.0 as separate launches")
    ap.add_argument("--no-overlap-ds", action="store_true", help="A/B: downsample convs on the main stream")
    ap.add_argument("--no-cat-chain", action="store_true", help="A/B: layer2.0's conv3 + downsample launch and layer2.1.conv1 as two igemm launches")
    ap.add_argument("--no-block2", action="store_true", help="A/B: layer2.1-.3 as conv2 launch + fused tail instead of one launch per bottleneck body")
    ap.add_argument("--block1", type=int, default=-1, help="A/B: option fuse_block1 (0 = layer1 as conv2 launch + fused tail, 1 = layer1.1 in one launch, 2 = layer1.2 too, 3 = layer1.0 too: the default)")
    ap.add_argument("--inplace", action="store_true", help="A/B: plain-identity blocks write their output over their input (same bits; measured: no gain)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="A/B: any library option (r50_set_option), e.g. --opt tail3_bp=112")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the batch-2 check forward behind the timed region (PMC passes: every profiled launch is then a full-batch one)")
    ap.add_argument("--dump-layers", default=None,
                    help="write the launches of one forward pass in launch order (name, algorithmic bytes and flops per launch, us) as JSON")
    ap.add_argument("--preheat", type=float, default=1.0,
                    help="seconds of untimed pre-heat of the same step before the timed region (independent of --warmup)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary fp16 (batch 256) and fp8 (batch 512) measurements appended after the headline one")
    ap.add_argument("--stream-frames", type=int, default=0,
                    help="fixed-stream mode (SURVEY 8d): this many frames IN TOTAL (e.g. 163840 = 4096 clips x 40), cut into contiguous "
                         "ranges per rank (strong scaling); --steps is then derived and reported.  0 = weak scaling, --batch per rank per step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")

    from implementation_phd_lab_vision_amd import _lib
    from implementation_phd_lab_vision_amd.backbone import ResNet50Backbone
    from implementation_phd_lab_vision_amd.weights import synthetic_frames, synthetic_state_dict

    # R50_BENCH_REHEARSAL=1: control-flow rehearsal of the N > 1 path on a ONE-GPU box (every rank on cuda:0, gloo collectives on device
    # tensors) -- checks that all ranks run the same sequence of collectives; its numbers mean nothing
    rehearsal = os.environ.get("R50_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    # a process group exists for N > 1, and also for ONE rank started by torchrun (RANK / WORLD_SIZE in the environment): that run
    # executes the multi-rank code path -- RCCL init with device_id, the gather of device tensors, barrier + fence -- on a one-GPU box
    from implementation_phd_lab_vision_amd.distributed import single_rank_group_requested
    in_group = world > 1 or single_rank_group_requested()          # one rank: only under a launcher that set RANK, WORLD_SIZE and MASTER_PORT
    if in_group:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if rank == 0:
        _lib.build_library()
    if dist is not None:
        dist.barrier()

    sd = synthetic_state_dict(0)
    if args.from_host or args.input != "f32":
        args.lanes = 1                     # the PCIe-inclusive and the uint8 / video variants are single-lane measurements
    n_lanes = max(1, args.lanes)
    lanes = None
    if n_lanes > 1:
        from implementation_phd_lab_vision_amd.backbone import BackboneLanes
        lanes = BackboneLanes(lanes=n_lanes, state_dict=sd, max_batch=args.batch, micro_batch=args.micro_batch, precision=args.precision).to(dev).eval()
        bb = lanes.lane0                   # lane 0 alone: the output check, the per-class event profile, the single-lane figure
    else:
        bb = ResNet50Backbone(state_dict=sd, max_batch=args.batch, micro_batch=args.micro_batch, precision=args.precision).to(dev).eval()
    _lane0 = bb

    class _AllLanes:                       # options go to every lane
        def set_option(self, k, v):
            (lanes or _lane0).set_option(k, v)
    bb_opts = _AllLanes()
    if args.streams:
        bb_opts.set_option("streams", args.streams)
    if args.no_fused_stem:
        bb_opts.set_option("fused_stem", 0)
    if args.no_stem_c1:
        bb_opts.set_option("fuse_stem_c1", 0)
    if args.no_ds_cat:
        bb_opts.set_option("fuse_ds_cat", 0)
    if args.no_fuse_tail:
        bb_opts.set_option("fuse_tail", 0)
    if args.no_overlap_ds:
        bb_opts.set_option("overlap_ds", 0)
    if args.inplace:
        bb_opts.set_option("inplace_out", 1)
    if args.no_block2:
        bb_opts.set_option("fuse_block2", 0)
    if args.no_cat_chain:
        bb_opts.set_option("fuse_cat_chain", 0)
    if args.block1 >= 0:
        bb_opts.set_option("fuse_block1", args.block1)
    for kv in args.opt:
        k_, v_ = kv.split("=", 1)
        bb_opts.set_option(k_, int(v_))
    x = synthetic_frames(args.batch, seed=1234 + rank).to(dev)          # random data, resident in HBM
    run = bb.features
    if args.input == "u8":
        g = torch.Generator().manual_seed(1234 + rank)
        x = torch.randint(0, 256, (args.batch, 3, 224, 224), generator=g, dtype=torch.uint8).to(dev)
        run = bb.features_u8
    if args.input == "video":          # what the video decoder hands over (H36M frames are 1002x1000): 3 MB per frame
        from implementation_phd_lab_vision_amd.frames import crop_and_resize_video_uint8
        g = torch.Generator().manual_seed(1234 + rank)
        x = torch.randint(0, 256, (args.batch, 1002, 1000, 3), generator=g, dtype=torch.uint8).to(dev)
        box = [251, 250, 500, 500]

        def run(frames, out=None):
            return bb.features_u8(crop_and_resize_video_uint8(frames, box), out=out)
    feats_l = [torch.empty((args.batch, 2048), dtype=torch.float32, device=dev) for _ in range(n_lanes)]      # one output buffer per lane
    feats = feats_l[0]
    gathered = [torch.empty_like(feats) for _ in range(world)] if (dist is not None and rank == 0) else None

    def gather_feats(f=None):
        f = feats if f is None else f
        if rehearsal:                    # gloo has no device-tensor gather: through host copies (control flow only)
            dist.gather(f.cpu(), [g.cpu() for g in gathered] if rank == 0 else None, dst=0)
        else:
            dist.gather(f, gathered, dst=0)

    if args.from_host:
        x_host = x.cpu().pin_memory()
        xbuf = [x, torch.empty_like(x)]
        copy_stream = torch.cuda.Stream(dev)
        copied = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        state = {"k": 0}
        copied[0].record(torch.cuda.current_stream(dev))
        consumed[1].record(torch.cuda.current_stream(dev))

        def step():
            k = state["k"]
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(copy_stream):                   # next step's frames: host -> the other device buffer
                copy_stream.wait_event(consumed[(k + 1) & 1])
                xbuf[(k + 1) & 1].copy_(x_host, non_blocking=True)
                copied[(k + 1) & 1].record(copy_stream)
            cur.wait_event(copied[k & 1])
            run(xbuf[k & 1], out=feats)
            consumed[k & 1].record(cur)
            state["k"] = k + 1
            if dist is not None:
                gather_feats()
    elif lanes is not None:
        lane_gain = lanes.tune(x)          # the lanes' streams really run side by side (changed if not; ONE lane if they cost time); untimed, results unaffected
        n_lanes = lanes.active_lanes
        lane_state = {"k": 0, "free": [None] * n_lanes}

        def step():
            # step k on lane k % L: its launches are queued behind the lane's previous step only; x is static and feats_l[lane] is free
            # once the gather that read it (multi-rank runs) has run -- `free[lane]`, recorded on the main stream behind that gather
            k = lane_state["k"]
            lane = k % n_lanes
            t = lanes.submit(x, out=feats_l[lane], after=lane_state["free"][lane])
            if dist is not None:
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(t.event)
                gather_feats(feats_l[lane])
                ev = torch.cuda.Event()
                ev.record(cur)
                lane_state["free"][lane] = ev
            lane_state["k"] = k + 1
    else:
        def step():
            run(x, out=feats)
            if dist is not None:
                gather_feats()

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if args.stream_frames:          # fixed total work: rank r owns the contiguous frame range [r*per, (r+1)*per) of the stream
        per_rank = (args.stream_frames + world - 1) // world
        args.steps = max(1, (per_rank + args.batch - 1) // args.batch)
    def sync_max(v: float) -> float:
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed, preheat_rate = timed_steps(step, fence, args.steps, args.warmup, args.preheat, world * args.batch, sync_max)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    single_lane = None
    if lanes is not None:
        torch.cuda.synchronize(dev)
        lanes_agree = all(bool(torch.equal(feats_l[0], f)) for f in feats_l[1:n_lanes])
        if dist is None:                   # the same K steps strictly one after the other on lane 0 (no second batch in flight)
            def step1():
                run(x, out=feats)
            el1, _ = timed_steps(step1, fence, args.steps, 1, min(args.preheat, 0.3), args.batch, None)
            single_lane = {"value": args.batch * args.steps / el1, "unit": "frames/s", "ms_per_step": 1e3 * el1 / args.steps,
                           "note": "lanes = 1: every step waits for the previous one's last launch (the measurement of rounds 1-2)"}
    else:
        lanes_agree = None

    # ---- the timed output is checked (outside the timed region): finite, and frames 0 / last equal to the same two frames run
    #      alone as a batch of 2 (the kernels' tile choice, persistent tile streams and ragged last tiles differ between the two runs;
    #      the arithmetic per frame must not)
    checked = None
    headline_feats = feats[:2].cpu() if (args.input == "f32" and not args.from_host and rank == 0) else None
    if args.input == "f32" and not args.from_host:
        failure = None
        finite = bool(torch.isfinite(feats).all())
        if not finite:
            failure = "non-finite features in the timed output"
        checked = {"finite": finite}
        if lanes_agree is not None:
            checked["lanes_equal"] = lanes_agree
            if not lanes_agree:
                failure = "the lanes' outputs for the same batch differ"
        if finite and not args.no_check:
            pick = [0, args.batch - 1] if args.batch > 1 else [0]
            small = bb.features(x[pick].contiguous())
            same = bool(torch.equal(small, feats[pick]))
            checked.update({"frames": pick, "equal_to_batch%d_run" % len(pick): same})
            if not same and args.precision != "fp32x":
                failure = (f"frames {pick} of the timed batch differ from the same frames run alone: "
                           f"max abs diff {float((small - feats[pick]).abs().max())}")
        # every rank leaves together: a rank that exited alone would leave the others hanging in the collectives below
        any_failed = sync_max(1.0 if failure else 0.0) > 0.0
        if any_failed:
            if dist is not None:
                dist.destroy_process_group()
            raise SystemExit(f"bench.py (rank {rank}): " + (failure or "another rank's timed output failed its check"))

    # ---- per-kernel-class HIP-event timing on the launch stream (rank 0, same steps again) ----
    roofline = None
    kernels = None
    if rank == 0:
        n_streams = bb.get_option("streams")
        bb.set_option("streams", 1)          # one stream: event brackets must not interleave with another stream's kernels
        bb.set_option("profile", 1)
        bb.profile_reset()
        for _ in range(args.steps):
            run(x, out=feats)
        torch.cuda.synchronize(dev)
        prof = bb.profile_collect()
        bb.set_option("profile", 0)
        bb.set_option("streams", n_streams)
        # HBM bytes per launch from rocprofv3 PMC passes (scripts/pmc_bench.sh: FETCH_SIZE x2 + WRITE_SIZE, full-batch launches only).
        # A PMC pass cannot run inside this process: the figures are STATIC ones from the committed profile of the same build on another
        # box of the pool, and the line says so (`traffic_source`); `algorithmic_bytes` beside them is computed live from this run.
        tpath = next((p for p in (ROOT / "profiles" / "r04_pmc_hbm_traffic.json", ROOT / "profiles" / "r03_pmc_hbm_traffic.json") if p.exists()), None)
        tdata = {}
        if tpath is not None:
            try:
                tdata = json.loads(tpath.read_text())
            except Exception:
                tdata = {}
        traffic_source = (f"profiles/{tpath.name} (rocprofv3 --pmc passes on ANOTHER box of the pool, static; not measured in this run)"
                          if tpath is not None else None)
        roofline = roofline_object(prof, args.steps, args.precision, tdata, traffic_source)
        if args.dump_layers:          # launches of one forward in launch order: stem, the bottleneck launches, avgpool
            order = ([k for k in ("stem_pack", "conv1", "maxpool") if k in prof and prof[k]["launches"]]
                     + [k for k in prof if k.startswith("layer") and prof[k]["launches"]] + ["avgpool"])
            Path(args.dump_layers).write_text(json.dumps([
                {"name": k, "launches_per_step": prof[k]["launches"] / args.steps, "bytes_per_launch": prof[k]["bytes"] / max(1, prof[k]["launches"]),
                 "flops_per_launch": prof[k]["flops"] / max(1, prof[k]["launches"]), "us_per_launch": 1e3 * prof[k]["ms"] / max(1, prof[k]["launches"])}
                for k in order], indent=1))
        classes = ("igemm", "bneck_tail", "bneck_tail3", "bneck_block2", "bneck_catchain", "conv1", "maxpool", "avgpool", "stem_pack")
        tot_ms = sum(prof[k]["ms"] for k in classes if k in prof)
        kernels = {k: {"launches_per_step": prof[k]["launches"] / args.steps, "ms_per_step": prof[k]["ms"] / args.steps,
                       "share": prof[k]["ms"] / tot_ms if tot_ms else 0.0} for k in classes if k in prof and prof[k]["launches"]}
        kernels["stages_ms_per_step"] = {f"layer{i}": sum(v["ms"] for n_, v in prof.items() if n_.startswith(f"layer{i}.")) / args.steps
                                         for i in (1, 2, 3, 4)}

    secondary = {}
    acc_in = {}
    if rank == 0 and world == 1 and not args.no_secondary and args.precision == "bf16" and args.input == "f32" and not args.from_host:
        (lanes or bb).close()         # frees the headline handles' workspaces before the secondary ones are created
        torch.cuda.empty_cache()
        for prec, b in (("fp16", 256), ("fp8", 512)):
            secondary[f"{prec}_b{b}"], acc_in[prec] = secondary_mode(prec, b, args.steps, args.warmup, min(args.preheat, 0.5), dev, sd, n_lanes)

    cpu = None
    accuracy = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if headline_feats is not None and args.precision in ("bf16", "fp16") and args.batch >= 2:
            acc_in["headline"] = {"feats": headline_feats, "precision": args.precision}
        cpu = cpu_baseline(accuracy_of=acc_in)
        for prec, a in (cpu.pop("accuracy", None) or {}).items():
            if prec == "headline":
                accuracy = a
                continue
            key = [k for k in secondary if k.startswith(prec)][0]
            secondary[key]["accuracy"] = a

    dist_used = dist is not None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        frames = world * args.batch * args.steps
        value = frames / elapsed
        out = {
            "metric": "H36M frames/sec ResNet-50 feature extraction",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if args.stream_frames else "weak",
            "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "fp16", "bf16w2": "bf16 (weights as bf16 head+tail pairs)",
                                            "fp32x": "bf16x3 (fp32-class)", "fp8": "fp8 e4m3 (layer2-4; stem + layer1 bf16)"}[args.precision],
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo; not a measurement)" if rehearsal else ""),
            "config": {"workload": f"ResNet-50[:-1] {args.precision} forward, batch {args.batch} x 224x224x3 fp32 NCHW frames per GPU "
                                   f"(BASELINE configs[{4 if args.precision == 'fp8' else 1}]), seeded synthetic weights, (N,2048) fp32 features"
                                   + (", RCCL gather to rank 0" if dist_used else ""),
                       "batch_per_gpu": args.batch, "micro_batch": args.micro_batch,
                       "lanes": n_lanes, "lanes_requested": max(1, args.lanes), "lanes_mode": (lanes.tune_mode if lanes is not None else "one lane"),
                       "lanes_tune": (getattr(lanes, "tune_log", None) if lanes is not None else None), "lanes_note": (f"{n_lanes} independent batches in flight, each on its own backbone copy + HIP stream (steps dealt round robin; "
                                                        "every step is a whole batch-%d forward; nothing is shared or skipped)" % args.batch) if n_lanes > 1 else "one batch at a time",
                       "input": args.input + (" from pinned host memory every step (PCIe-inclusive, H2D overlapped)" if args.from_host else ""),
                       "parallelism": f"frames sharded over {world} rank(s)"},
            # `value` is measured in this execution mode; `single_lane` (same run) is the strictly-one-batch-at-a-time figure, and `roofline` /
            # `kernels` are measured on lane 0 alone (with two batches in flight a launch's duration is not its own)
            "mode": (f"{n_lanes} batches in flight (BackboneLanes)" if n_lanes > 1 else "one batch at a time"),
            "tflops": value * GFLOP_PER_FRAME / 1e3,
            "frac_of_mfma_peak_whole_step": value * GFLOP_PER_FRAME / 1e3 / MFMA_BF16_PEAK_TFLOPS / world,
            "frac_of_mfma_peak_whole_step_single_lane": (single_lane["value"] * GFLOP_PER_FRAME / 1e3 / MFMA_BF16_PEAK_TFLOPS) if single_lane else None,
            "preheat": {"seconds": args.preheat, "frames_per_s": preheat_rate},
            "checked": checked, "single_lane": single_lane,
            # the tolerance statement of the HEADLINE mode: bf16 is the reference's own GPU dtype (torch.autocast(bfloat16),
            # src/preprocess_resnet_features.py:290-294) and lands 2.5e-3 from the fp32/fp64 view of the network (weight rounding);
            # north_star's 1e-3 is met by the fp16 mode (secondary `fp16_b256`, same kernels with IEEE half operands)
            "accuracy": accuracy,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        }
        if args.stream_frames:
            out["config"]["stream_frames"] = args.stream_frames
            out["config"]["frames_processed"] = frames
        out.update(secondary)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
