#!/usr/bin/env python3
"""Headline benchmark: refined masks / second on synthetic 640x480 RGB-D, N = 20 initial instances per frame,
batch 16 per GPU (BASELINE.json configs[1]); weak scaling over --gpus (one process per GPU, frames sharded,
RCCL broadcast of the weights at start-up and gather of the refined label maps per step).

A step = one pass of the hot path over one resident batch:
    encode initial masks -> network (fp32 MFMA) -> grouping / merge / scores -> per-instance masks.
Prints ONE JSON line on rank 0: value = whole-job masks/s; `roofline` for the dominant kernel (the implicit-GEMM
convolution, MFMA-bound) plus one `hbm_stages` entry per HBM-bound stage, all measured live with HIP events on the
launch stream (quber_profile_begin/end); `cpu_baseline` = the oracle (pure-torch CPU restatement of the reference
path) timed on this box's host cores over a bounded sample.

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the N ranks itself (child processes
started BEFORE anything touches the GPU); under the driver's torchrun launch it just joins the group.  A world
size that differs from --gpus is an error, never a silent 1-rank run.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA" (dense)
HBM_PEAK_GBPS = 8000.0          # same guide, "HBM3E peak BW" (spec; 6.29 TB/s measured float4 copy)

# stage tags (quber_profile_stage) of the convolution family and of the HBM-bound stages reported in `hbm_stages`
CONV_GEMM = ("conv_gemm", "conv_gemm_h8", "conv_gemm_x8", "wino_gemm", "wino_gemm_x8", "wino_fused")   # conv_gemm_h8: the fp16 path's 256 x 256 LDS-DMA kernel (csrc/conv_h8.hip); wino_fused: a Winograd layer as ONE kernel (transforms inside), priced on what it multiplies
CONV_GEMM_F32PIPE = ("conv_gemm_f32pipe",)   # launches of the bf16x3 mode that keep the exact fp32 MFMA kernel (short K, narrow tiles)
CONV_FAMILY = CONV_GEMM + CONV_GEMM_F32PIPE + ("splitk_reduce", "wino_input", "wino_output", "stem_fused")   # stem_fused: a3 + stem.conv1, vector FMAs (csrc/stem.hip)
HBM_STAGES = ("encode_reduce", "encode_paint", "errmaps_pack", "errmaps_erode", "errmaps_quadruple", "preprocess", "stem_fused", "wino_input",
              "wino_output", "splitk_reduce", "gn_stats", "gn_apply", "maxpool", "bilinear", "predictor", "upsample_logits",
              "post_nms", "post_select", "post_group", "post_paint_stats", "extract_masks")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--instances", type=int, default=20)
    ap.add_argument("--dtype", default="f32", choices=("f32", "f32-bf16x3", "f16", "bf16"),
                    help="arithmetic of the convolutions: f32 (headline; exact fp32 MFMA); f32-bf16x3 (fp32 operands split into "
                         "three bf16 terms, six partial products on the bf16 matrix pipe, fp32 accumulation: fp32-equivalent, "
                         "same 1e-4 bar); f16 / bf16 operands with fp32 accumulation (BASELINE.json configs[4] stand-in; "
                         "their own tolerance, see DESIGN.md)")
    ap.add_argument("--heads", default="loud", choices=("loud", "faithful"),
                    help="loud: O(1) predictors + calibrated centre bias so post-processing sees K ~ N instances per frame; "
                         "faithful: the reference's N(0, 0.001) predictor init (K = 0)")
    ap.add_argument("--cpu-frames", type=int, default=10, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-grad-frames", type=int, default=3, help="frames timed as the reference runs (autograd graph built)")
    ap.add_argument("--no-gather", action="store_true", help="skip the per-step RCCL gather of label maps")
    ap.add_argument("--graph", action="store_true", help="capture the step's ~330 launches in one hipGraph and replay it")
    ap.add_argument("--host-io", action="store_true",
                    help="host-inclusive mode: inputs start in pinned host memory and the refined masks end there every step "
                         "(double-buffered H2D / D2H on copy streams, as the reference's timed region includes them); "
                         "reported as host_io, never as `value`")
    ap.add_argument("--predict-calls", type=int, default=60,
                    help="calls of the drop-in MaskRefinerPredictor.predict() timed for `predict_api` (numpy in -> "
                         "pred_masks.numpy() out, one frame per call as the reference runs; 0 = skip)")
    ap.add_argument("--foreground-filter", action="store_true",
                    help="also run the reference adapter's LMFFNet foreground post-filter on the refined masks in every "
                         "step (eval/refiner_model.py:273-277; off for the headline metric, which is the refiner path)")
    ap.add_argument("--no-graph", action="store_true", help="N > 1 runs replay the step as one hipGraph by default (--graph); this keeps them eager")
    ap.add_argument("--tuning", default="", help="A/B options of the engine (quber_set_option), e.g. 5=0 (include/quber_hip.h)")
    ap.add_argument("--no-split-mode", action="store_true",
                    help="skip the extra timing of the fp32-equivalent bf16x3 mode that a default (f32, 1 GPU) run appends as "
                         "`fp32_equivalent_bf16x3` (never the headline `value`)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the sub-records a default (f32, 1 GPU, 640x480 batch 16) run appends under `configs`: BASELINE.json configs[2] "
                         "(1280x720, 30 instances, batch 1, hipGraph replay) and configs[4] (fp16 data path, 1024x1024, batch 8)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N = 1 through the N > 1 code path: a ONE-rank RCCL process group on this GPU (broadcast of the weights, hipGraph "
                         "capture beside the group's watchdog, the asynchronous label-map gather, MAX all-reduce, barrier) - the contact "
                         "with ProcessGroupNCCL / RCCL a one-GPU box allows; the auxiliary records are skipped")
    ap.add_argument("--dry", action="store_true",
                    help="no GPU work: exercise the launch / rendezvous / broadcast / gather path only (CPU tests, gloo)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(a):
    """--gpus N without a torchrun environment: start the N ranks as child processes and exit with their status.
    Nothing in this process has touched the GPU (no torch.cuda call, no HIP library loaded)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def source_digest():
    """sha256 over the kernel sources: measured-traffic files are only trusted for the code they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "quber_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                phys.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, (min(len(phys), usable) if phys else usable), usable


def cpu_baseline(sd, h, w, n, frames, grad_frames, eng=None, batch_eng=None, batch=None):
    """The oracle end to end (encode -> network -> grouping -> instances), batch 1 like the reference, executed the way the
    reference executes it (op by op, no folding, the head-fusion stack re-evaluated per key: model.py:760-762).
    Timed under torch.no_grad() (favours the baseline) and, on fewer frames, with the autograd graph the reference builds
    (predictor.py:358); `value` is the faster.  With `batch_eng`, frames of the BENCHMARKED batch are compared with the HIP
    results of the batch-sized engine (the metric's 'IoU delta vs ref')."""
    import torch
    from oracle import encode_np, postproc_ref
    from oracle.network_torch import ArchCfg, MaskRefinerNet
    from quber_amd import synth
    model, cores, usable = cpu_info()
    net = MaskRefinerNet(ArchCfg(repeat_fusion=True)).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)

    def one(sc, grad):
        t0 = time.perf_counter()
        offs = encode_np.encode_initial_masks(sc["masks"])
        image = torch.from_numpy(np.concatenate([sc["rgb"], sc["depth"]], -1)).permute(2, 0, 1)[None]
        with torch.enable_grad() if grad else torch.no_grad():
            out = net(image, torch.from_numpy(offs[None]))
            out = {k: v.detach() for k, v in out.items()}
        ref = postproc_ref.postprocess(out["foreground"][0], out["center"][0], out["offset"][0])
        return time.perf_counter() - t0, out, ref

    # the thread count that serves the baseline best (more is not faster: batch-1 convolutions of this size lose to the
    # fork / join and cache traffic of 128 threads): one frame per candidate after a warm-up frame, the fastest is used
    probe = {}
    sc0 = synth.make_scene(99, h, w, n)
    for t in sorted({c for c in (8, 16, 32, 64, cores) if c <= usable}):
        torch.set_num_threads(t)
        one(sc0, False)
        probe[t] = one(sc0, False)[0]
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    res = {}
    for name, grad, cnt in (("no_grad", False, frames), ("grad", True, grad_frames)):
        if cnt <= 0:
            continue
        times = [one(synth.make_scene(100 + i, h, w, n), grad)[0] for i in range(cnt + 1)][1:]   # first sample dropped (eval_utils.py:342)
        res[name] = {"frames": cnt, "median_ms_per_frame": float(np.median(times)) * 1e3,
                     "masks_per_s": n / float(np.median(times))}
    parity = None
    if batch_eng is not None:
        # frames 0 and B-1 of the benchmarked batch through the oracle (float32 as the reference computes, and float64 as
        # the anchor) vs the batch-B HIP results already on the device: the stated tolerance "within 1e-4 (float) /
        # bit-exact (label maps)" adjudicated as tests/test_gpu_loud_parity.py::test_benchmarked_plan_float64_anchor does
        from oracle import fp64_anchor as fa
        lg_all, pan_all = batch["logits"].cpu(), batch["panoptic"].cpu()
        chk_frames = sorted({0, lg_all.shape[0] - 1})
        anchor, flips, flip_error, iou, ks, dl = fa.AnchorErrors(), [], None, [], [], []
        image = torch.from_numpy(np.concatenate([batch["host"]["rgb"], batch["host"]["depth"]], -1)).permute(0, 3, 1, 2)
        offs = np.stack([encode_np.encode_initial_masks(batch["host"]["masks"][i]) for i in range(lg_all.shape[0])])
        o64 = {i: fa.cat_heads(o)[0] for i, o, _ in fa.oracle64(sd, image, offs, frames=chk_frames, want_taps=())}
        for i in chk_frames:
            sc = {k: batch["host"][k][i] for k in ("rgb", "depth", "masks")}
            _, out, ref = one(sc, False)
            exp = fa.cat_heads(out)
            anchor.add_heads(lg_all[i:i + 1], exp, o64[i][None])
            dl.append(float((lg_all[i] - exp[0]).abs().max()))
            try:
                flips.append(fa.explain_label_flips(lg_all[i], exp[0], o64[i], pan_hip=pan_all[i]))
            except AssertionError as e:
                flip_error = flip_error or f"frame {i}: {e}"
            a_, b_ = pan_all[i] >= 0, ref["panoptic"] >= 0
            union = int((a_ | b_).sum())
            iou.append(float((a_ & b_).sum()) / union if union else 1.0)
            ks.append(len(ref["labels"]))
        ok, bad = anchor.verdict()
        tot = fa.summarize(flips)
        parity = {"engine_batch": int(lg_all.shape[0]), "frames_checked": len(chk_frames),
                  "within_stated_tolerance": bool(ok and flip_error is None),
                  "stated_tolerance": "1e-4 on every head output in head units (what the predictors emit; the offset planes are "
                                      "multiplied by the common stride 4 afterwards, model.py:700) against the fp32 oracle; max |HIP - "
                                      "fp64| <= 1.5 x max |oracle_fp32 - fp64| per head; every label pixel that differs from the "
                                      "oracle's map is a float64 near-tie of the decision that produced it (margins = 1e-4 on "
                                      "logits, 1.13e-3 px on centre distances), otherwise bit-exact",
                  "failures": bad + ([flip_error] if flip_error else []),
                  "per_head": {k: {"max_abs_hip_vs_oracle_fp32": e["hip_vs_oracle32"], "max_abs_hip_vs_fp64": e["hip"],
                                   "max_abs_oracle_fp32_vs_fp64": e["oracle32"],
                                   "unit": "px (raw, after x4)" if k == "offset_px_raw" else "head units"}
                               for k, e in anchor.err.items()},
                  "max_abs_dlogit_raw": max(dl),
                  "label_map_equal_fraction": tot["label_map_equal_fraction"], "label_flips": tot,
                  "fg_iou": min(iou), "oracle_instances_per_frame": ks}
    best = max(res.values(), key=lambda r: r["masks_per_s"])
    return {"value": best["masks_per_s"], "unit": "refined masks/s", "cores": threads, "kind": "port",
            "cpu_model": model, "physical_cores": cores, "usable_cpus": usable, "torch": torch.__version__,
            "threads_probe_s_per_frame": {str(k): v for k, v in probe.items()}, "variants": res, "parity_vs_hip": parity,
            "sample": f"{frames} frames {w}x{h} N={n} batch 1 under no_grad + {grad_frames} with the autograd graph, median per "
                      f"frame, first sample dropped; reference-style execution (fusion stack per key); torch.set_num_threads = the "
                      f"fastest of {sorted(probe)} on one probe frame each (`cores` = that count)"}


def dry_main(a, world, rank):
    """CPU rehearsal of everything around the hot path: rendezvous, weight broadcast, frame sharding, label-map gather,
    max-over-ranks timing, the JSON line.  No GPU, no engine."""
    import torch
    import torch.distributed as dist
    from quber_amd import arch, dist as qdist
    if world > 1:
        torch.set_num_threads(1)
        dist.init_process_group(os.environ.get("QUBER_DIST_BACKEND", "gloo"))
    specs = arch.param_specs()
    wire = qdist.label_wire_dtype()
    sd = arch.init_state_dict(seed=0) if rank == 0 else None
    if world > 1:
        sd = qdist.broadcast_state_dict(sd, specs, src=0, device="cpu")
    digest = hashlib.sha256(np.concatenate([np.asarray(sd[k]).ravel()[:64] for k in specs]).tobytes()).hexdigest()[:12]
    B, H, W = a.batch, 32, 48
    local = torch.full((B, H, W), float(rank), dtype=torch.float32)
    got, pending, gather_s, step_ms = None, None, 0.0, []
    t0 = time.perf_counter()
    for _ in range(a.steps):                     # the step loop of the real run: asynchronous gather, waited one step later
        ts = time.perf_counter()
        if world > 1:
            tg = time.perf_counter()
            if pending is not None:
                got = pending.wait()
            pending = qdist.gather_label_maps(local, [B] * world, dst=0, async_op=True, wire_dtype=wire)
            gather_s += time.perf_counter() - tg
        else:
            got = local
        step_ms.append((time.perf_counter() - ts) * 1e3)
    if pending is not None:
        tg = time.perf_counter()
        got = pending.wait()
        gather_s += time.perf_counter() - tg
    elapsed = time.perf_counter() - t0
    alone = None
    if world > 1:
        tg = time.perf_counter()
        qdist.gather_label_maps(local, [B] * world, dst=0, wire_dtype=wire)
        alone = (time.perf_counter() - tg) * 1e3
    ranks = [{"rank": rank, "world_size": world, "device": "cpu", "device_name": "cpu (dry)", "backend": dist.get_backend() if world > 1 else None,
              "weights": digest, "step_ms_median": float(np.median(step_ms)), "step_ms_max": float(np.max(step_ms)),
              "elapsed_ms_per_step": elapsed / a.steps * 1e3, "gather_ms_per_step": gather_s / a.steps * 1e3,
              "gather_alone_ms": alone, "instances_out": 0, "frames": B}]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        allr = [None] * world
        dist.all_gather_object(allr, ranks[0])
        ranks = allr
    if rank == 0:
        assert got.shape[0] == world * B and all(float(got[r * B].mean()) == r for r in range(world))
        assert len({r["weights"] for r in ranks}) == 1, "weight broadcast diverged"
        print(json.dumps({"metric": "refined masks/sec on 640x480 RGB-D (N=20 inst)", "value": None, "unit": "refined masks/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                          "dry": True, "config": {"workload": "dry run: rendezvous + broadcast + gather only"},
                          "rccl_ranks": ranks,
                          "gather": None if ranks[0].get("backend") is None else {
                              "ms_per_step_max_over_ranks": max(r["gather_ms_per_step"] for r in ranks),
                              "alone_ms_max_over_ranks": max((r["gather_alone_ms"] or 0.0) for r in ranks),
                              "bytes_per_rank_per_step": B * H * W * 2, "wire_dtype": "int16"},
                          "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launch environment has WORLD_SIZE={world}; refusing to run "
                 f"(use `python bench.py --gpus {a.gpus}` alone, or torchrun with --nproc-per-node {a.gpus})")
    if a.dry:
        return dry_main(a, world, rank)
    multi = world > 1 or a.rccl_selftest       # the collective path (a one-rank group exercises the same calls)
    if a.rccl_selftest and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")

    import torch
    from quber_amd import arch, dist as qdist, engine, synth
    dist = None
    cpu_share = None
    if multi:
        # one rank per GPU on ONE node: give every rank its own slice of the host's CPUs and one OpenMP thread - eight ranks each
        # forking a 128-thread OpenMP team for a torch CPU op cost 90 ms per occurrence (profiles/r03q_predict_profile.txt)
        try:
            cpus = sorted(os.sched_getaffinity(0))
            share = len(cpus) // world
            if share >= 1:
                os.sched_setaffinity(0, cpus[local * share:(local + 1) * share])
                cpu_share = share
        except (AttributeError, OSError):
            pass
        torch.set_num_threads(1)
        import torch.distributed as dist
        backend = os.environ.get("QUBER_DIST_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 path on one GPU
        ndev = torch.cuda.device_count()                           # (counting devices does not initialise the GPU)
        if backend == "nccl" and ndev < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} over RCCL needs {a.gpus} visible GPUs, this node shows {ndev} "
                     f"(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); QUBER_DIST_BACKEND=gloo rehearses the N > 1 path on fewer")
        local = local % max(ndev, 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != a.gpus:
            sys.exit(f"bench.py: process group has {dist.get_world_size()} ranks, expected {a.gpus}")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    B, H, W, N = a.batch, a.height, a.width, a.instances
    loud = a.heads == "loud"

    def make_engine(sd, a=a):
        qc = engine.make_config(H, W, max_batch=B, max_instances=max(N, 1))
        qc.compute_dtype = {"f32": 0, "bf16": 1, "f16": 2, "f32-bf16x3": 3}[a.dtype]
        e = engine.Engine(qc, dev)
        for kv in filter(None, a.tuning.split(",")):      # before the plan is built: some options act at plan time
            k, v = kv.split("=")
            if int(k) in (2, 11, 12, 26):     # keys of the stand-alone test ops: process-wide, no engine reads them
                raise SystemExit(f"--tuning {k}={v}: key {k} belongs to the quber_op_* test harness (quber_set_tuning), not to an engine")
            e.set_option(int(k), int(v))
        e.load_state_dict(sd)
        return e

    # ---- synthetic inputs; each rank has its own frames ----
    host = synth.make_batch(7 + rank, B, H, W, N)
    masks = torch.from_numpy(host["masks"]).to(dev)
    gt_masks = torch.from_numpy(host["gt_masks"]).to(dev)
    bgr = torch.from_numpy(host["rgb"]).to(dev)
    depth = torch.from_numpy(host["depth"]).to(dev)
    offsets = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)

    # ---- weights: rank 0 owns the checkpoint, the others receive it in one RCCL broadcast ----
    specs = arch.param_specs()
    center_bias = 0.0
    if rank == 0:
        sd = arch.init_state_dict(seed=0, loud_heads=loud)
        if loud:
            # centre-head bias such that ~N local maxima per frame pass the 0.3 threshold (arch.calibrate_center_bias):
            # one forward of rank 0's frames with bias 0, then the weights are rebuilt with the bias
            eng0 = make_engine(sd)
            lg0 = eng0.forward(bgr, depth, eng0.encode(masks))
            center_bias = arch.calibrate_center_bias(lg0[:, 1:2].float().cpu(), N)
            eng0.close()
            del eng0, lg0
            sd = arch.init_state_dict(seed=0, loud_heads=True, center_bias=center_bias)
    else:
        sd = None
    if dist is not None:
        sd = qdist.broadcast_state_dict(sd, specs, src=0, device=dev)
    eng = make_engine(sd)

    logits = torch.empty((B, eng.planes, H, W), dtype=torch.float32, device=dev)
    post = eng.alloc_post(B)
    max_inst = min(eng.cap, max(N, 1) + 12)
    out_masks = torch.empty((B, max_inst, H, W), dtype=torch.uint8, device=dev)
    counts = [B] * world

    lmff = None
    if a.foreground_filter:
        from quber_amd import lmff_arch
        from quber_amd.foreground.predictor import LmffEngine
        lmff = LmffEngine(lmff_arch.init_state_dict(0), H, W, B, device=str(dev))

    def gpu_step(m=masks, b_=bgr, d_=depth, om=out_masks):
        eng.encode(m, offsets)
        eng.forward(b_, d_, offsets, logits)
        eng.postprocess(logits, post)
        eng.extract_masks(post, max_inst, om)
        if lmff is not None:
            lmff.foreground(b_, d_, om)

    graph = None
    use_graph = a.graph or (multi and not a.no_graph)      # N > 1: one launch per step keeps eight host processes out of each other's way
    if use_graph:
        # every launch of the step goes to torch's current stream and nothing allocates or synchronises,
        # so the whole step is capturable (include/quber_hip.h contract)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gpu_step()
        torch.cuda.current_stream().wait_stream(side)
        try:
            graph = torch.cuda.CUDAGraph()
            # with a process group alive its watchdog thread queries events while this thread captures: "thread_local" keeps those
            # calls from invalidating the capture (the default "global" mode treats them as errors)
            with torch.cuda.graph(graph, capture_error_mode="thread_local" if multi else "global"):
                gpu_step()
        except RuntimeError as e:                 # a node that cannot capture must still produce the line: eager steps
            if a.graph:
                raise
            print(f"[bench rank {rank}] hipGraph capture failed ({e}); running eager steps", file=sys.stderr, flush=True)
            graph = None
            torch.cuda.synchronize()

    wire = qdist.label_wire_dtype(top_k=eng.cap)     # int16 label maps on the wire (values -1, 1000 ... 1200): half the f32 bytes
    pending = [None]      # the label-map gather of the previous step: it travels over xGMI while this step computes
    gather_host_s = [0.0]  # host time inside the gather calls (issue + wait) - what the gather costs the step loop

    def step():
        if graph is not None:
            graph.replay()
        else:
            gpu_step()
        if dist is not None and not a.no_gather:
            t_g = time.perf_counter()
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = qdist.gather_label_maps(post["panoptic"], counts, dst=0, async_op=True, wire_dtype=wire)
            gather_host_s[0] += time.perf_counter() - t_g

    def drain():
        if pending[0] is not None:
            t_g = time.perf_counter()
            pending[0].wait()
            pending[0] = None
            gather_host_s[0] += time.perf_counter() - t_g

    for _ in range(a.warmup):
        step()
    drain()
    gather_host_s[0] = 0.0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # per-step device times (spread only)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        step()
        marks[i + 1].record()
    drain()                     # the last step's gather belongs to the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    # the gather alone, synchronous, nothing beside it: what a step would pay if it were NOT overlapped
    gather_alone_ms = None
    if dist is not None and not a.no_gather:
        ts = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t_g = time.perf_counter()
            qdist.gather_label_maps(post["panoptic"], counts, dst=0, wire_dtype=wire)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t_g) * 1e3)
        gather_alone_ms = float(np.median(ts[1:]))
    cnt_host = post["count"].cpu().numpy()
    rank_info = {"rank": rank, "world_size": dist.get_world_size() if dist is not None else 1, "device": str(dev),
                 "device_name": torch.cuda.get_device_name(dev), "backend": dist.get_backend() if dist is not None else None,
                 "step_ms_median": float(np.median(step_ms)), "step_ms_max": float(np.max(step_ms)),
                 "elapsed_ms_per_step": elapsed / a.steps * 1e3,
                 "gather_ms_per_step": gather_host_s[0] / a.steps * 1e3 if dist is not None else 0.0,
                 "gather_alone_ms": gather_alone_ms,
                 "instances_out": int(cnt_host.sum()), "frames": int(B), "cpus": cpu_share, "hipgraph": graph is not None,
                 # what a first contact with an 8-GPU node needs to explain itself: the collective library, the visible devices
                 "rccl_version": _rccl_version(torch), "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES"),
                 "rocr_visible_devices": os.environ.get("ROCR_VISIBLE_DEVICES"), "local_rank": os.environ.get("LOCAL_RANK"),
                 "hip_runtime": getattr(torch.version, "hip", None)}
    ranks = [rank_info]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_info)

    host_io = None
    if a.host_io and rank == 0:
        host_io = host_io_run(a, eng, host, offsets, logits, post, max_inst, dev)

    if rank == 0:
        line = report(a, eng, world, elapsed, ranks, center_bias, sd, host, step_ms,
                      dict(masks=masks, gt_masks=gt_masks, bgr=bgr, depth=depth, offsets=offsets, logits=logits, post=post,
                           out_masks=out_masks, max_inst=max_inst), gpu_step)
        if host_io is not None:
            line["host_io"] = host_io
        if not multi and a.predict_calls > 0 and a.dtype == "f32":
            line["predict_api"] = predict_api_run(a, sd, host, dev)
            if H == 480 and W == 640:        # the adapter resizes every frame to 640x480 (eval/refiner_model.py:246)
                try:
                    line["predict_api"]["streamed"] = predict_stream_run(a, sd, host, dev)
                except Exception as e:           # an auxiliary figure must never cost the line its headline
                    line["predict_api"]["streamed"] = {"error": repr(e)}
        if a.dtype == "f32" and not multi and not a.no_split_mode:
            line["fp32_equivalent_bf16x3"] = split_mode_run(a, make_engine, sd, gpu_step_args=(masks, bgr, depth, offsets, max_inst),
                                                            exact_logits=logits, exact_pan=post["panoptic"])
        if (a.dtype == "f32" and not multi and not a.no_configs and not a.tuning and (B, H, W, N) == (16, 480, 640, 20)
                and a.heads == "loud" and not a.graph and not a.foreground_filter):
            eng.close()              # the headline engine's buffers are not needed any more
            line["configs"] = {}
            for name, kw in (("stream_1280x720_b1_graph", dict(H=720, W=1280, B=1, N=30, dtype="f32", graph=True, steps=max(2 * a.steps, 40), warmup=max(a.warmup, 10))),
                             ("f16_1024x1024_b8", dict(H=1024, W=1024, B=8, N=20, dtype="f16", graph=False, steps=a.steps, warmup=a.warmup))):
                try:
                    line["configs"][name] = config_run(dev, **kw)
                except Exception as e:           # an auxiliary record must never cost the line its headline
                    line["configs"][name] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def conv_roofline(stages, dtype, algorithmic):
    """`roofline` of one profiled step (stage table of quber_profile_end): the convolution family priced as report() prices it."""
    fam_ms = sum(stages[k]["ms"] for k in CONV_FAMILY if k in stages)
    f32p_ms = sum(stages[k]["ms"] for k in CONV_GEMM_F32PIPE if k in stages)
    executed = sum(stages[k]["flops"] for k in CONV_GEMM if k in stages)
    if dtype == "f32-bf16x3":
        executed *= 6.0
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    t = fam_ms - f32p_ms
    ach = executed / (t * 1e-3) / 1e12 if t else 0.0
    return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "frac_algorithmic": algorithmic / (fam_ms * 1e-3) / 1e12 / peak if fam_ms else None,
            "conv_family_ms": {k: stages[k]["ms"] for k in CONV_FAMILY if k in stages}}


def config_run(dev, H, W, B, N, dtype, graph, steps, warmup):
    """One more BASELINE.json configuration, timed like the headline (resident inputs, encode -> network -> grouping -> mask
    extraction per step, `warmup` untimed steps, `steps` timed ones between two synchronisations) and reported beside it - never as
    `value`.  graph = the step captured once in a hipGraph and replayed (configs[2]: 'hipGraph-captured steady state')."""
    import torch
    from quber_amd import arch, engine, synth

    def make(sd):
        qc = engine.make_config(H, W, max_batch=B, max_instances=max(N, 1))
        qc.compute_dtype = {"f32": 0, "bf16": 1, "f16": 2, "f32-bf16x3": 3}[dtype]
        e = engine.Engine(qc, dev)
        e.load_state_dict(sd)
        return e

    host = synth.make_batch(11, B, H, W, N)
    masks, bgr, depth = (torch.from_numpy(host[k]).to(dev) for k in ("masks", "rgb", "depth"))
    offsets = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    eng0 = make(arch.init_state_dict(seed=0, loud_heads=True))
    lg0 = eng0.forward(bgr, depth, eng0.encode(masks))
    bias = arch.calibrate_center_bias(lg0[:, 1:2].float().cpu(), N)
    eng0.close()
    del eng0, lg0
    eng = make(arch.init_state_dict(seed=0, loud_heads=True, center_bias=bias))
    logits = torch.empty((B, eng.planes, H, W), dtype=torch.float32, device=dev)
    post = eng.alloc_post(B)
    max_inst = min(eng.cap, max(N, 1) + 12)
    om = torch.empty((B, max_inst, H, W), dtype=torch.uint8, device=dev)

    def gpu_step():
        eng.encode(masks, offsets)
        eng.forward(bgr, depth, offsets, logits)
        eng.postprocess(logits, post)
        eng.extract_masks(post, max_inst, om)

    g = None
    if graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gpu_step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            gpu_step()
    step = g.replay if g is not None else gpu_step
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    eager_ms = None
    if g is not None:                 # the same step launched eagerly, beside the replayed graph (at batch <= 2 the side lanes do better outside a graph)
        for _ in range(warmup):
            gpu_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            gpu_step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / steps * 1e3
    runs = []
    for _ in range(3):
        eng.profile_begin()
        gpu_step()
        runs.append(eng.profile_end())
    stages = {k: dict(runs[0][k], ms=float(np.median([r[k]["ms"] for r in runs]))) for k in runs[0]}
    count = post["count"].cpu().numpy()
    out = {"value": B * N * steps / el, "unit": "refined masks/s", "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warmup,
           "step_ms_min_median_max": [float(np.min(dev_ms)), float(np.median(dev_ms)), float(np.max(dev_ms))],
           "frames_per_s": B * steps / el, "dtype": dtype, "hipgraph": g is not None, "eager_ms_per_step": eager_ms,
           "workload": f"batch={B} {W}x{H} RGB-D, {N} initial instances/frame, ResNet-50 RGB-D refiner, "
                       f"encode+network+grouping+mask extraction, inputs resident",
           "instances_out_per_frame_mean": float(count.mean()),
           "roofline": conv_roofline(stages, dtype, eng.forward_flops() * B)}
    eng.close()
    return out


def split_mode_run(a, make_engine, sd, gpu_step_args, exact_logits, exact_pan):
    """The same step with quber_config.compute_dtype = 3: every fp32 operand of the convolutions split into three bf16 terms,
    six exact partial products per multiply on the bf16 matrix pipe, fp32 accumulation (dropped terms < 2^-26 of a product) -
    fp32-equivalent arithmetic held to the same 1e-4 parity bars (tests/test_gpu_loud_parity.py), reported BESIDE the exact
    fp32 MFMA headline, never as it."""
    import torch
    masks, bgr, depth, offsets, max_inst = gpu_step_args
    B, H, W, N = a.batch, a.height, a.width, a.instances
    a2 = argparse.Namespace(**vars(a))
    a2.dtype = "f32-bf16x3"
    eng = make_engine(sd, a2)
    logits = torch.empty_like(exact_logits)
    post = eng.alloc_post(B)
    om = torch.empty((B, max_inst, H, W), dtype=torch.uint8, device=exact_logits.device)

    def step():
        eng.encode(masks, offsets)
        eng.forward(bgr, depth, offsets, logits)
        eng.postprocess(logits, post)
        eng.extract_masks(post, max_inst, om)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    d = (logits - exact_logits).abs()
    out = {"value": B * N * a.steps / el, "unit": "refined masks/s", "ms_per_step": el / a.steps * 1e3,
           "vs_exact_fp32_mfma": {"max_abs_dlogit_head_units": float(torch.maximum(d[:, [0, 1] + list(range(4, d.shape[1]))].max(),
                                                                                  d[:, 2:4].max() / 4)),
                                  "label_map_equal_fraction": float((post["panoptic"] == exact_pan).float().mean())},
           "arithmetic": "fp32 operands split into 3 bf16 terms (round-to-nearest), 6 partial products per multiply on "
                         "v_mfma_f32_32x32x16_bf16, fp32 accumulation; same plan (Winograd F(4x4) included) and same parity bars as "
                         "the exact fp32 MFMA mode"}
    eng.close()
    return out


def predict_api_run(a, sd, host, dev):
    """The call eval/run_eval.py actually makes, timed as the reference times it (eval/refiner_model.py:265-271):
        start; output = predictor.predict(rgb, depth, masks)[0]; masks = output['instances'].to('cpu').pred_masks.numpy(); stop
    numpy arrays in, numpy masks out, one frame per call (predictor.py:358), cycling through the frames of the step.
    Beside it, the engine's own batch-1 step on resident inputs (encode + network + grouping + extraction of the same
    number of masks) - what the API call would cost with no host side at all - and the general batched path fed one frame."""
    import torch
    from quber_amd import engine
    from quber_amd.maskrefiner.predictor import MaskRefinerPredictor
    B, H, W, N = a.batch, a.height, a.width, a.instances
    pred = MaskRefinerPredictor(None, device=str(dev), state_dict=sd)

    def call(i):
        t0 = time.perf_counter()
        out = pred.predict(host["rgb"][i % B], host["depth"][i % B], host["masks"][i % B])[0]
        m = out["instances"].to("cpu").pred_masks.numpy() if "instances" in out else []
        return time.perf_counter() - t0, len(m)

    def series(n):
        for i in range(5):
            call(i)
        r = [call(i) for i in range(n)]
        return np.array([t for t, _ in r]) * 1e3, [k for _, k in r]

    fast, ks = series(max(a.predict_calls, 10))
    pred.fast_path = False
    general, _ = series(max(a.predict_calls // 3, 10))
    # engine-only batch-1 step on resident inputs
    eng = pred.model.engine_for(H, W, 1, N)
    masks = torch.from_numpy(host["masks"][:1]).to(dev)
    bgr, depth = torch.from_numpy(host["rgb"][:1]).to(dev), torch.from_numpy(host["depth"][:1]).to(dev)
    offsets = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
    logits = torch.empty((1, eng.planes, H, W), dtype=torch.float32, device=dev)
    post = eng.alloc_post(1)
    om = torch.empty((1, max(ks[0], 1), H, W), dtype=torch.uint8, device=dev)

    def step():
        eng.encode(masks, offsets)
        eng.forward(bgr, depth, offsets, logits)
        eng.postprocess(logits, post)
        eng.extract_masks(post, om.shape[1], om)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(max(a.predict_calls, 10)):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    eng_ms = float(np.median(ts))
    med = float(np.median(fast))
    return {"median_ms_per_call": med, "p10_p90_ms": [float(np.percentile(fast, 10)), float(np.percentile(fast, 90))],
            "calls": len(fast), "value": N / (med * 1e-3), "unit": "refined masks/s",
            "engine_batch1_step_ms": eng_ms, "over_engine_step": med / eng_ms,
            "general_batched_path_ms_per_call": float(np.median(general)),
            "instances_out_per_call_mean": float(np.mean(ks)),
            "note": "MaskRefinerPredictor.predict(rgb, depth, masks) + output['instances'].to('cpu').pred_masks.numpy(), numpy in / "
                    "numpy out, one frame per call (the reference's timed region, eval/refiner_model.py:265-271); "
                    "engine_batch1_step = the same frame on resident device buffers, no host side"}


def predict_stream_run(a, sd, host, dev, frames=128):
    """The reference's evaluation loop (eval/eval_utils.py:235-286: MaskRefiner.predict(rgb_path, depth_path, masks) per frame) through
    this repo's drop-in adapter in its streamed, batched form: files on disk in, numpy masks out.  Host work (PNG decoding, resize,
    depth normalisation, TELEA in-painting of the depth holes, upload) on worker threads, `batch` frames per engine call, one batch in
    flight while the previous one is copied out (quber_amd/eval/refiner_model.py:predict_stream)."""
    import tempfile
    from PIL import Image
    from quber_amd.eval.refiner_model import MaskRefiner
    B, N = a.batch, a.instances
    workers = max(2, min(16, (os.cpu_count() or 4) - 2))
    with tempfile.TemporaryDirectory() as d:
        items = []
        rng = np.random.default_rng(0)
        for i in range(min(B, 8)):
            Image.fromarray(host["rgb"][i][:, :, ::-1].copy()).save(os.path.join(d, f"rgb{i}.png"))
            mm = host["depth"][i][:, :, 0].astype(np.uint16) * 5 + 300
            for _ in range(12):                                            # ~8 000 zero-depth pixels in a dozen holes: work for the in-painting
                y, x = int(rng.integers(0, mm.shape[0] - 40)), int(rng.integers(0, mm.shape[1] - 40))
                mm[y:y + 22, x:x + 30] = 0
            Image.fromarray(mm).save(os.path.join(d, f"depth{i}.png"))
            items.append((os.path.join(d, f"rgb{i}.png"), os.path.join(d, f"depth{i}.png"), host["masks"][i] != 0, None))
        ref = MaskRefiner(None, None, dataset="OSD", device=str(dev))
        ref.refiner_predictor.model.state_dict = sd
        ref.refiner_predictor.model._engines.clear()
        work = [items[i % len(items)] for i in range(frames)]
        out = {}
        for name, kw in (("batch1_2workers", dict(workers=2, batch=1)), (f"batch{B}_{workers}workers", dict(workers=workers, batch=B))):
            list(ref.predict_stream(work[:max(2 * B, 8)], **kw))           # warm-up: engine for this batch size, worker streams
            t0 = time.perf_counter()
            res = list(ref.predict_stream(work, **kw))
            dt = time.perf_counter() - t0
            out[name] = {"frames_per_s": frames / dt, "masks_per_s": N * frames / dt, "ms_per_frame": dt / frames * 1e3,
                         "instances_out_mean": float(np.mean([len(r[0]) for r in res]))}
        out["note"] = ("files -> refined numpy masks through MaskRefiner.predict_stream (PNG decode, resize, normalize_depth, TELEA in-painting, "
                       "upload on worker threads; eval/eval_utils.py:235-286 drives the reference's adapter frame after frame)")
        out["frames"] = frames
        return out


def host_io_run(a, eng, host, offsets, logits, post, max_inst, dev):
    """The reference's timed region (eval/refiner_model.py:265-271) starts with numpy inputs and ends with numpy masks.
    Pinned staging buffers, two device buffer sets, one H2D and one D2H copy stream: the upload of step i+1 and the
    download of step i-1 overlap the compute of step i."""
    import torch
    B, H, W, N = a.batch, a.height, a.width, a.instances
    pin = {k: torch.from_numpy(host[k]).pin_memory() for k in ("masks", "rgb", "depth")}
    dev_in = [{k: torch.empty_like(v, device=dev) for k, v in pin.items()} for _ in range(2)]
    dev_out = [torch.empty((B, max_inst, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
    host_out = [torch.empty((B, max_inst, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
    host_cnt = [torch.empty((B,), dtype=torch.int32).pin_memory() for _ in range(2)]
    dev_cnt = [torch.empty((B,), dtype=torch.int32, device=dev) for _ in range(2)]   # per-slot snapshot: post["count"] is rewritten by step i+1
    up, down, main = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()
    ev_up = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_free_in = [torch.cuda.Event() for _ in range(2)]
    ev_down = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        s = i & 1
        with torch.cuda.stream(up):
            up.wait_event(ev_free_in[s])            # the compute that last read this input set has finished
            for k in pin:
                dev_in[s][k].copy_(pin[k], non_blocking=True)
            ev_up[s].record(up)

    def run(steps):
        for s in range(2):
            ev_free_in[s].record(main)
            ev_down[s].record(down)
        upload(0)
        for i in range(steps):
            s = i & 1
            if i + 1 < steps:
                upload(i + 1)
            main.wait_event(ev_up[s])
            main.wait_event(ev_down[s])             # the previous download of this output set has finished
            eng.encode(dev_in[s]["masks"], offsets)
            eng.forward(dev_in[s]["rgb"], dev_in[s]["depth"], offsets, logits)
            eng.postprocess(logits, post)
            eng.extract_masks(post, max_inst, dev_out[s])
            dev_cnt[s].copy_(post["count"])         # on the main stream, before ev_done: the download reads the snapshot
            ev_free_in[s].record(main)
            ev_done[s].record(main)
            with torch.cuda.stream(down):
                down.wait_event(ev_done[s])
                host_out[s].copy_(dev_out[s], non_blocking=True)
                host_cnt[s].copy_(dev_cnt[s], non_blocking=True)
                ev_down[s].record(down)
        torch.cuda.synchronize()

    run(max(a.warmup, 2))
    t0 = time.perf_counter()
    run(a.steps)
    el = time.perf_counter() - t0
    per_step_mb = (sum(v.numel() for v in pin.values()) + host_out[0].numel()) / 1e6
    return {"value": B * N * a.steps / el, "unit": "refined masks/s", "ms_per_step": el / a.steps * 1e3,
            "pcie_mb_per_step": per_step_mb,
            "note": "inputs (bgr, depth, initial masks) start in pinned host memory, refined masks u8 [B,max_inst,H,W] end there; "
                    "double-buffered copy streams; single GPU"}


def _rccl_version(torch):
    """RCCL's version as torch reports it (backend "nccl" IS RCCL on ROCm), or the reason it cannot."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:      # noqa: BLE001 - a diagnostic field must never fail the benchmark
        return f"unavailable ({type(e).__name__})"


def report(a, eng, world, elapsed, ranks, center_bias, sd, host, step_ms, t, gpu_step):
    import torch
    B, H, W, N = a.batch, a.height, a.width, a.instances
    # ---- stage profile: HIP events around every kernel of one step (+ the explicit error maps, a2), median of 3 ----
    runs = []
    for _ in range(3):
        eng.profile_begin()
        gpu_step()
        eng.error_maps(t["masks"], t["gt_masks"])
        runs.append(eng.profile_end())
    stages = {}
    for k in runs[0]:
        stages[k] = dict(runs[0][k])
        stages[k]["ms"] = float(np.median([r[k]["ms"] for r in runs]))
    fam_ms = sum(stages[k]["ms"] for k in CONV_FAMILY if k in stages)
    gemm_ms = sum(stages[k]["ms"] for k in CONV_GEMM if k in stages)
    gemm_n = sum(stages[k]["launches"] for k in CONV_GEMM if k in stages)
    executed = sum(stages[k]["flops"] for k in CONV_GEMM if k in stages)       # what the matrix pipe multiplies (2*M*K*N per launch)
    # bf16x3 mode: the launches that stayed on the exact fp32 MFMA kernel are priced on THEIR pipe, apart
    f32p_ms = sum(stages[k]["ms"] for k in CONV_GEMM_F32PIPE if k in stages)
    f32p_n = sum(stages[k]["launches"] for k in CONV_GEMM_F32PIPE if k in stages)
    f32p_flops = sum(stages[k]["flops"] for k in CONV_GEMM_F32PIPE if k in stages)
    algorithmic = eng.forward_flops() * B                                        # 2 x MAC of the direct convolutions
    peak = FP32_MFMA_PEAK_TFLOPS if a.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    nominal = executed + f32p_flops   # fp32 multiply-adds the GEMM launches evaluate
    if a.dtype == "f32-bf16x3":
        executed *= 6.0          # six bf16 partial products per fp32 multiply: what the bf16 matrix pipe executes
    fam_ms_pipe = fam_ms - f32p_ms                                               # the dominant pipe's share of the family time
    hbm = {}
    for k in HBM_STAGES:
        if k in stages and stages[k]["ms"] > 0:
            s = stages[k]
            gbps = s["bytes"] / (s["ms"] * 1e-3) / 1e9
            hbm[k] = {"bytes": s["bytes"], "ms": s["ms"], "launches": s["launches"], "GBps": gbps, "frac": gbps / HBM_PEAK_GBPS}
    traffic, traffic_note = None, "no measurement on file"
    tpath = os.path.join(ROOT, "profiles", "conv_hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("source_digest") == source_digest() and tj.get("dtype", "f32") == a.dtype:
                traffic = tj.get("bytes_per_launch")
                traffic_note = f"rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE per convolution-family launch, measured on kernel sources {tj['source_digest']}"
            else:
                traffic_note = "profiles/conv_hbm_traffic.json was measured on other kernel sources (or another dtype): not reported"
        except Exception:
            traffic_note = "profiles/conv_hbm_traffic.json unreadable"
    count = t["post"]["count"].cpu().numpy()
    ms_per_step = elapsed / a.steps * 1e3
    ex_tf = executed / (fam_ms_pipe * 1e-3) / 1e12 if fam_ms_pipe else 0.0
    line = {
        "metric": f"refined masks/sec on {W}x{H} RGB-D (N={N} inst)",
        "value": world * B * N * a.steps / elapsed,
        "unit": "refined masks/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "step_ms_min_median_max_rank0": [float(np.min(step_ms)), float(np.median(step_ms)), float(np.max(step_ms))],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"batch={B} {W}x{H} RGB-D, {N} initial instances/frame, ResNet-50 RGB-D refiner "
                               f"(boundary-error -> fg/centre/offset), encode+network+grouping+mask extraction",
                   "frames_per_step_per_gpu": B, "parallelism": f"dp{world}", "hipgraph": bool(ranks[0].get("hipgraph", a.graph)),
                   "foreground_filter": bool(a.foreground_filter),
                   "weights": ("seeded synthetic, loud predictors N(0, %.2f), centre bias %.4f calibrated for ~N peaks per frame"
                               % (arch_sigma(), center_bias)) if a.heads == "loud"
                              else "seeded synthetic, reference init (predictors N(0, 0.001): K = 0)",
                   "instances_out_per_frame_mean": float(count.mean()),
                   "instances_out_per_frame_min_max": [int(count.min()), int(count.max())]},
        "rccl_ranks": ranks,
        "gather": None if ranks[0].get("backend") is None else {
            "ms_per_step_max_over_ranks": max(r["gather_ms_per_step"] for r in ranks),
            "alone_ms_max_over_ranks": max((r["gather_alone_ms"] or 0.0) for r in ranks),
            "bytes_per_rank_per_step": B * H * W * 2, "wire_dtype": "int16 (labels -1, 1000 ... 1200; f32 again on rank 0)",
            "note": "ms_per_step = host time inside the asynchronous gather's issue + wait calls per step (what the overlapped "
                    "gather costs the step loop); alone = one synchronous gather with nothing beside it (what it would cost "
                    "un-overlapped); per-rank values and step medians in rccl_ranks"},
        "roofline": {"bound": "mfma", "achieved": ex_tf, "peak": peak, "unit": "TFLOP/s", "frac": ex_tf / peak,
                     # SURVEY 8(d)'s definition beside the conservative one above: the ALGORITHMIC FLOPs of the direct convolutions
                     # (2 x MAC) over the same time and peak - above `frac`, and possibly above 1, exactly by what Winograd does not multiply
                     "frac_algorithmic": (algorithmic / (fam_ms * 1e-3) / 1e12 / peak) if fam_ms else None,
                     "traffic": traffic, "traffic_note": traffic_note,
                     "kernel": "conv_igemm (all instantiations).  achieved = FLOPs the matrix pipe EXECUTES per step (2*M*K*N of "
                               "every GEMM launch; the Winograd layers' transformed GEMMs count what they multiply, not the direct "
                               "convolution's 2*MAC) / device time of the whole convolution family (GEMM launches + split-K reduce + "
                               "Winograd input / output transforms), HIP events on the launch stream",
                     "gemm_kernel_tflops": executed / (gemm_ms * 1e-3) / 1e12 if gemm_ms else 0.0,
                     "gemm_kernel_frac": executed / (gemm_ms * 1e-3) / 1e12 / peak if gemm_ms else 0.0,
                     "algorithmic_tflops": algorithmic / (fam_ms * 1e-3) / 1e12 if fam_ms else 0.0,
                     "algorithmic_note": "2 x MAC of the direct convolutions (SURVEY 8d: 375.6 GFLOP per 640x480 frame) over the "
                                         "same time; exceeds `achieved` because Winograd F(m x m,3x3) executes (m+2)^2 / 9m^2 of them",
                     "executed_over_algorithmic": nominal / algorithmic if algorithmic else None,
                     # `achieved` counts what the matrix pipe multiplies, the padding of ragged Winograd tiles included (a 30x40 map under
                     # 4x4 tiles carries 6.7 % of it): the share of it, and `frac` with it taken out
                     "tile_padding_share_of_executed": (eng.forward_flops_padding() * B / nominal) if nominal else None,
                     "frac_without_tile_padding": (ex_tf / peak) * (1.0 - eng.forward_flops_padding() * B / nominal) if nominal else None,
                     "fp32_equivalent_tflops": nominal / (fam_ms * 1e-3) / 1e12 if fam_ms else 0.0,
                     "fp32_pipe_launches": None if not f32p_n else {
                         "launches": f32p_n, "ms": f32p_ms, "tflops": f32p_flops / (f32p_ms * 1e-3) / 1e12,
                         "frac_of_fp32_mfma_peak": f32p_flops / (f32p_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                         "note": "launches of this mode that keep an exact fp32 MFMA kernel (HBM-bound short-K / narrow-tile "
                                 "layers, the single-kernel Winograd layers); excluded from `achieved`, `frac` and gemm_kernel_* above, which price the bf16 pipe only"},
                     "conv_stages": {k: {"ms": stages[k]["ms"], "launches": stages[k]["launches"],
                                         "executed_tflops": stages[k]["flops"] / (stages[k]["ms"] * 1e-3) / 1e12 if stages[k]["ms"] else 0.0}
                                     for k in CONV_FAMILY if k in stages},
                     "gemm_launches_per_step": gemm_n, "avg_launch_ms": gemm_ms / max(gemm_n, 1),
                     "flops_per_launch": executed / max(gemm_n, 1),
                     "conv_family_ms": {k: stages[k]["ms"] for k in CONV_FAMILY if k in stages},
                     "hbm_stages": hbm,
                     "hbm_note": "bytes = algorithmic (every operand of the stage read / written once), ms = HIP-event device time "
                                 "per step, frac = GB/s / 8000 (HBM3E spec; 6.3 TB/s is the measured copy ceiling)"},
        "stage_ms_other": {k: stages[k]["ms"] for k in stages if k not in CONV_FAMILY and k not in hbm},
    }
    if world == 1 and a.cpu_frames > 0 and not a.rccl_selftest:
        bt = {"logits": t["logits"], "panoptic": t["post"]["panoptic"], "host": host}
        gpu_step()
        torch.cuda.synchronize()
        line["cpu_baseline"] = cpu_baseline(sd, H, W, N, a.cpu_frames, a.cpu_grad_frames, eng, eng, bt)
    else:
        line["cpu_baseline"] = None
    return line


def arch_sigma():
    from quber_amd import arch
    return arch.LOUD_PRED_SIGMA


if __name__ == "__main__":
    main()
