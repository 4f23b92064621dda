#!/usr/bin/env python
"""bench.py — image-pairs/s of the POPE hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic 640x480 pairs resident in HBM:
DINOv2-S/14 patch descriptors for both images of every pair (476x630 centre crop, 1531 tokens) +
dense dual-softmax / mutual-NN matching of the 1530x1530 descriptor pairs (BASELINE config 3:
"Extract + dense cosine-sim matcher + mutual-NN on 640x480 pairs, batch=128").

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W      # no launcher: starts that same command itself (self_launch)

Pairs are independent: every rank works on its own batch (weak scaling, no data-path collective);
the only exchange is the RCCL all_gather of per-pair match counts at the end of each step.
Rank 0 prints ONE JSON line.  After the timed region the step's own outputs are spot-checked (pairs of the batch
re-run one at a time must reproduce them bit for bit): `"verified": true`, non-zero exit otherwise.  At N = 1 the
default run also times a short strict-fp32 leg (`strict_f32`), BASELINE config 5's two encoders (`config5`) and the CPU
oracle (`cpu_baseline`).

`--workload linemod` (BASELINE config 4): one step = one pass over the 5 796 pair ids of the LINEMOD evaluation list
(tests/golden/linemod_pairs.json; pixels synthetic, keyed by pair id), sharded contiguously over the ranks, walked
in 128-pair batches with ragged tails, counts gathered once at the end (strong scaling).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H_IMG, W_IMG, PATCH = 476, 630, 14
NTOK = 1 + (H_IMG // PATCH) * (W_IMG // PATCH)  # 1531
DIM, HEADS, HIDDEN, DEPTH = 384, 6, 1536, 12
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16), same guide
PEAK_HBM_GBS = 8000.0
# arithmetic modes of the contractions (pope_hip.h POPE_PREC_*): MFMA dtype, dense peak, MFMA FLOPs
# executed per algorithmic FLOP (f16x3 = three partial products per product block)
PRECISION_INFO = {"f16x3": ("f16x3 (fp32 operands split hi+lo into f16, 3 MFMAs per product, fp32 accumulate)",
                            PEAK_F16_MFMA_TFLOPS, 3),
                  "f32": ("f32 (v_mfma_f32_32x32x2_f32)", PEAK_F32_MFMA_TFLOPS, 1),
                  # opt-in reduced precision (never the default line: the reference path is fp32): the ViT blocks in plain f16,
                  # one MFMA per product; patch embed and the matcher contraction stay f16x3
                  "f16": ("f16 (ViT blocks: plain f16 operands, 1 MFMA per product, fp32 accumulate; patch embed + matcher f16x3)",
                          PEAK_F16_MFMA_TFLOPS, 1)}
PAIR_LIST = os.path.join(ROOT, "tests", "golden", "linemod_pairs.json")


def flops_per_image(np_=NTOK - 1):
    n = np_ + 1
    return 451584 * np_ + 12 * (3538944 * n + 1536 * n * n)


def flops_per_pair():
    L = NTOK - 1
    return 2 * flops_per_image() + 2 * L * L * DIM


def gpu_pairs(n_pairs, device, seed):
    """Seeded synthetic pairs generated on the device: uniform[0,1) 640x480 frame,
    centre crop 476x630, ImageNet normalisation; img1 = roll(img0,(14,28)) + N(0,0.1^2)."""
    from pope_amd.synth import IMAGENET_MEAN, IMAGENET_STD
    g = torch.Generator(device=device).manual_seed(seed)
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(1, 3, 1, 1)
    img = torch.rand(n_pairs, 3, 480, 640, generator=g, device=device)
    img0 = ((img[:, :, 2:2 + H_IMG, 5:5 + W_IMG] - mean) / std).contiguous()
    img1 = torch.roll(img0, shifts=(14, 28), dims=(2, 3))
    img1 = (img1 + 0.1 * torch.randn(img1.shape, generator=g, device=device)).contiguous()
    return img0, img1


def kernel_flops(kind, chunk):
    """Algorithmic FLOPs (2*MAC) of ONE launch of a ViT kernel at the bench shape (chunk images)."""
    rows = chunk * NTOK
    return {"attention": 4 * chunk * NTOK * NTOK * DIM, "gemm_qkv": 2 * rows * DIM * 3 * DIM,
            "gemm_proj": 2 * rows * DIM * DIM, "gemm_fc1_gelu": 2 * rows * DIM * HIDDEN,
            "gemm_fc2": 2 * rows * HIDDEN * DIM,
            "patch_embed_gemm": 2 * chunk * (NTOK - 1) * 3 * PATCH * PATCH * DIM}.get(kind, 0)


def kernel_bytes(kind, chunk):
    """Algorithmic HBM bytes of one launch (activations in + out; weights are L2/MALL resident).  The LN-fused residual
    GEMMs (proj, FC2) read the operand planes and the residual and write x AND the next LayerNorm's output planes."""
    rows = chunk * NTOK
    return 4 * {"attention": rows * 4 * DIM, "gemm_qkv": rows * 4 * DIM, "gemm_proj": rows * 4 * DIM,
                "gemm_fc1_gelu": rows * (DIM + HIDDEN), "gemm_fc2": rows * (HIDDEN + 3 * DIM),
                "layernorm": rows * 2 * DIM,
                "patch_embed_gemm": chunk * 3 * H_IMG * W_IMG + rows * DIM}.get(kind, 0)


def cpu_baseline(n_pairs):
    """The CPU oracle (torch fp32 restatement validated against the reference) on the host cores."""
    from oracle import coarse_match_ref, dinov2_ref
    from pope_amd import synth
    # host-core share of this process (the GPU box gives a 1-GPU job 16 of its cores)
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    sd = synth.synthetic_state_dict(seed=0)
    i0, i1 = synth.synthetic_pairs(n_pairs, H_IMG, W_IMG, seed=0)
    hw_c = (H_IMG // PATCH, W_IMG // PATCH)

    def run():
        f0 = dinov2_ref.forward_features(sd, i0)["x_norm_patchtokens"]
        f1 = dinov2_ref.forward_features(sd, i1)["x_norm_patchtokens"]
        return coarse_match_ref.dense_match(f0, f1, hw_c, hw_c, (H_IMG, W_IMG))

    run()  # warm-up
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = run()
        times.append(time.perf_counter() - t0)
    rates = sorted(n_pairs / t for t in times)
    return {"value": round(rates[-1], 4), "unit": "image-pairs/s", "cores": torch.get_num_threads(),
            "kind": "port", "runs_pairs_per_s": [round(r, 4) for r in rates],
            "sample": f"{n_pairs} pairs 476x630 (batch {n_pairs} per forward), 1 warm-up + best of 3 "
                      f"(all three listed: run-to-run spread on shared host cores), {int(len(out['i_ids']))} matches"}


def verify_step(model, img0, img1, out, picks):
    """The step's own outputs against the unbatched drop-in path: for each picked pair the two images are run
    through the model alone and matched alone; descriptors, match indices and confidences must be bit-identical to
    the pair's slice of the batched step (images and pairs are independent in the reference).  Returns a list of
    failure strings (empty = verified)."""
    from pope_amd.matcher import dense_match
    hw_c = (H_IMG // PATCH, W_IMG // PATCH)
    bad = []
    counts = out["counts"]
    if int(counts.sum()) != len(out["b_ids"]):
        bad.append("counts do not add up to the match list")
    for k in picks:
        f0 = model(img0[k:k + 1], is_training=True)["x_norm_patchtokens"]
        f1 = model(img1[k:k + 1], is_training=True)["x_norm_patchtokens"]
        if not (torch.equal(f0[0], out["feat0"][k]) and torch.equal(f1[0], out["feat1"][k])):
            bad.append(f"pair {k}: descriptors differ from the batch-1 run")
        one = dense_match(f0, f1, hw_c, hw_c, (H_IMG, W_IMG), precision=getattr(model, "_bench_match_precision", None))
        sel = out["b_ids"] == k
        for key in ("i_ids", "j_ids", "mconf"):
            if not torch.equal(out[key][sel], one[key]):
                bad.append(f"pair {k}: {key} differs from the batch-1 run")
        if int(counts[k]) != len(one["i_ids"]) or len(one["i_ids"]) < 500:
            bad.append(f"pair {k}: {int(counts[k])} matches (batch-1 run: {len(one['i_ids'])}; planted: ~1131)")
    if not bool(torch.isfinite(out["feat0"]).all()):
        bad.append("non-finite descriptors")
    return bad


def self_launch(n):
    """Run `python bench.py --gpus N ...` as N ranks of ONE node: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` in a child process.  Rank 0's
    JSON line goes to stdout as it arrives, everything else to stderr; returns the launcher's exit code (non-zero when any
    rank failed).  Nothing here imports a GPU library or calls HIP."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    for line in proc.stdout:
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["batch128", "linemod"], default="batch128",
                    help="batch128 = BASELINE config 3 (the headline metric); linemod = config 4 (whole pair list per step)")
    ap.add_argument("--pairs", type=int, default=128, help="pairs per GPU per step (BASELINE config 3) / per batch (linemod)")
    ap.add_argument("--chunk", type=int, default=64, help="images per ViT launch sequence (BASELINE config 2)")
    ap.add_argument("--cpu-pairs", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--no-strict-f32", action="store_true", help="skip the short strict-fp32 leg of the N = 1 run")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the short BASELINE config 5 leg (DINOv2 ViT-L/14 and the SAM ViT-H image encoder) of the N = 1 run")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the LoFTR Matcher / driver-step / pose legs of the N = 1 run (bench_legs.py)")
    ap.add_argument("--only", choices=["loftr_matcher", "driver_step", "pose"], default=None,
                    help="profiling only: run just this leg (no headline line; prints the leg's JSON)")
    ap.add_argument("--no-verify", action="store_true",
                    help="profiling only: skip the self-check (its batch-1 launches would enter rocprof's per-kernel averages); "
                         "the line then carries \"verified\": null")
    ap.add_argument("--from-host", action="store_true",
                    help="add a leg that starts from uint8 640x480 frames in pinned HOST memory (PCIe upload on a copy stream, "
                         "crop + normalise on the GPU); reported as `from_host`, never as `value`")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the ViT chunks of a step are spread over")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="dev: run the N-rank code path with every rank on cuda:0 and the gloo backend (collectives on "
                         "CPU copies) — checks the multi-process flow on a 1-GPU box; the numbers mean nothing")
    ap.add_argument("--precision", choices=sorted(PRECISION_INFO), default="f16x3",
                    help="arithmetic of the Linear layers, attention and the matcher contraction (both are held to the "
                         "same parity tests)")
    ap.add_argument("--launch-selftest", choices=["ok", "fail"], default=None,
                    help="dev: the ranks only rendezvous over gloo on the CPU and rank 0 prints a stub line (\"fail\": rank 1 "
                         "exits non-zero) — tests the launch / relay / exit-code path of `bench.py --gpus N` without a GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: start the N ranks ourselves.  This parent must never touch the
        # GPU (a process that has initialised HIP cannot hand the card to children cleanly): it only spawns
        # torch.distributed.run, relays the ranks' output and passes the exit code on.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must equal --gpus")
    import torch.distributed as dist
    if args.launch_selftest:
        dist.init_process_group(backend="gloo")
        got = [None] * world
        dist.all_gather_object(got, rank)
        if args.launch_selftest == "fail" and rank == world - 1:
            raise SystemExit(3)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "ranks_seen": got}), flush=True)
        dist.destroy_process_group()
        return
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = torch.device("cpu") if args.rehearse_on_one_gpu else device   # where collective operands live
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)  # nccl == RCCL on ROCm

    from pope_amd import synth
    from pope_amd.dinov2_utils import load_dinov2_model
    from pope_amd.pipeline import PairPipeline, gather_counts, load_pair_list, shard_range, walk_pair_list

    if args.only in ("loftr_matcher", "driver_step"):   # profiling passes of the secondary legs (scripts/profile_round.sh)
        import bench_legs
        vit, matcher = bench_legs.build_models(device)
        if args.only == "loftr_matcher":
            leg = {f"pairs_{n}": bench_legs.loftr_matcher_leg(matcher, device, n, cpu_baseline=False) for n in (3, 24)}
        else:
            leg = bench_legs.driver_step_leg(vit, matcher, device, cpu_baseline=False)
        print(json.dumps({args.only: leg}))
        return

    model = load_dinov2_model(state_dict=synth.synthetic_state_dict(seed=0)).to(device)
    match_precision = "f16x3" if args.precision == "f16" else args.precision
    model.precision = args.precision
    model._bench_match_precision = match_precision
    dtype_name, peak_tflops, mfma_factor = PRECISION_INFO[args.precision]
    pipe = PairPipeline(model, chunk=args.chunk, streams=args.streams, match_precision=match_precision)
    from pope_amd.profiling import KernelProfiler

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    linemod = args.workload == "linemod"
    if linemod:
        # BASELINE config 4: the list is sharded contiguously; every rank pre-generates its shard's pixels in HBM
        # (5 796 pairs x 2 x 3.6 MB = 41.7 GB at N = 1), keyed by pair id, before anything is timed
        n_list = len(load_pair_list(PAIR_LIST))
        lo, hi = shard_range(n_list, rank, world)
        shard = {}
        for s in range(lo, hi, args.pairs):
            e = min(s + args.pairs, hi)
            shard[s] = synth.pairs_by_id(torch.arange(s, e), H_IMG, W_IMG, device=device)
        last = {}

        def process_batch(s, e):
            last["lo"], last["img"] = s, shard[s]
            last["out"] = pipe(*shard[s])
            return last["out"]["counts"]

        def step():
            counts, nb = walk_pair_list(n_list, process_batch, batch=args.pairs, rank=rank, world=world, device=coll_device)
            return last["out"], counts, nb
        pairs_per_step_total = n_list
        n_chunks = sum(2 * -(-(min(s + args.pairs, hi) - s) // args.chunk) for s in range(lo, hi, args.pairs))
    else:
        img0, img1 = gpu_pairs(args.pairs, device, seed=rank)

        def step():
            out = pipe(img0, img1)
            counts = gather_counts(out["counts"].to(coll_device))  # the only exchange: per-pair match counts
            return out, counts, 1
        pairs_per_step_total = args.pairs * world
        n_chunks = -(-args.pairs // args.chunk) * 2  # ViT launch sequences per step

    # The LAST warm-up step runs with every launch bracketed by HIP events (the per-kernel table and the choice
    # of the dominant kernel); the timed steps bracket only that dominant kernel, so every other launch stays
    # back to back exactly as in the untimed product path (bracketing all ~86 launches per forward costs ~2 %).
    survey = None if (args.no_kernel_table or args.warmup < 1) else KernelProfiler(DEPTH, n_chunks)
    prof = None
    dominant = "attention"
    for i in range(args.warmup):
        if survey is not None and i == args.warmup - 1:
            torch.cuda.synchronize()
            model.profiler = survey
        out, counts, _ = step()
    model.profiler = None
    tab = []
    if survey is not None:
        torch.cuda.synchronize()
        for kind, v in survey.summary().items():
            fl, by = kernel_flops(kind, args.chunk), kernel_bytes(kind, args.chunk)
            tab.append({"kernel": kind, "launches": v["launches"], "avg_ms": round(v["avg_ms"], 4),
                        "ms_per_step": round(v["total_ms"], 3),
                        "tflops": round(fl / v["avg_ms"] / 1e9, 2) if (fl and not linemod) else None,
                        "gbs": round(by / v["avg_ms"] / 1e6, 1) if (by and not linemod) else None})
        dominant = max((t for t in tab if kernel_flops(t["kernel"], args.chunk)), key=lambda t: t["ms_per_step"])["kernel"]
        survey.close()
    if not args.no_kernel_table and not linemod:
        prof = KernelProfiler(DEPTH, n_chunks * args.steps, kinds=(dominant,))

    fence()
    model.profiler = prof  # HIP events around every launch of the dominant kernel in the TIMED region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, counts, n_batches = step()
    fence()
    elapsed = time.perf_counter() - t0
    model.profiler = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # ---- self-check of the timed step's own outputs (every rank; all must pass) --------------------------------
    if linemod:
        v_img0, v_img1 = last["img"]
        picks = sorted({0, len(out["counts"]) // 2, len(out["counts"]) - 1})
    else:
        v_img0, v_img1 = img0, img1
        picks = sorted({0, args.pairs // 2, args.pairs - 1})
    failures = [] if args.no_verify else verify_step(model, v_img0, v_img1, out, picks)
    if counts.numel() != pairs_per_step_total:
        failures.append(f"gathered {counts.numel()} counts for {pairs_per_step_total} pairs")
    if model.overflow_events:
        failures.append(f"{model.overflow_events} f16x3 range-guard events on synthetic data")
    ok = torch.tensor([0 if failures else 1], dtype=torch.int32, device=coll_device)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    verified = None if args.no_verify else bool(int(ok))

    total_pairs = pairs_per_step_total * args.steps
    value = total_pairs / elapsed
    workload = ("extract(2 x DINOv2-S/14 @476x630 centre crop of 640x480, 1531 tokens) + dense "
                "dual-softmax mutual-NN match (1530x1530x384) per pair")
    result = {
        "metric": "image-pairs/s (DINOv2-S/14 extract+match, 640x480)",
        "value": round(value, 2), "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "strong" if linemod else "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": workload + (f"; one step = the {pairs_per_step_total} pairs of the LINEMOD evaluation list "
                                           f"(BASELINE config 4), {args.pairs}-pair batches with ragged tails"
                                           if linemod else " (BASELINE config 3)"),
                   "pairs_per_gpu_per_step": (hi - lo) if linemod else args.pairs, "vit_chunk_images": args.chunk,
                   "weights": "seeded synthetic, reference state-dict layout",
                   "want_conf": pipe.want_conf,   # False: match lists only — conf_matrix [n, L, S] (1.2 GB per step) is not published
                   "parallelism": f"pairs sharded over {world} GPU(s); RCCL all_gather of match counts"},
        "gflop_per_pair": round(flops_per_pair() / 1e9, 3),
        "achieved_tflops_whole_step": round(value * flops_per_pair() / 1e12 / world, 2),
        "matches_per_pair_mean": round(float(counts.float().mean()), 1),
        "verified": verified,
        "verified_note": f"pairs {picks} of the last timed step re-run one at a time (batch-1 ViT + batch-1 matcher): "
                         "descriptors, match indices and confidences bit-identical; counts consistent; no range-guard event",
    }
    if linemod:
        result["batches_per_rank_per_step"] = n_batches
    if failures:
        result["verify_failures"] = failures[:8]
    if prof is not None and rank == 0:
        v = prof.summary()[dominant]
        fl = kernel_flops(dominant, args.chunk)
        dom = {"kernel": dominant, "launches": v["launches"], "avg_ms": round(v["avg_ms"], 4),
               "tflops": round(fl / v["avg_ms"] / 1e9, 2)}
        result["roofline"] = roofline_entry(dominant, dom, peak_tflops, mfma_factor, args.chunk, pmc=args.precision != "f16")
        if tab:
            result["kernels"] = tab
            result["kernels_note"] = "every launch of the last warm-up step bracketed by HIP events"
            result["vit_kernel_ms_per_step"] = round(sum(t["ms_per_step"] for t in tab), 3)
    elif tab and rank == 0:
        result["kernels"] = tab

    # ---- N = 1 extras: strict-fp32 leg (driver-observed figure for the fp32-MFMA mode) and the CPU oracle -------
    if rank == 0 and world == 1 and not linemod and args.precision != "f32" and not args.no_strict_f32:
        result["strict_f32"] = strict_f32_leg(model, pipe, img0, img1, args)
    if rank == 0 and world == 1 and not linemod and args.from_host:
        result["from_host"] = from_host_leg(pipe, args, device)
    if rank == 0 and world == 1 and not linemod and not args.no_config5:
        torch.cuda.empty_cache()
        result["config5"] = config5_leg(device)
    if rank == 0 and world == 1 and not linemod and (not args.no_legs or args.only == "pose"):
        import bench_legs
        cpu = not args.no_cpu_baseline
        torch.cuda.empty_cache()
        try:
            result["pose"] = bench_legs.pose_leg(pipe, img0, img1, device, cpu_baseline=cpu)
        except Exception as e:  # noqa: BLE001 — reported in place, the headline line stands
            result["pose"] = {"error": f"{type(e).__name__}: {e}"}
        if args.only is None:
            try:   # BASELINE config 2: extraction only, one 64-image batch
                result["extract_only"] = bench_legs.extract_only_leg(model, img0, device, chunk=args.chunk)
            except Exception as e:  # noqa: BLE001
                result["extract_only"] = {"error": f"{type(e).__name__}: {e}"}
            try:   # the same step publishing conf_matrix [n, L, S] like the reference's CoarseMatching always does (coarse_matching.py:145)
                pipe.want_conf = True
                o2 = pipe(img0, img1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    o2 = pipe(img0, img1)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 2
                k = args.pairs // 2
                from pope_amd.matcher import dense_match
                one = dense_match(o2["feat0"][k:k + 1], o2["feat1"][k:k + 1], (H_IMG // PATCH, W_IMG // PATCH), (H_IMG // PATCH, W_IMG // PATCH),
                                  (H_IMG, W_IMG), precision=match_precision)
                result["with_conf_matrix"] = {"value": round(args.pairs / dt, 2), "unit": "image-pairs/s", "ms_per_step": round(dt * 1e3, 3),
                                              "conf_matrix_bytes_per_step": int(o2["conf_matrix"].numel()) * 4,
                                              "verified": bool(torch.equal(o2["conf_matrix"][k], one["conf_matrix"][0])
                                                               and torch.equal(o2["i_ids"][o2["b_ids"] == k], one["i_ids"])),
                                              "note": "want_conf=True: the matcher also publishes conf_matrix; pair %d's matrix bit-equal to its batch-1 call" % k}
                del o2, one
            except Exception as e:  # noqa: BLE001
                result["with_conf_matrix"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                pipe.want_conf = False
                torch.cuda.empty_cache()
            if args.streams == 1:
                try:   # the same step with the ViT chunks of a step spread over two HIP streams (PairPipeline(streams=2)): the chunks' tile-quantisation
                    # tails and HBM-bound kernels overlap the other chunk's MFMA phases.  Not the headline: overlapping launches stretch every
                    # kernel's own duration, and the headline's roofline entry is per-launch time of the dominant kernel on its stream.
                    pipe2 = PairPipeline(model, chunk=args.chunk, streams=2, match_precision=match_precision)
                    ref_out = pipe(img0, img1)
                    o2 = pipe2(img0, img1)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        o2 = pipe2(img0, img1)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t1) / 3
                    same = all(torch.equal(o2[k], ref_out[k]) for k in ("feat0", "feat1", "counts", "i_ids", "j_ids", "mconf"))
                    result["two_streams"] = {"value": round(args.pairs / dt, 2), "unit": "image-pairs/s", "ms_per_step": round(dt * 1e3, 3), "streams": 2,
                                             "verified": bool(same), "note": "every output bit-equal to the one-stream step's"}
                    del pipe2, o2, ref_out
                except Exception as e:  # noqa: BLE001
                    result["two_streams"] = {"error": f"{type(e).__name__}: {e}"}
                finally:
                    torch.cuda.empty_cache()
            try:
                ramp = bench_legs.attention_ramp_leg(device, chunk=args.chunk)
                result["attention_ramp"] = ramp
                # the headline if the model's attention scores behaved like the steepest ramp instead of the synthetic weights'
                # diffuse rows: every attention launch of the step charged the leg's measured difference to its ramp-0 time
                # (same run, same box).  Since round 4 the kernel advances its softmax reference BEFORE a tile's
                # exponentials (nothing is recomputed) and the difference is a few per cent either way.
                r = ramp["ramps_log2_units_per_tile"]
                worst = max(r, key=lambda k: r[k]["ms"])
                extra_ms = (r[worst]["ms"] - r["0.0"]["ms"]) * n_chunks * DEPTH
                result["value_at_ramp"] = {"ramp": float(worst), "value": round(pairs_per_step_total / ((1e3 * elapsed / args.steps + max(extra_ms, 0.0)) * 1e-3), 2),
                                           "unit": "image-pairs/s", "attention_ms": {k: v["ms"] for k, v in r.items()},
                                           "how": "derived: ms_per_step + (attention launches per step) x (ms at the slowest ramp - ms at ramp 0) of "
                                                  "the attention_ramp leg of this run; advance_rate at that ramp: "
                                                  f"{r[worst]['advance_rate']}"}
            except Exception as e:  # noqa: BLE001
                result["attention_ramp"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
            try:
                vit, matcher = bench_legs.build_models(device)
                result["loftr_matcher"] = {"pairs_3": bench_legs.loftr_matcher_leg(matcher, device, 3, cpu_baseline=cpu),
                                           "pairs_24": bench_legs.loftr_matcher_leg(matcher, device, 24)}
                result["driver_step"] = bench_legs.driver_step_leg(vit, matcher, device, cpu_baseline=cpu)
                del vit, matcher
            except Exception as e:  # noqa: BLE001
                result["loftr_matcher"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # reported at N = 1 only (the other ranks would idle)
        result["cpu_baseline"] = cpu_baseline(args.cpu_pairs)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
    if verified is False:
        raise SystemExit(3)


def from_host_leg(pipe, args, device, steps=4):
    """The same step when the inputs are uint8 frames in pinned host memory (what a capture or decode stage hands over):
    per step 2 x pairs frames of 640x480x3 bytes cross PCIe on a copy stream into one of two device staging buffers while
    the previous step computes; centre crop 476x630 + ToTensor + Normalize is one kernel (pope_crop_normalize_u8_f32).
    DESIGN.md §5: this PCIe-inclusive rate is reported next to `value`, never as `value`."""
    from pope_amd.preprocess import crop_normalize
    n = args.pairs
    g = torch.Generator().manual_seed(5)
    host = []
    for _ in range(2):   # second frame of a pair = the first one shifted by (14, 28) pixels: the matcher has work to do
        a = torch.randint(0, 256, (n, 480, 640, 3), dtype=torch.uint8, generator=g)
        host.append(torch.cat([a, torch.roll(a, shifts=(14, 28), dims=(1, 2))], 0).pin_memory())
    stage = [torch.empty(2 * n, 480, 640, 3, dtype=torch.uint8, device=device) for _ in range(2)]
    x = torch.empty(2 * n, 3, H_IMG, W_IMG, dtype=torch.float32, device=device)
    copy_stream = torch.cuda.Stream(device)
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]

    def upload(k):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(freed[k & 1])            # the compute stream is done with this staging buffer
            stage[k & 1].copy_(host[k & 1], non_blocking=True)
            ready[k & 1].record(copy_stream)

    def compute(k):
        cur = torch.cuda.current_stream(device)
        cur.wait_event(ready[k & 1])
        crop_normalize(stage[k & 1], (H_IMG, W_IMG), out=x)
        freed[k & 1].record(cur)
        return pipe(x[:n], x[n:])

    for e in freed:
        e.record(torch.cuda.current_stream(device))
    upload(0)
    upload(1)
    compute(0)                                               # warm-up
    torch.cuda.synchronize(device)
    upload(2)
    t0 = time.perf_counter()
    for k in range(1, 1 + steps):
        out = compute(k)
        upload(k + 2)                                        # two uploads ahead of the step that consumes them
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(n / dt, 2), "unit": "image-pairs/s", "steps": steps, "ms_per_step": round(dt * 1e3, 3),
            "host_bytes_per_step": 2 * n * 480 * 640 * 3, "matches_per_pair_mean": round(float(out["counts"].float().mean()), 1),
            "note": "uint8 640x480 frames in pinned host memory -> async H2D on a copy stream (double-buffered) -> GPU crop + "
                    "normalise -> the same extract + match step; PCIe-inclusive, not the headline"}


def roofline_entry(dominant, dom, peak_tflops, mfma_factor, chunk, pmc=True):
    traffic = None  # PMC-derived bytes per launch, collected in separate rocprofv3 --pmc passes (profiles/)
    try:            # (measured on the default f16x3 kernels: pmc=False for a mode whose kernels were not profiled that way)
        traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(dominant) if pmc else None
    except (OSError, ValueError):
        pass
    return {
        "kernel": dominant, "bound": "mfma", "achieved": dom["tflops"], "peak": peak_tflops,
        "unit": "TFLOP/s", "frac": round(dom["tflops"] / peak_tflops, 4), "traffic": traffic,
        "mfma_flops_per_algorithmic_flop": mfma_factor,
        "frac_of_executed_mfma_flops": round(mfma_factor * dom["tflops"] / peak_tflops, 4),
        "algorithmic_bytes_per_launch": kernel_bytes(dominant, chunk),
        "flops_per_launch": kernel_flops(dominant, chunk), "avg_ms_per_launch": dom["avg_ms"],
        "launches_timed": dom["launches"],
        "note": "algorithmic FLOPs / HIP-event duration of every launch of this kernel inside the timed "
                "region (events on the launch stream); peak = dense MFMA peak of the MFMA dtype "
                "(MI355X_MICROARCH.md); f16x3 executes 3 MFMA FLOPs per algorithmic FLOP; `traffic` = HBM bytes per "
                "launch from the committed rocprofv3 --pmc pass (profiles/pmc_traffic.json <- profiles/r04/pmc_summary.txt), "
                "not measured in this run",
    }


def config5_leg(device, iters=3):
    """BASELINE config 5 as two timed forward passes on synthetic weights and inputs (never part of `value`): the
    DINOv2 backbone swapped to ViT-L/14 on 640 x 480 crops (476 x 630, 21 images per launch sequence) and the SAM
    ViT-H image encoder on 1024 x 1024 (12 images per launch sequence).  Each figure carries a self-check: finite outputs and image 0 of the
    batch bit-equal to its own single-image run.  A failure is reported in place, the headline line stands."""
    from functools import partial
    from pope_amd import dinov2, synth
    from pope_amd.sam_encoder import ImageEncoderViT

    def timed(fn, x):
        fn(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            y = fn(x)
        e1.record()
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(y).all()) and torch.equal(fn(x[:1])[0], y[0])
        return e0.elapsed_time(e1) / iters, ok

    out = {"note": "synthetic weights and inputs; images resident in HBM; f16x3 arithmetic unless an entry says dtype f16", "iters": iters}
    try:
        m = dinov2.vit_large(patch_size=14, img_size=518, init_values=1e-5, ffn_layer="mlp", block_chunks=0)
        m.load_state_dict(synth.synthetic_state_dict(seed=0, dim=1024, depth=24), strict=True)
        m = m.eval().to(device)
        LB = 21   # images per launch sequence: 32 151 token rows = 126 row tiles of 256 -> 1.97 / 5.9 / 7.9 rounds of the 256 x 256 GEMM tiles
        x = synth.synthetic_images(LB, H_IMG, W_IMG, seed=3, device=device)
        npt = NTOK - 1
        fl = 2.0 * (1024 * 3 * 196 * npt + 24 * (NTOK * 12 * 1024 * 1024 + 2 * NTOK * NTOK * 1024))
        ref_out = None
        for prec, key in (("f16x3", "dinov2_vit_l14"), ("f16", "dinov2_vit_l14_f16")):
            m.precision = prec
            ms, ok = timed(lambda t: m(t, is_training=True)["x_norm_patchtokens"], x)
            out[key] = {"value": round(LB * 1e3 / ms, 1), "unit": "images/s", "batch": LB, "image": [H_IMG, W_IMG], "dtype": prec,
                        "ms_per_image": round(ms / LB, 3), "tflops_algorithmic": round(fl * LB / ms / 1e9, 1),
                        "frac_of_f16_mfma_peak_executed": round((3 if prec == "f16x3" else 1) * fl * LB / ms / 1e9 / PEAK_F16_MFMA_TFLOPS, 4),
                        "verified": ok and m.overflow_events == 0}
            y = m(x[:1], is_training=True)["x_norm_patchtokens"]
            if ref_out is None:
                ref_out = y
            else:
                out[key]["max_abs_diff_vs_f16x3"] = round(float((y - ref_out).abs().max()), 5)
        del m, x
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 — reported, not hidden
        out["dinov2_vit_l14"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        gidx = (7, 15, 23, 31)   # build_sam.py:13-21
        enc = ImageEncoderViT(depth=32, embed_dim=1280, img_size=1024, mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                              num_heads=16, patch_size=16, qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(gidx),
                              window_size=14, out_chans=256)
        enc.load_state_dict(synth.synthetic_sam_encoder_state_dict(seed=0, global_idx=gidx), strict=True)
        enc = enc.eval().to(device)
        SB = 16   # images per launch sequence: 256 row tiles of 256 -> 5 / 15 / 20 whole rounds of the 256 x 256 GEMM tiles (12 images: 3.75 / 11.25 / 15; 8: 2.5 / 7.5 / 10)
        enc.max_batch = SB
        x = synth.synthetic_images(SB, 1024, 1024, seed=3, device=device)
        n, dim, hd, heads = 4096, 1280, 80, 16
        lin = n * (768 * dim + 32 * 12 * dim * dim + dim * 256 + 9 * 256 * 256)
        att = 4 * heads * (n * n * 2 * hd + n * 128 * hd) + 28 * 25 * heads * (196 * 196 * 2 * hd + 196 * 28 * hd)
        fl = 2.0 * (lin + att)
        ref_out = None
        for prec, key in (("f16x3", "sam_vit_h_encoder"), ("f16", "sam_vit_h_encoder_f16")):
            enc.precision = prec   # f16x3: fp32-level results; f16: config 5's dtype, one MFMA per product
            ms, ok = timed(enc, x)
            out[key] = {"value": round(SB * 1e3 / ms, 1), "unit": "images/s", "batch": SB, "image": [1024, 1024], "dtype": prec,
                        "ms_per_image": round(ms / SB, 3), "tflops_algorithmic": round(fl * SB / ms / 1e9, 1),
                        "frac_of_f16_mfma_peak_executed": round((3 if prec == "f16x3" else 1) * fl * SB / ms / 1e9 / PEAK_F16_MFMA_TFLOPS, 4),
                        "verified": ok}
            y = enc(x[:1])
            if ref_out is None:
                ref_out = y
            else:   # the f16 mode against the f16x3 result of the same weights and image
                out[key]["max_abs_diff_vs_f16x3"] = round(float((y - ref_out).abs().max()), 5)
        del enc, x
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        out["sam_vit_h_encoder"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def strict_f32_leg(model, pipe, img0, img1, args, steps=2):
    """The same step with every contraction on the fp32-in MFMA (v_mfma_f32_32x32x2_f32: an exact k-ordered fp32 fma
    chain — the reference's arithmetic): 1 warm-up + `steps` timed steps, attention bracketed."""
    from pope_amd.profiling import KernelProfiler
    dtype_name, peak, factor = PRECISION_INFO["f32"]
    model.precision = pipe.match_precision = "f32"
    try:
        pipe(img0, img1)
        n_chunks = -(-args.pairs // args.chunk) * 2
        prof = KernelProfiler(DEPTH, n_chunks * steps, kinds=("attention",))
        torch.cuda.synchronize()
        model.profiler = prof
        t0 = time.perf_counter()
        for _ in range(steps):
            out = pipe(img0, img1)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        model.profiler = None
        v = prof.summary()["attention"]
        prof.close()
        dom = {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"],
               "tflops": round(kernel_flops("attention", args.chunk) / v["avg_ms"] / 1e9, 2)}
        value = args.pairs * steps / elapsed
        return {"value": round(value, 2), "unit": "image-pairs/s", "dtype": dtype_name, "steps": steps, "warmup": 1,
                "ms_per_step": round(1e3 * elapsed / steps, 3),
                "achieved_tflops_whole_step": round(value * flops_per_pair() / 1e12, 2),
                "frac_of_f32_mfma_peak_whole_step": round(value * flops_per_pair() / 1e12 / peak, 4),
                "matches_per_pair_mean": round(float(out["counts"].float().mean()), 1),
                "roofline": roofline_entry("attention", dom, peak, factor, args.chunk)}
    finally:
        model.profiler = None
        model.precision = args.precision
        pipe.match_precision = "f16x3" if args.precision == "f16" else args.precision


if __name__ == "__main__":
    main()
