#!/usr/bin/env python3
"""Headline benchmark: person-crops/sec (+ decode ms) of the ProbPose forward +
decode path, ViT-B 256x192 K=17 bf16, batch 64 per MI355X (BASELINE.json
configs[1]).  One process per GPU (rank r takes crops [r*64, (r+1)*64) of the
global batch, weak scaling); the decoded keypoints are all-gathered over RCCL
every step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...      # starts its own 8 ranks (see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus 8 ...      # or under an external launcher

Prints ONE JSON line on rank 0.  A step = patchify -> ViT -> ProbMapHead ->
fused decode (-> all-gather), inputs resident in HBM, synthetic crops and
seeded random-init weights (no network for datasets/checkpoints).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (embed_dim, depth, heads, img (H,W), K, pools, per-GPU batch)
    "vit_s": dict(C=384, depth=12, heads=12, img=(256, 192), K=17, pools=[(4, 3), (2, 2), (2, 2)], batch=64),
    "vit_b": dict(C=768, depth=12, heads=12, img=(256, 192), K=17, pools=[(4, 3), (2, 2), (2, 2)], batch=64),
    "vit_l": dict(C=1024, depth=24, heads=16, img=(256, 192), K=17, pools=[(4, 3), (2, 2), (2, 2)], batch=256),
    "vit_h_wholebody": dict(C=1280, depth=32, heads=16, img=(384, 288), K=133, pools=[(4, 3), (2, 2), (3, 3)],
                            batch=128),
}
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_F32_TFLOPS = 157.3
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 spec figure (the block-scaled MFMA path; the non-scaled fp8 MFMA runs at the bf16 rate)
PEAK_HBM_GBPS = 8000.0


def flops_per_crop(cfg) -> float:
    """Algorithmic forward FLOPs (2*MAC) per crop, SURVEY.md section 8d formulae."""
    C, depth, (H, W), K = cfg["C"], cfg["depth"], cfg["img"], cfg["K"]
    gh, gw = H // 16, W // 16
    N = gh * gw
    patch = 2 * 768 * C * N
    blocks = depth * (N * 24 * C * C + 4 * N * N * C)
    deconv = 2 * gh * gw * C * 256 * 16 + 2 * (2 * gh) * (2 * gw) * 256 * 256 * 16
    final = 2 * (4 * gh) * (4 * gw) * 256 * K
    aux, h, w = 0, gh, gw
    for p in cfg["pools"]:
        aux += 2 * h * w * 9 * C * C
        h, w = h // p[0], w // p[1]
    aux = 4 * (aux + 2 * C * K)
    return float(patch + blocks + deconv + final + aux)


def pmc_traffic():
    """HBM bytes per launch of the GEMM / decode kernels from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None, None
    with open(files[-1]) as f:
        k = json.load(f)["kernels"]
    n = b = 0
    dec = None
    for name, v in k.items():
        if any(t in name for t in ("gemm_kernel", "gemm_persist_kernel", "gemm_duo_kernel", "gemm_quad_stream_kernel")):
            n += v["launches"]
            b += v["launches"] * v["hbm_bytes_per_launch"]
        if "pp::decode_" in name and "pass_kernel" not in name and "argmax" not in name:
            dec = v["hbm_bytes_per_launch"]
    return (b / n if n else None), dec, os.path.basename(files[-1])


def measured_peaks():
    """What tools/ubench/peaks measured on this device class (profiles/*_peaks.txt): bf16 MFMA TFLOP/s on random
    operands (the chip lowers its clock under MFMA load) and HBM read / write GB/s.  Reported beside the spec
    peaks, never instead of them."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_peaks.txt")))
    if not files:
        return None
    txt = open(files[-1]).read()
    mf = [float(v) for v in re.findall(r"random operands.*?(\d+) TFLOP/s", txt)]
    rd = re.search(r"stream read\s+2048 MiB.*?(\d+) GB/s", txt)
    wr = re.search(r"stream write\s+2048 MiB.*?(\d+) GB/s", txt)
    if not mf or not rd or not wr:
        return None
    return {"mfma_bf16_random_tflops": round(sum(mf) / len(mf), 0), "hbm_read_gbps": float(rd.group(1)),
            "hbm_write_gbps": float(wr.group(1)), "source": os.path.basename(files[-1])}


def decode_at_scale(codec, K, H4, W4, device, iters: int = 20):
    """The decode kernel alone on a batch large enough to fill the chip for many rounds (B = 1024 at K = 17, 128 at
    K = 133: the sizes VERDICT r01 item 4 names), synthetic peaked heatmaps resident in HBM (the recipe of
    oracle.synthetic_heatmaps / tools/decode_ab.py: Gaussian blobs + noise, every 7th peak on a border, every 11th map
    zero), HIP events on the launch stream.  The bs-64 step's own decode (roofline_decode) is one partial round of the chip: a latency, not a rate."""
    B = 1024 if K <= 32 else 128
    g = torch.Generator(device=device).manual_seed(4321)
    yy = torch.arange(H4, device=device, dtype=torch.float32)[None, None, :, None]
    xx = torch.arange(W4, device=device, dtype=torch.float32)[None, None, None, :]
    cx = torch.rand((B, K, 1, 1), device=device, generator=g) * (W4 - 1)
    nn_ = torch.arange(B * K, device=device).reshape(B, K, 1, 1)
    cx = torch.where(nn_ % 7 == 3, torch.where((nn_ // 7) % 2 == 0, 0.0, W4 - 1.0), cx)   # every 7th peak on a border
    cy = torch.rand((B, K, 1, 1), device=device, generator=g) * (H4 - 1)
    sg = 1 + 2 * torch.rand((B, K, 1, 1), device=device, generator=g)
    amp = 0.3 + 0.7 * torch.rand((B, K, 1, 1), device=device, generator=g)
    hm = amp * torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
    hm = (hm + 0.02 * torch.rand(hm.shape, device=device, generator=g)).clamp_(0, 1)
    n = torch.arange(B * K, device=device).reshape(B, K, 1, 1)
    hm = torch.where(n % 11 == 5, torch.zeros_like(hm), hm).contiguous()     # every 11th map is an all-zero channel
    # the bare C entry point on preallocated outputs (Codec.decode_device's per-call allocations and Python time would
    # otherwise show up in a 100 us launch)
    from probpose_pytorch_amd import _lib
    from probpose_pytorch_amd.heatmap import oks_tap_table
    L = _lib.lib()
    taps, radius = oks_tap_table(K, H4, W4, sigmas_for(K))
    taps, radius = torch.from_numpy(taps).to(device), torch.from_numpy(radius).to(device)
    kpts = torch.empty((B, K, 2), dtype=torch.float64, device=device)
    scores = torch.empty((B, K), dtype=torch.float32, device=device)
    locs = torch.empty((B, K, 2), dtype=torch.float32, device=device)
    ws = torch.zeros((max(int(L.pp_decode_workspace_bytes(B, K, H4, W4)), 16),), dtype=torch.uint8, device=device)
    stream = _lib.stream_ptr()

    def call():
        rc = L.pp_decode_f32(_lib.ptr(hm), None, None, None, None, B, K, H4, W4, _lib.ptr(taps), _lib.ptr(radius),
                             float(W4 - 1), float(H4 - 1), float(4 * W4), float(4 * H4), _lib.ptr(kpts), _lib.ptr(scores),
                             _lib.ptr(locs), None, None, None, None, _lib.ptr(ws), 0, stream)
        _lib.check(rc, "pp_decode_f32")

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        call()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) * 1e-3 / iters
    nbytes = float(B * K * (H4 * W4 * 4 + 16 + 28))
    return {"bound": "hbm", "workload": f"{B} crops x {K} maps of {H4}x{W4}", "achieved": round(nbytes / t / 1e9, 1),
            "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(nbytes / t / 1e9 / PEAK_HBM_GBPS, 4),
            "us_per_launch": round(t * 1e6, 1), "bytes_per_launch": nbytes}


def sigmas_for(K: int) -> np.ndarray:
    from probpose_pytorch_amd.synthetic import torch as _t  # noqa: F401
    if K == 17:
        return np.array([.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062, .107, .107, .087, .087,
                         .089, .089])
    return np.random.default_rng(133).uniform(0.02, 0.11, K)


def build(cfg, dtype, device):
    from probpose_pytorch_amd import Codec, ProbMap
    from probpose_pytorch_amd.backbone import ScratchViTBackbone
    from probpose_pytorch_amd.head import ProbMapHead
    from probpose_pytorch_amd.model import ProbPoseModel
    from probpose_pytorch_amd.synthetic import synthetic_model_state
    H, W = cfg["img"]
    model = ProbPoseModel(ScratchViTBackbone((H, W), 16, embed_dim=cfg["C"], depth=cfg["depth"], num_heads=cfg["heads"]),
                          ProbMapHead(cfg["C"], cfg["K"], cfg["pools"], (256, 256), (4, 4), final_layer_kernel_size=1))
    sd = synthetic_model_state((H, W), 16, cfg["C"], cfg["depth"], cfg["K"], len(cfg["pools"]), (256, 256), seed=0)
    model.load_state_dict(sd)
    model = model.to(device).eval().set_compute_dtype(dtype)
    codec = Codec(ProbMap((W, H), (W // 4, H // 4), sigmas_for(cfg["K"])))
    return model, codec, sd


def _progress(msg: str) -> None:
    """One line on stderr per phase: stdout carries only the JSON line; a silent run reads as hung to the GPU box."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """(threads to use, description): os.cpu_count() bounded by the affinity mask and by the cgroup CPU quota of this
    process -- more OpenMP threads than the container may run makes every CPU matmul spin against itself."""
    total = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else total
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(float(q) / float(period) + 0.5))
    except (OSError, ValueError):
        pass
    n = max(1, min(v for v in (total, affinity, quota) if v))
    return n, f"os.cpu_count() {total}, affinity mask {affinity}, cgroup quota {quota if quota else 'none'} -> {n} threads"


def _cpu_pass(orc, sd, cfg, x, sig):
    """One forward + decode pass of the CPU path; returns (forward s, decode s)."""
    H, W = cfg["img"]
    t0 = time.perf_counter()
    with torch.no_grad():
        out = orc.model_forward(sd, x, patch=16, heads=cfg["heads"], pools=cfg["pools"])
    t1 = time.perf_counter()
    orc.codec_decode([o.numpy() for o in out], (W, H), (W // 4, H // 4), sig)
    return t1 - t0, time.perf_counter() - t1


def cpu_baseline(cfg, sd, seconds_budget: float = 60.0):
    """The reference's CPU path (restated in oracle/, pinned on the reference's goldens): plain PyTorch fp32 eval
    forward + per-crop scipy decode on this host's cores, BASELINE.md section 3's B64 line: the FULL batch of the bench
    workload (64 crops for ViT-B; fewer only if a pass would not fit the time budget -- stated in `sample`),
    torch.set_num_threads(os.cpu_count()), 1 warm-up + 3 timed passes, median; forward and decode ms/crop from the same
    passes; plus the S1 line (ViT-S, one crop: forward + decode latency, median of 5 after 2 warm-ups)."""
    from oracle import probpose_oracle as orc
    from probpose_pytorch_amd.synthetic import synthetic_crops, synthetic_model_state
    H, W = cfg["img"]
    cores, cores_how = host_cores()
    torch.set_num_threads(cores)
    _progress(f"cpu_baseline: {cores_how}")
    sig = sigmas_for(cfg["K"])
    f, d = _cpu_pass(orc, sd, cfg, synthetic_crops(2, H, W, seed=99), sig)     # calibration (also a warm-up)
    per_crop = (f + d) / 2
    n = int(max(2, min(cfg["batch"], seconds_budget / 4 / max(per_crop, 1e-3))))
    _progress(f"cpu_baseline: calibration {per_crop * 1e3:.0f} ms/crop -> {n} crops per pass, 1 warm-up + 3 timed passes")
    x = synthetic_crops(n, H, W, seed=1234)
    passes = []
    for i in range(4):                                                           # 1 warm-up, 3 timed
        passes.append(_cpu_pass(orc, sd, cfg, x, sig))
        _progress(f"cpu_baseline: pass {i} forward {passes[-1][0]:.1f} s decode {passes[-1][1]:.1f} s")
    passes = passes[1:]
    tot = sorted(f_ + d_ for f_, d_ in passes)[1]
    fwd = sorted(f_ for f_, _ in passes)[1]
    dec = sorted(d_ for _, d_ in passes)[1]
    # S1 (BASELINE.json configs[0]): single 256x192 crop, ViT-S (C 384, depth 12, 12 heads), K = 17
    s1 = dict(CONFIGS["vit_s"])
    sd1 = synthetic_model_state(s1["img"], 16, s1["C"], s1["depth"], s1["K"], len(s1["pools"]), (256, 256), seed=0)
    x1 = synthetic_crops(1, *s1["img"], seed=7)
    sig1 = sigmas_for(s1["K"])
    p1 = [_cpu_pass(orc, sd1, s1, x1, sig1) for _ in range(7)][2:]
    return {"value": round(n / tot, 3), "unit": "crops/s", "cores": cores, "kind": "port",
            "threads": f"torch.set_num_threads({cores}): {cores_how}",
            "sample": f"{n} crops = {'the full batch' if n == cfg['batch'] else 'a bounded part'} of the bench workload "
                      f"(batch {cfg['batch']}), torch fp32 CPU eval forward + per-crop scipy decode "
                      f"(oracle/probpose_oracle.py); median of 3 passes after 1 warm-up",
            "forward_ms_per_crop": round(fwd / n * 1e3, 2), "decode_ms_per_crop": round(dec / n * 1e3, 3),
            "s1_vit_s_single_crop": {"forward_ms": round(sorted(f_ for f_, _ in p1)[2] * 1e3, 2),
                                     "decode_ms": round(sorted(d_ for _, d_ in p1)[2] * 1e3, 2),
                                     "latency_ms": round(sorted(f_ + d_ for f_, d_ in p1)[2] * 1e3, 2),
                                     "runs": "median of 5 after 2 warm-ups"}}


def parity_block(cfg, sd, model, codec, x, dtype, n_crops: int = 8, fp32_steps: int = 3):
    """What the benchmarked dtype costs in accuracy and what the exact-fp32 mode costs in speed (BASELINE.json
    north_star: decoded keypoints / probabilities within 1e-4 of the CPU path).  Measured on this workload the
    exact-fp32 MFMA mode keeps every heatmap / auxiliary value within 1e-4 (max 4.3e-5) and 99.3 % of the keypoints
    within 1e-4 px, the rest within 2.2e-4 px: the sub-pixel step divides by the convolved map's curvature and
    amplifies the heatmap deviation on flat peaks (tests/test_model_gpu.py checks all 64 crops); bf16 / fp8 report
    their measured deviation, `fp32.contract_met_1e-4_px` says which side of the contract this run fell on.  The
    reference here is oracle.model_forward + codec_decode on the same crops (checker only, never the thing measured)."""
    from oracle import probpose_oracle as orc
    H, W = cfg["img"]
    sig = sigmas_for(cfg["K"])
    xs = x[:n_crops]
    with torch.no_grad():
        ref_out = orc.model_forward(sd, xs.cpu(), patch=16, heads=cfg["heads"], pools=cfg["pools"])
    ref = orc.codec_decode([o.numpy() for o in ref_out], (W, H), (W // 4, H // 4), sig)

    def deviation(out):
        dec = codec.decode(out)
        d = np.abs(dec[0][0] - ref[0][0]).max(-1)               # per keypoint, input pixels
        hm = (out[0].float().cpu() - ref_out[0]).abs()
        return {"kpt_px_median": round(float(np.median(d)), 6), "kpt_px_p90": round(float(np.percentile(d, 90)), 6),
                "kpt_px_max": round(float(d.max()), 6),
                "frac_kpts_within_1e-4_px": round(float((d <= 1e-4).mean()), 4),
                "contract_met_1e-4_px": bool((d <= 1e-4).all()),
                "peak_moved_rate": round(float((d > 2.0).mean()), 4),      # > half a heatmap cell (4 input px)
                "heatmap_abs_mean": round(float(hm.mean()), 7), "heatmap_abs_max": round(float(hm.max()), 6),
                "aux_abs_max": round(max(float(np.abs(a - b).max()) for a, b in zip(dec[1:], ref[1:])), 7)}

    res = {"reference": f"oracle.model_forward + codec_decode (CPU fp32) on the first {n_crops} crops of the batch",
           "contract": "<= 1e-4 abs (north_star); flips between near-tied pixels of the random-weight heatmaps "
                       "are counted in peak_moved_rate (see tests/test_model_gpu.py)"}
    with torch.no_grad():
        res[{torch.bfloat16: "bf16", torch.float32: "fp32"}.get(dtype, "fp8")] = deviation(model(xs))
        if dtype != torch.float32:
            model.set_compute_dtype(torch.float32)
            res["fp32"] = deviation(model(xs))
            for _ in range(2):
                codec.decode_device(model(x))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(fp32_steps):
                codec.decode_device(model(x))
            torch.cuda.synchronize()
            res["fp32_mode_crops_per_s"] = round(x.shape[0] * fp32_steps / (time.perf_counter() - t0), 1)
            model.set_compute_dtype(dtype)
    return res


def launcher_command(argv, n_ranks: int, port: int):
    """argv of the N-rank launch of this script: one process per GPU under torch.distributed.run, rendezvous on
    127.0.0.1 (the container hostname may not resolve).  ``argv`` = this script's own arguments, passed through."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(argv, n_ranks: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: THIS process has made no GPU call and makes none (the build
    runs in a child too -- loading the HIP library is the child's business); it starts the N ranks as a child process
    tree, relays their output (rank 0 prints the one JSON line) and returns their exit status.  Never an exec: a process
    that touched the GPU must not be replaced, and the children each initialise their own device."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    rc = subprocess.call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, env=env)
    if rc != 0:
        return rc
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return subprocess.call(launcher_command(argv, n_ranks, port), cwd=ROOT, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="vit_b", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode-scale", action="store_true", help="skip the decode-at-scale measurement (profiler runs: "
                    "its launches would be averaged into the step's decode kernel)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block (fp32-mode throughput + deviation "
                                                              "of the decoded results from the CPU oracle on 8 crops)")
    ap.add_argument("--single-stream", action="store_true",
                    help="no second HIP stream for the aux branches: every kernel runs alone, so a kernel trace of "
                         "this run shows the kernels' own durations (what roofline.avg_launch_us is measured on)")
    ap.add_argument("--tune-cache", default=None,
                    help="JSON file of per-shape GEMM tile winners: loaded if present, written after the warm-up "
                         "(lets a profiled run start tuned, so its trace holds no tuning launches)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # PP_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo instead of RCCL, to walk the multi-rank code path on a
    # one-GPU box (RCCL refuses two ranks on one device); never a measurement
    rehearsal = os.environ.get("PP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and rehearsal:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from probpose_pytorch_amd import ops, parallel
    from probpose_pytorch_amd.synthetic import synthetic_crops

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    B = cfg["batch"]
    H, W = cfg["img"]
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp8": torch.float8_e4m3fn}[args.dtype]
    ops.AUTOTUNE = True     # per-shape tile selection measured on this device during the warm-up
    if args.single_stream:
        from probpose_pytorch_amd import engine as _engine
        _engine.SERIALIZE_HEAD = True
    if args.tune_cache and os.path.exists(args.tune_cache):
        ops.load_tune_cache(args.tune_cache)
    if rank == 0:
        _progress(f"building {args.config} ({args.dtype}), batch {B}/GPU, world {world}")
    model, codec, sd = build(cfg, dtype, device)
    x = synthetic_crops(B, H, W, seed=1234 + rank).to(device)       # resident in HBM before timing

    def local_step():
        out = model(x)
        dec = codec.decode_device(out)
        return parallel.pack_decoded(dec)

    def step(fn):
        packed = fn()
        return parallel.all_gather_decoded(packed) if world > 1 else packed

    # ---- warm-up (also builds plans, tables, tap caches), then optional graph capture
    with torch.no_grad():
        for _ in range(max(1, min(2, args.warmup))):
            step(local_step)
        torch.cuda.synchronize()
        if args.tune_cache and rank == 0:
            ops.save_tune_cache(args.tune_cache)
        graph, static_out = None, None
        if not args.no_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                local_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = local_step()

            def replay():
                graph.replay()
                return static_out
            run_local = replay
        else:
            run_local = local_step
        for _ in range(args.warmup):
            step(run_local)

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        sync_all()
        if rank == 0:
            _progress(f"warm-up done (tiles tuned, graph {'captured' if graph is not None else 'off'}); timing {args.steps} steps")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            result = step(run_local)
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert result.shape == (B * world, cfg["K"], 7)

        # ---- kernel-level timing with HIP events on the launch stream (eager pass, same inputs)
        from probpose_pytorch_amd import engine
        prof = []
        engine.SERIALIZE_HEAD = True      # one stream: per-launch durations are the kernels' own
        ops.set_profile(prof)
        for _ in range(3):
            local_step()
        torch.cuda.synchronize()
        ops.set_profile(None)
        engine.SERIALIZE_HEAD = bool(args.single_stream)
    agg = {}
    for name, work, s, e, _info in prof:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += work
        a[2] += s.elapsed_time(e) * 1e-3
    per_step = {k: (v[0] / 3, v[1] / 3, v[2] / 3) for k, v in agg.items()}

    ms_per_step = elapsed / args.steps * 1e3
    crops_per_s = B * world * args.steps / elapsed
    if rank == 0:
        g_n, g_flops, g_t = per_step.get("gemm", (0, 0.0, 1.0))
        peak = {"bf16": PEAK_BF16_TFLOPS, "fp32": PEAK_F32_TFLOPS, "fp8": PEAK_FP8_TFLOPS}[args.dtype]
        achieved = g_flops / g_t / 1e12
        d_n, d_bytes, d_t = per_step.get("decode", (0, 0.0, 1.0))
        a_n, a_flops, a_t = per_step.get("attention", (0, 0.0, 1.0))
        g_traffic, d_traffic, traffic_src = pmc_traffic() if args.config == "vit_b" and B == 64 else (None, None, None)
        tiles = {f"M{k[0]}xN{k[1]}xK{k[2]}" + (f"x{k[3]}" if k[3] > 1 else ""): v for k, v in ops._TUNE_CACHE.items()}
        line = {
            "metric": "person_crops_per_sec", "value": round(crops_per_s, 2), "unit": "crops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ProbPose forward+decode, {args.config} {H}x{W} K={cfg['K']}, "
                                   f"batch {B}/GPU, {'HIP graph replay' if graph is not None else 'eager launches'}",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "gflop_per_crop": round(flops_per_crop(cfg) / 1e9, 2)},
            "decode_ms": round(d_t * 1e3, 4),
            "model_tflops": round(flops_per_crop(cfg) * crops_per_s / world / 1e12, 2),
            "roofline": {"bound": "mfma", "kernel": "pp::gemm_kernel (tile forms 2-10) / pp::gemm_persist_kernel (13) / pp::gemm_duo_kernel (14) / pp::gemm_quad_stream_kernel (18-20), whichever the tuner picked per shape", "achieved": round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": g_traffic,
                         "traffic_source": traffic_src, "launches_per_step": g_n, "avg_launch_us": round(g_t / max(g_n, 1) * 1e6, 2),
                         "flop_per_launch": round(g_flops / max(g_n, 1), 0)},
            "roofline_decode": {"bound": "hbm", "kernel": ("pp::decode_wave_kernel" if B * cfg["K"] > 2 * 256 * (3 if H // 4 == 64 else 1)
                                                           else "pp::decode_lds_kernel (batches up to two rounds of its workgroups)")
                                if (H // 4, W // 4) in ((64, 48), (96, 72)) else "pp::decode_lds_kernel / pp::decode_screen_kernel",
                                "achieved": round(d_bytes / d_t / 1e9, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                "frac": round(d_bytes / d_t / 1e9 / PEAK_HBM_GBPS, 4), "traffic": d_traffic,
                                "bytes_per_launch": d_bytes},
            "roofline_decode_at_scale": None if args.no_decode_scale else decode_at_scale(codec, cfg["K"], H // 4, W // 4, device),
            "attention": {"achieved_tflops": round(a_flops / a_t / 1e12, 2), "ms_per_step": round(a_t * 1e3, 3)},
            "kernel_ms_per_step": {k: round(v[2] * 1e3, 3) for k, v in per_step.items()},
            "gemm_tiles_autotuned": tiles,
        }
        mp = measured_peaks()
        if mp is not None:
            line["measured_device_peaks"] = mp
            if args.dtype == "bf16":
                line["roofline"]["frac_of_measured_mfma"] = round(achieved / mp["mfma_bf16_random_tflops"], 4)
        _progress(f"timed region done: {ms_per_step:.3f} ms/step; parity block / CPU baseline follow")
        if world == 1 and not args.no_parity:
            line["parity"] = parity_block(cfg, sd, model, codec, x, dtype)
            _progress("parity block done")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
