"""Execution plans: packed weights + cached workspaces + the launch sequence of
the HIP kernels for the ViT backbone and the ProbMapHead.

A plan is built from an ``nn.Module``'s parameters (the modules in
backbone.py / head.py only hold parameters and structure, mirroring the
reference's attribute names) and re-built when the parameters change.  All
activations stay on the device as channels-last rows; the residual stream is
fp32, GEMM operands are ``dtype`` (bf16: ``v_mfma_f32_16x16x32_bf16``; fp32:
exact ``v_mfma_f32_16x16x4_f32``).
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib, ops, pack
from .ops import EPI_GELU, EPI_OUT_F32, EPI_RELU

_PLANS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()

# bench.py's kernel-timing pass sets this to run the head's two branches back to back on one stream,
# so that per-launch event timings are not inflated by the other branch sharing the CUs
SERIALIZE_HEAD = False

# Run the ViT blocks of a batch as two half-batch kernel chains on two HIP streams (see VitPlan.forward_tokens).
# fp8 mode: also run attn.proj on the fp8 MFMA (attention then writes e4m3 with a static per-tensor scale).  Off: on the
# synthetic ViT-B it buys +2.5 % throughput but moves the decoded keypoints by a median of 4 px against the bf16 path
# (0.3 px without it): one scale per tensor is too coarse for the attention output (tools/fp8_deviation.py).
FP8_PROJ = False
DUAL_CHAIN = False
DUAL_CHAIN_MIN_BATCH = 16
# aux stages >= 1 (M = 16 B / 4 B rows, K = 9 C) as split-K launches with the reduction folded into the pooling kernel
SPLITK_AUX = True
# bf16 heads with K <= 32 keypoints: the last deconvolution's epilogue applies the final 1x1 layer (see HeadPlan._forward)
FUSE_FINAL = True
# head-major q / k / v for head dims that are not whole cache lines (see VitPlan.__init__)
HEADMAJOR_QKV = True


def _signature(module: torch.nn.Module):
    sig = []
    for t in list(module.parameters()) + list(module.buffers()):
        sig.append((t.data_ptr(), t._version, t.dtype, t.device))
    return tuple(sig)


def plan_for(module, builder, dtype: torch.dtype, device):
    """Cached plan of ``module`` for (dtype, device); rebuilt when any parameter changed."""
    entry = _PLANS.get(module)
    key = (dtype, str(device))
    sig = _signature(module)
    if entry is None or entry[0] != key or entry[1] != sig:
        entry = (key, sig, builder(module, dtype, device))
        _PLANS[module] = entry
    return entry[2]


def _dev(t: torch.Tensor, device, dtype=None):
    t = t.detach()
    return t.to(device=device, dtype=dtype if dtype is not None else t.dtype).contiguous()


class _Workspace:
    """Named scratch tensors.  Every buffer is row-major [rows, ...]: one allocation per name serves every batch
    size up to the largest seen (smaller batches get the leading rows: contiguous views at a stable address, so
    captured graphs stay valid).  A larger batch allocates a new buffer; the old one is kept alive (a graph captured
    at the old size still points into it) but is no longer handed out."""

    def __init__(self):
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self._in_graph = set()        # keys handed out while a graph was being captured
        self._retired = []

    def get(self, name: str, shape, dtype, device) -> torch.Tensor:
        shape = tuple(shape)
        key = (name, shape[1:], dtype)
        t = self._bufs.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if t is None or t.shape[0] < shape[0]:
            if t is not None and key in self._in_graph:
                self._retired.append(t)      # a captured graph points into it; buffers no graph has seen are freed
                self._in_graph.discard(key)
            t = torch.empty(shape, dtype=dtype, device=device)
            self._bufs[key] = t
        if capturing:
            self._in_graph.add(key)
        return t[: shape[0]]


# ---------------------------------------------------------------------------
# ViT backbone
# ---------------------------------------------------------------------------
class VitPlan:
    def __init__(self, vit, dtype: torch.dtype, device, fp8: bool = False):
        self.dtype, self.device, self.fp8 = dtype, device, fp8
        pe = vit.patch_embed
        self.patch = int(pe.patch_size[0])
        self.img_size = tuple(pe.img_size)
        self.C = C = vit.embed_dim
        self.heads = vit.num_heads
        self.hd = C // self.heads
        self.N = pe.num_patches
        self.pe_w = _dev(pe.proj.weight.reshape(C, -1), device, dtype)
        self.pe_b = _dev(pe.proj.bias, device, torch.float32)
        self.pos = _dev(vit.pos_embed.reshape(self.N, C), device, torch.float32)
        self.blocks = []
        for blk in vit.blocks:
            self.blocks.append(dict(
                n1w=_dev(blk.norm1.weight, device, torch.float32), n1b=_dev(blk.norm1.bias, device, torch.float32),
                eps1=blk.norm1.eps,
                qkv_w=_dev(blk.attn.qkv.weight, device, dtype), qkv_b=_dev(blk.attn.qkv.bias, device, torch.float32),
                proj_w=_dev(blk.attn.proj.weight, device, dtype), proj_b=_dev(blk.attn.proj.bias, device, torch.float32),
                n2w=_dev(blk.norm2.weight, device, torch.float32), n2b=_dev(blk.norm2.bias, device, torch.float32),
                eps2=blk.norm2.eps,
                fc1_w=_dev(blk.mlp.fc1.weight, device, dtype), fc1_b=_dev(blk.mlp.fc1.bias, device, torch.float32),
                fc2_w=_dev(blk.mlp.fc2.weight, device, dtype), fc2_b=_dev(blk.mlp.fc2.bias, device, torch.float32),
            ))
        if fp8:
            # qkv / proj / fc1 / fc2 on fp8 MFMA: e4m3 weights, one scale per output channel (row of W); the activation
            # scales are static per tensor and come from a calibration pass on the first batch (_calibrate_fp8)
            for b, blk in zip(self.blocks, vit.blocks):
                for name, lin in (("qkv", blk.attn.qkv), ("proj", blk.attn.proj), ("fc1", blk.mlp.fc1),
                                  ("fc2", blk.mlp.fc2)):
                    w8, sw = ops.quantize_rows_fp8(lin.weight.detach().to(device))
                    b[name + "_w8"], b[name + "_sw"] = w8, sw
            self.fp8_calibrated = False
        self.nw = _dev(vit.norm.weight, device, torch.float32)
        self.nb = _dev(vit.norm.bias, device, torch.float32)
        self.neps = vit.norm.eps
        self.hidden = self.blocks[0]["fc1_w"].shape[0] if self.blocks else 4 * C
        # bf16 models whose head rows are not whole 128-byte lines (head_dim 80: ViT-H; 32 with N > 192) and that run the
        # streaming attention kernel: the qkv GEMM writes q / k / v head-major, so one head's rows are contiguous
        self.headmajor = (HEADMAJOR_QKV and dtype == torch.bfloat16 and not fp8 and
                          (self.hd == 80 or (self.hd == 32 and self.N > 192)))
        self.ws = _Workspace()
        self._chain_stream = None

    def forward_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """x (B,3,H,W) f32 on the device -> tokens [B*N, C] in the compute dtype
        (= channels-last (B,gh,gw,C) feature map), final LayerNorm applied."""
        B, ch, H, W = x.shape
        if ch != 3 or (H, W) != self.img_size:
            raise AssertionError(f"Input size ({H}, {W}) doesn't match model {self.img_size}")
        with torch.cuda.device(self.device):      # kernels launch on the current stream of the plan's device
            return self._forward_tokens(x, B)

    def _forward_tokens(self, x: torch.Tensor, B: int) -> torch.Tensor:
        dev, dt, C, N = self.device, self.dtype, self.C, self.N
        M = B * N
        g = self.ws.get
        a0 = g("a0", (M, 3 * self.patch ** 2), dt, dev)
        xres = g("xres", (M, C), torch.float32, dev)
        h = g("h", (M, C), dt, dev)
        qkv = g("qkv", (M, 3 * C), dt, dev)
        ao = g("ao", (M, C), dt, dev)
        hid = g("hid", (M, self.hidden), dt, dev)
        feats = g("feats", (M, C), dt, dev)
        bufs = (a0, xres, h, qkv, ao, hid, feats)
        if self.fp8:
            if not self.fp8_calibrated:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("fp8 mode calibrates its activation scales on the first batch: run one eager "
                                       "forward before capturing a graph")
                self._calibrate_fp8(x, B, bufs)
            h8 = g("h8", (M, C), ops.FP8, dev)
            ao8 = g("ao8", (M, C), ops.FP8, dev)
            hid8 = g("hid8", (M, self.hidden), ops.FP8, dev)
            self._run_chain_fp8(x, B, (a0, xres, h8, qkv, ao8, hid8, feats))
            return feats
        if DUAL_CHAIN and not SERIALIZE_HEAD and B % 2 == 0 and B >= DUAL_CHAIN_MIN_BATCH:
            # Two half-batches as two independent kernel chains on two HIP streams (row slices of the same
            # buffers, same weights).  Every kernel of the path is bulk-synchronous: all its workgroups
            # run their matrix phase together and then their store phase together, so one chain alone
            # alternates between an idle HBM and an idle matrix pipe.  The second chain starts one
            # GEMM later, so its memory phases fall into the first chain's matrix phases, and each
            # chain's tail wave is filled by the other chain's next kernel.
            if self._chain_stream is None:
                self._chain_stream = torch.cuda.Stream(device=dev)
            cur, s2 = torch.cuda.current_stream(dev), self._chain_stream
            Bh, Mh = B // 2, M // 2
            offset = torch.cuda.Event()
            self._run_chain(x[:Bh], Bh, tuple(t[:Mh] for t in bufs), offset)
            s2.wait_event(offset)
            with torch.cuda.stream(s2):
                self._run_chain(x[Bh:], Bh, tuple(t[Mh:] for t in bufs))
            cur.wait_stream(s2)
        else:
            self._run_chain(x, B, bufs)
        return feats

    def _run_chain(self, x, B, bufs, offset_event=None):
        a0, xres, h, qkv, ao, hid, feats = bufs
        C, N = self.C, self.N
        M = B * N
        ops.patchify(x, a0, self.patch)
        # patch_embed.proj as a GEMM, + bias, + pos_embed (row m uses pos[m % N]), fp32 residual stream
        ops.gemm(a0, self.pe_w, xres, M=M, N=C, Kd=a0.shape[1], lda=a0.shape[1], ldw=a0.shape[1], ldc=C,
                 bias=self.pe_b, rowbias=self.pos, rowbias_period=N, epilogue=EPI_OUT_F32)
        if offset_event is not None and not self.blocks:
            offset_event.record()
        for i, b in enumerate(self.blocks):
            ops.layernorm(xres, b["n1w"], b["n1b"], b["eps1"], h)
            if i == 0 and offset_event is not None:
                offset_event.record()      # the other chain starts here: one patch-embed GEMM + LN behind
            hm = (self.heads, self.hd) if self.headmajor else None
            ops.linear(h, b["qkv_w"], b["qkv_b"], out=qkv, headmajor=hm)
            ops.attention(qkv, ao, B, N, self.heads, self.hd, headmajor=self.headmajor)
            ops.linear(ao, b["proj_w"], b["proj_b"], out=xres, residual=xres)
            ops.layernorm(xres, b["n2w"], b["n2b"], b["eps2"], h)
            ops.linear(h, b["fc1_w"], b["fc1_b"], out=hid, epilogue=EPI_GELU)
            ops.linear(hid, b["fc2_w"], b["fc2_b"], out=xres, residual=xres)
        ops.layernorm(xres, self.nw, self.nb, self.neps, feats)


FP8_MARGIN = 1.25      # head-room of the static activation scales over the calibration amax (e4m3 saturates at 448)


def _vit_calibrate_fp8(self, x, B, bufs, margin: float = None):
    """One bf16 pass over the batch recording amax of the four quantised activations of every block
    (LN1 output, attention output, LN2 output, GELU output) -> static scales margin * amax / 448, and the per-column
    dequantisation vectors (activation scale x weight-row scale) of the fp8 GEMMs.  Repeated calls
    (VitPlan.calibrate over several batches) keep the running maximum."""
    margin = FP8_MARGIN if margin is None else margin
    a0, xres, h, qkv, ao, hid, feats = bufs
    C, N = self.C, self.N
    M = B * N
    ops.patchify(x, a0, self.patch)
    ops.gemm(a0, self.pe_w, xres, M=M, N=C, Kd=a0.shape[1], lda=a0.shape[1], ldw=a0.shape[1], ldc=C,
             bias=self.pe_b, rowbias=self.pos, rowbias_period=N, epilogue=EPI_OUT_F32)
    def amax(t, blk, key):
        a = max(float(t.float().abs().max()), 1e-6, blk.get("amax_" + key, 0.0))
        blk["amax_" + key] = a
        return a * margin / ops.FP8_MAX

    for b in self.blocks:
        ops.layernorm(xres, b["n1w"], b["n1b"], b["eps1"], h)
        b["s_h1"] = amax(h, b, "h1")
        ops.linear(h, b["qkv_w"], b["qkv_b"], out=qkv)
        ops.attention(qkv, ao, B, N, self.heads, self.hd)
        b["s_ao"] = amax(ao, b, "ao")
        ops.linear(ao, b["proj_w"], b["proj_b"], out=xres, residual=xres)
        ops.layernorm(xres, b["n2w"], b["n2b"], b["eps2"], h)
        b["s_h2"] = amax(h, b, "h2")
        ops.linear(h, b["fc1_w"], b["fc1_b"], out=hid, epilogue=EPI_GELU)
        b["s_hid"] = amax(hid, b, "hid")
        ops.linear(hid, b["fc2_w"], b["fc2_b"], out=xres, residual=xres)
        b["qkv_cs"] = (b["qkv_sw"] * b["s_h1"]).contiguous()
        b["proj_cs"] = (b["proj_sw"] * b["s_ao"]).contiguous()
        b["fc1_cs"] = (b["fc1_sw"] * b["s_h2"]).contiguous()
        b["fc2_cs"] = (b["fc2_sw"] * b["s_hid"]).contiguous()
    self.fp8_calibrated = True


def _vit_run_chain_fp8(self, x, B, bufs):
    a0, xres, h8, qkv, ao8, hid8, feats = bufs
    C, N = self.C, self.N
    M = B * N
    ops.patchify(x, a0, self.patch)
    ops.gemm(a0, self.pe_w, xres, M=M, N=C, Kd=a0.shape[1], lda=a0.shape[1], ldw=a0.shape[1], ldc=C,
             bias=self.pe_b, rowbias=self.pos, rowbias_period=N, epilogue=EPI_OUT_F32)
    for b in self.blocks:
        ops.layernorm(xres, b["n1w"], b["n1b"], b["eps1"], h8, out_scale=b["s_h1"])
        ops.linear(h8, b["qkv_w8"], b["qkv_b"], out=qkv, colscale=b["qkv_cs"])
        if FP8_PROJ:
            ops.attention(qkv, ao8, B, N, self.heads, self.hd, out_scale=b["s_ao"])
            ops.linear(ao8, b["proj_w8"], b["proj_b"], out=xres, residual=xres, colscale=b["proj_cs"])
        else:
            ao = self.ws.get("ao", (M, C), self.dtype, self.device)
            ops.attention(qkv, ao, B, N, self.heads, self.hd)
            ops.linear(ao, b["proj_w"], b["proj_b"], out=xres, residual=xres)
        ops.layernorm(xres, b["n2w"], b["n2b"], b["eps2"], h8, out_scale=b["s_h2"])
        ops.linear(h8, b["fc1_w8"], b["fc1_b"], out=hid8, epilogue=EPI_GELU, colscale=b["fc1_cs"],
                   out_scale=b["s_hid"])
        ops.linear(hid8, b["fc2_w8"], b["fc2_b"], out=xres, residual=xres, colscale=b["fc2_cs"])
    ops.layernorm(xres, self.nw, self.nb, self.neps, feats)





def _vit_calibrate(self, batches, margin: float = None):
    """Explicit fp8 calibration over representative batches (iterable of (B,3,H,W) device tensors): the static
    activation scales become margin * (max |activation| over all batches) / 448.  Without it the first forward
    calibrates on its own batch."""
    if not self.fp8:
        raise RuntimeError("calibrate() applies to the fp8 compute mode")
    for b in self.blocks:
        for k in [k for k in b if k.startswith("amax_")]:
            del b[k]
    with torch.cuda.device(self.device), torch.no_grad():
        for x in batches:
            x = x.detach().contiguous().float()
            B = x.shape[0]
            M = B * self.N
            g = self.ws.get
            bufs = (g("a0", (M, 3 * self.patch ** 2), self.dtype, self.device), g("xres", (M, self.C), torch.float32, self.device),
                    g("h", (M, self.C), self.dtype, self.device), g("qkv", (M, 3 * self.C), self.dtype, self.device),
                    g("ao", (M, self.C), self.dtype, self.device), g("hid", (M, self.hidden), self.dtype, self.device),
                    g("feats", (M, self.C), self.dtype, self.device))
            self._calibrate_fp8(x, B, bufs, margin)
    return self


VitPlan.calibrate = _vit_calibrate
VitPlan._calibrate_fp8 = _vit_calibrate_fp8
VitPlan._run_chain_fp8 = _vit_run_chain_fp8


def build_vit_plan(vit, dtype, device):
    if dtype == ops.FP8:
        return VitPlan(vit, torch.bfloat16, device, fp8=True)
    return VitPlan(vit, dtype, device)


# ---------------------------------------------------------------------------
# ProbMapHead
# ---------------------------------------------------------------------------
AUX_NAMES = ("probability", "visibility", "oks", "error")


class HeadPlan:
    def __init__(self, head, dtype: torch.dtype, device):
        from torch import nn
        self.dtype, self.device = dtype, device
        self.K = head.out_channels
        self.C = head.in_channels
        self.temperature = float(head.temperature)
        # normalize != None: Sparsemax over H*W, * normalize, clamp (head.py:237-245,526-532) as pp_sparsemax_rows on
        # the unclamped logits; the third-party library is not in the reference checkout -> parity unpinned
        self.normalize = None if head.normalize is None else float(head.normalize)
        # --- heatmap branch: deconvs (+BN+ReLU), optional convs (+BN+ReLU), final conv
        self.deconvs = []
        cin = self.C
        if not isinstance(head.deconv_layers, nn.Identity):
            layers = list(head.deconv_layers)
            for i in range(0, len(layers), 3):
                dc, bn = layers[i], layers[i + 1]
                k = int(dc.kernel_size[0])
                wf, bf = pack.fold_bn(dc.weight.detach().float().cpu(), None, bn.weight.detach().cpu(),
                                      bn.bias.detach().cpu(), bn.running_mean.cpu(), bn.running_var.cpu(),
                                      bn.eps, out_dim=1)
                self.deconvs.append(dict(k=k, cin=cin, cout=dc.out_channels,
                                         w=_dev(pack.pack_deconv_parities(wf, k), device, dtype),
                                         b=_dev(bf, device, torch.float32)))
                cin = dc.out_channels
        self.convs = []
        if not isinstance(head.conv_layers, nn.Identity):
            layers = list(head.conv_layers)
            for i in range(0, len(layers), 3):
                cv, bn = layers[i], layers[i + 1]
                wf, bf = pack.fold_bn(cv.weight.detach().float().cpu(),
                                      None if cv.bias is None else cv.bias.detach().cpu(),
                                      bn.weight.detach().cpu(), bn.bias.detach().cpu(), bn.running_mean.cpu(),
                                      bn.running_var.cpu(), bn.eps)
                self.convs.append(dict(k=int(cv.kernel_size[0]), pad=int(cv.padding[0]), cin=cin,
                                       cout=cv.out_channels, w=_dev(pack.conv_taps_major(wf), device, dtype),
                                       b=_dev(bf, device, torch.float32)))
                cin = cv.out_channels
        if isinstance(head.final_layer, nn.Identity):
            # final_layer_kernel_size=None (head.py:234-235): the last conv / deconv layer's (ReLU'd) channels ARE the maps
            if cin != self.K:
                raise ValueError(f"final_layer_kernel_size=None: the heatmap branch ends with {cin} channels but the "
                                 f"head has out_channels={self.K} (Codec.decode and the aux branches expect K maps)")
            self.final = None
        else:
            fl = head.final_layer
            self.final = dict(k=int(fl.kernel_size[0]), pad=int(fl.padding[0]), cin=cin,
                              w=_dev(pack.conv_taps_major(fl.weight.detach().float().cpu()), device, dtype),
                              b=_dev(fl.bias.detach().float(), device, torch.float32))
        # the implicit-GEMM layers walk their input one K-tile (128 bytes of channels) at a time inside a tap: every layer
        # INPUT width of the heatmap branch (and the aux branches' C) must be a whole number of K-tiles
        gran = 64 if dtype == torch.bfloat16 else 32
        widths = [self.C] + [d["cout"] for d in self.deconvs] + [c["cout"] for c in self.convs]
        consumed = widths[: len(self.deconvs) + len(self.convs) + (1 if self.final is not None else 0)]
        bad = [w for w in consumed if w % gran]
        if bad:
            raise ValueError(f"ProbMapHead on the HIP path ({dtype}): layer input widths {bad} are not multiples of {gran} "
                             f"channels (in_channels / deconv_out_channels / conv_out_channels feeding another layer); "
                             "the reference's own configurations use 256-channel layers")
        # --- four aux branches: [conv3x3+BN, pool, relu] x n -> conv1x1 -> act
        prob_layers = list(head.probability_layers)
        n_stage = (len(prob_layers) - 2) // 4
        self.pools = []
        for i in range(n_stage):
            ks = prob_layers[4 * i + 2].kernel_size
            self.pools.append((int(ks), int(ks)) if isinstance(ks, int) else (int(ks[0]), int(ks[1])))
        n = len(self.pools)
        C = self.C
        stage_w = [[] for _ in range(n)]
        stage_b = [[] for _ in range(n)]
        tail_w, tail_b = [], []
        for name in AUX_NAMES:
            layers = list(getattr(head, name + "_layers"))
            assert len(layers) == 4 * n + 2
            for i in range(n):
                cv, bn = layers[4 * i], layers[4 * i + 1]
                wf, bf = pack.fold_bn(cv.weight.detach().float().cpu(), cv.bias.detach().cpu(),
                                      bn.weight.detach().cpu(), bn.bias.detach().cpu(), bn.running_mean.cpu(),
                                      bn.running_var.cpu(), bn.eps)
                stage_w[i].append(pack.conv_taps_major(wf))
                stage_b[i].append(bf)
            last = layers[4 * n]
            tail_w.append(last.weight.detach().float().cpu().reshape(self.K, C))
            tail_b.append(last.bias.detach().float().cpu())
        # stage 0 shares its input across branches -> one GEMM with N = 4C; later stages batch = 4
        self.aux_w = [_dev(torch.cat(stage_w[0], 0), device, dtype)] + \
                     [_dev(torch.stack(stage_w[i]), device, dtype) for i in range(1, n)]
        self.aux_b = [_dev(torch.cat(stage_b[0], 0), device, torch.float32)] + \
                     [_dev(torch.stack(stage_b[i]), device, torch.float32) for i in range(1, n)]
        self.tail_w = _dev(torch.stack(tail_w), device, dtype)
        self.tail_b = _dev(torch.stack(tail_b), device, torch.float32)
        self.ws = _Workspace()
        self._tables: Dict[tuple, dict] = {}
        self._aux_stream = None

    # gather/scatter tables depend on (B, h, w) only
    def _tables_for(self, B: int, h: int, w: int) -> dict:
        key = (B, h, w)
        capturing = torch.cuda.is_current_stream_capturing()
        t = self._tables.get(key)
        if t is not None:
            self._tables[key] = self._tables.pop(key)          # most recently used last
            t["pinned"] = t["pinned"] or capturing
            return t
        # bounded: batch sizes vary per frame in inference.  The tables are device tensors whose ADDRESSES a captured
        # graph has baked in, so an entry that was handed out during a capture is pinned and never evicted (a replay
        # would otherwise gather through freed memory); only eager-only entries age out.
        evictable = [k for k, v in self._tables.items() if not v["pinned"]]
        while len(evictable) >= 8:
            self._tables.pop(evictable.pop(0))
        dev = self.device
        t = dict(deconv=[], conv=[], aux=[], pinned=capturing)
        hh, ww = h, w
        for d in self.deconvs:
            ro, rm = pack.deconv_tables(B, hh, ww, d["k"], d["cin"])
            t["deconv"].append((ro.to(dev), rm.to(dev), hh, ww))
            hh, ww = 2 * hh, 2 * ww
        for c in self.convs:
            t["conv"].append(pack.conv_gather_table(B, hh, ww, c["k"], c["k"], c["pad"], c["pad"], c["cin"]).to(dev)
                             if c["k"] > 1 else None)
        f = self.final
        t["final"] = (pack.conv_gather_table(B, hh, ww, f["k"], f["k"], f["pad"], f["pad"], f["cin"]).to(dev)
                      if f is not None and f["k"] > 1 else None)
        t["hm_hw"] = (hh, ww)
        ah, aw = h, w
        for i, p in enumerate(self.pools):
            stride = self.C if i == 0 else 4 * self.C
            t["aux"].append((pack.conv_gather_table(B, ah, aw, 3, 3, 1, 1, stride).to(dev), ah, aw))
            _, _, ah, aw = pack.pool_out(ah, aw, p)
        if (ah, aw) != (1, 1):
            raise ValueError(
                f"alt_head_kernel_sizes {self.pools} reduce a {h}x{w} feature map to {ah}x{aw}; the aux "
                "branches must end at 1x1 (Codec.decode reshapes them to (B,1,K), reference codec.py:254-257)")
        self._tables[key] = t
        return t

    def forward(self, feats: torch.Tensor, B: int, h: int, w: int):
        """feats: channels-last rows [B*h*w, C] in the compute dtype.
        Returns (heatmaps (B,K,Hh,Wh) f32, prob, vis, oks, err (B,K,1,1) f32)."""
        with torch.cuda.device(self.device):      # kernels launch on the current stream of the plan's device
            return self._forward(feats, B, h, w)

    def _forward(self, feats: torch.Tensor, B: int, h: int, w: int):
        dev, dt, C, K = self.device, self.dtype, self.C, self.K
        tb = self._tables_for(B, h, w)
        g = self.ws.get
        # ---- aux branches on a second HIP stream: after their first conv they are long-K GEMMs over a
        # few rows (M = B*16, B*4) that cannot fill 256 CUs; the independent deconvolution branch fills
        # the idle CUs meanwhile.  Fork/join by events, so the pair is graph-capturable.
        if self._aux_stream is None:
            self._aux_stream = torch.cuda.Stream(device=dev)
        aux = torch.empty((4, B, K), dtype=torch.float32, device=dev)
        heat = torch.empty((B, K, h * 2 ** len(self.deconvs), w * 2 ** len(self.deconvs)), dtype=torch.float32,
                           device=dev)
        aux_stream = torch.cuda.current_stream(dev) if SERIALIZE_HEAD else self._aux_stream
        aux_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(aux_stream):
            a, ah, aw = feats, h, w
            for i, (p, (ro, _, _)) in enumerate(zip(self.pools, tb["aux"])):
                M = B * ah * aw
                conv = g(f"aux_conv{i}", (M, 4 * C), dt, dev)
                if i == 0:
                    ops.gemm(a, self.aux_w[0], conv, M=M, N=4 * C, Kd=9 * C, lda=C, ldw=9 * C, ldc=4 * C,
                             bias=self.aux_b[0], rowoff=ro, seg_len=C)
                    split = 1
                else:
                    # few rows (16 B, then 4 B) against K = 9 C: split the taps over workgroups when the plain launch
                    # would leave most CUs idle behind 108 sequential K-tiles (S f32 partials, summed in the pooling)
                    # The split is a property of the LAYER (3 for the first pooled stage, 9 from the second on), never
                    # of the batch size: a crop's outputs stay bit-identical whatever batch it is computed in (the
                    # partial-sum order would otherwise change with B).  At large batches the partials cost ~0.2 % of
                    # the step.
                    split = (3 if i == 1 else 9) if SPLITK_AUX else 1
                    if split == 1:
                        ops.gemm(a, self.aux_w[i], conv, M=M, N=C, Kd=9 * C, lda=4 * C, ldw=9 * C, ldc=4 * C,
                                 bias=self.aux_b[i], rowoff=ro, seg_len=C, batch=4, strideA=C, strideW=C * 9 * C,
                                 strideC=C, strideBias=C)
                    else:
                        # rows leading ([split * M, 4C], partial s = rows [s * M, (s + 1) * M)): the workspace key does
                        # not hold M, so one allocation serves every batch size up to the largest seen
                        parts = g(f"aux_part{i}", (split * M, 4 * C), torch.float32, dev).view(split, M, 4 * C)
                        taps = 9 // split
                        ops.gemm(a, self.aux_w[i], parts, M=M, N=C, Kd=taps * C, lda=4 * C, ldw=9 * C, ldc=4 * C,
                                 rowoff=ro, seg_len=C, batch=4, strideA=C, strideW=C * 9 * C, strideC=C,
                                 epilogue=EPI_OUT_F32, splitk=split, strideW_k=taps * C, strideRowoff_k=taps * M,
                                 strideC_k=M * 4 * C)
                kh, kw, oh, ow = pack.pool_out(ah, aw, p)
                pooled = g(f"aux_pool{i}", (B * oh * ow, 4 * C), dt, dev)
                if split > 1:
                    ops.maxpool_relu_sum(parts, self.aux_b[i].reshape(-1), pooled, B, ah, aw, 4 * C, kh, kw)
                else:
                    ops.maxpool_relu(conv, pooled, B, ah, aw, 4 * C, kh, kw)
                a, ah, aw = pooled, oh, ow
            ops.aux_tail(a, self.tail_w, self.tail_b, aux, B, C, K)
        # ---- heatmap branch
        x, hh, ww, cin = feats, h, w, C
        f = self.final
        clamp = self.normalize is None
        fused_final = final_done = False
        for li, (d, (ro, rm, _, _)) in enumerate(zip(self.deconvs, tb["deconv"])):
            M = B * hh * ww
            last = li == len(self.deconvs) - 1
            if (FUSE_FINAL and last and not self.convs and f is not None and dt == torch.bfloat16 and d["cout"] == 256
                    and f["k"] == 1 and K <= 32):
                # the last deconvolution's epilogue applies the final 1x1 layer itself: its 256-channel output
                # (100 MB at bs 64) is never stored, and the final-layer launch is gone
                ops.gemm(x, d["w"], heat, M=M, N=256, Kd=4 * cin, lda=cin, ldw=4 * cin, ldc=256, bias=d["b"],
                         rowoff=ro, seg_len=cin, out_rowmap=rm, batch=4, strideW=256 * 4 * cin, strideRowoff=4 * M,
                         strideRowmap=M, epilogue=EPI_RELU,
                         fuse_final=(f["w"], f["b"], K, 4 * hh * ww, self.temperature, clamp))
                fused_final = True
                hh, ww, cin = 2 * hh, 2 * ww, 256
                break
            if f is None and last and not self.convs:
                # Identity final layer: this layer's ReLU'd outputs are the maps -> /T, clamp, NCHW store in its epilogue
                ops.gemm(x, d["w"], heat, M=M, N=K, Kd=4 * cin, lda=cin, ldw=4 * cin, ldc=K, bias=d["b"], rowoff=ro,
                         seg_len=cin, out_rowmap=rm, batch=4, strideW=K * 4 * cin, strideRowoff=4 * M, strideRowmap=M,
                         strideBias=0, epilogue=EPI_RELU, heatmap=(K, 4 * hh * ww, self.temperature, clamp))
                final_done = True
                hh, ww, cin = 2 * hh, 2 * ww, K
                break
            out = g(f"deconv{li}", (4 * M, d["cout"]), dt, dev)
            ops.gemm(x, d["w"], out, M=M, N=d["cout"], Kd=4 * cin, lda=cin, ldw=4 * cin, ldc=d["cout"],
                     bias=d["b"], rowoff=ro, seg_len=cin, out_rowmap=rm, batch=4,
                     strideW=d["cout"] * 4 * cin, strideRowoff=4 * M, strideRowmap=M, epilogue=EPI_RELU)
            x, hh, ww, cin = out, 2 * hh, 2 * ww, d["cout"]
        for li, (c, ro) in enumerate(zip(self.convs, tb["conv"])):
            M = B * hh * ww
            kk = c["k"] * c["k"]
            if f is None and li == len(self.convs) - 1:
                ops.gemm(x, c["w"], heat, M=M, N=K, Kd=kk * cin, lda=cin, ldw=kk * cin, ldc=K, bias=c["b"], rowoff=ro,
                         seg_len=cin, epilogue=EPI_RELU, heatmap=(K, hh * ww, self.temperature, clamp))
                final_done = True
                break
            out = g(f"conv{li}", (M, c["cout"]), dt, dev)
            ops.gemm(x, c["w"], out, M=M, N=c["cout"], Kd=kk * cin, lda=cin, ldw=kk * cin, ldc=c["cout"],
                     bias=c["b"], rowoff=ro, seg_len=cin, epilogue=EPI_RELU)
            x, cin = out, c["cout"]
        M = B * hh * ww
        assert heat.shape == (B, K, hh, ww)
        es = 2 if dt == torch.bfloat16 else 4
        if f is None and not final_done:
            # no deconv, no conv, no final layer: the maps are the input features themselves (degenerate but constructible)
            nchw = torch.empty((B, K, hh, ww), dtype=torch.float32, device=dev)
            ops.tokens_to_nchw(x, nchw, B, hh * ww, K)
            heat.copy_(nchw / self.temperature)
            if clamp:
                heat.clamp_(0, 1)
        kk = f["k"] * f["k"] if f is not None else 0
        if fused_final or f is None:
            pass
        elif f["k"] == 1 and 64 * (cin * es + 16) + K * cin * es <= 150 * 1024:
            ops.final_heatmap(x, f["w"], f["b"], heat, B, hh * ww, cin, K, self.temperature, clamp=clamp)
        else:
            ops.gemm(x, f["w"], heat, M=M, N=K, Kd=kk * cin, lda=cin, ldw=kk * cin, ldc=K, bias=f["b"],
                     rowoff=tb["final"], seg_len=cin, heatmap=(K, hh * ww, self.temperature, clamp))
        if not clamp:
            ops.sparsemax_rows(heat.view(B * K, hh * ww), self.normalize)
        # ---- join the aux-branch stream (forked above)
        torch.cuda.current_stream(dev).wait_stream(aux_stream)
        return (heat, aux[0].reshape(B, K, 1, 1), aux[1].reshape(B, K, 1, 1), aux[2].reshape(B, K, 1, 1),
                aux[3].reshape(B, K, 1, 1))


def build_head_plan(head, dtype, device):
    return HeadPlan(head, dtype, device)
