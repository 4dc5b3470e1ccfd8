"""GPU numerics of every HIP kernel on the forward path against a plain
PyTorch reference of the same op (float64 math on the same inputs).

fp32 mode uses v_mfma_f32_16x16x4_f32 (an exact fp32 FMA chain): tolerance
~1e-5 relative.  bf16 mode: operands are bf16 (the reference is computed from
the same bf16-rounded operands), fp32 accumulate; an fp32 output is held to
1e-4 relative of the operand scale, a bf16 output to one bf16 ulp (2^-8)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def ops(built_lib):
    assert torch.cuda.is_available()
    from probpose_pytorch_amd import ops as o
    return o


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).cuda()


def _tol(dtype, out_dtype):
    if dtype == torch.float32:
        return dict(rtol=2e-5, atol=2e-5)
    if out_dtype == torch.float32:
        return dict(rtol=1e-4, atol=1e-3)
    return dict(rtol=2 ** -7, atol=2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (192, 768, 768), (300, 200, 128), (1, 17, 256), (257, 129, 64),
                                   (384, 2304, 768), (192, 96, 64), (193, 97, 128), (400, 400, 3072)])
def test_gemm_plain_bias_tails(ops, dtype, tile, M, N, K):
    if tile in (8, 9) and dtype == torch.float32:
        pytest.skip("the 256-wide tiles are bf16-only")
    A, W = _rand((M, K), dtype, 1), _rand((N, K), dtype, 2, K ** -0.5)
    b = _rand((N,), torch.float32, 3)
    out = ops.linear(A, W, b, tile=tile)
    ref = A.double() @ W.double().t() + b.double()
    torch.testing.assert_close(out.double(), ref, **_tol(dtype, dtype))


@pytest.mark.parametrize("tile", [13, 14])
@pytest.mark.parametrize("M,N,K,epi", [(384, 768, 768, "gelu"), (1536, 2304, 768, "none"), (200, 264, 128, "relu"),
                                       (2000, 388, 1024, "resid"),
                                       # more tiles than CUs: the persistent form walks 2 - 3 tiles per workgroup (XCD-blocked
                                       # order / plain order), ragged edges, a one-K-tile stream
                                       (12288, 2304, 768, "gelu"), (3000, 4000, 128, "none"), (5000, 3080, 64, "relu")])
def test_gemm_experimental_forms_tile13_tile14(ops, tile, M, N, K, epi):
    """The persistent stream (13) and the two-workgroups-per-CU form (14): bf16 plain layers, checked like any tile."""
    dtype = torch.bfloat16
    A, W = _rand((M, K), dtype, 1), _rand((N, K), dtype, 2, K ** -0.5)
    b = _rand((N,), torch.float32, 3)
    pre = A.double() @ W.double().t() + b.double()
    if epi == "resid":
        if tile == 13:
            pytest.skip("tile 13 writes bf16 outputs only")
        res = _rand((M, N), torch.float32, 4)
        want = res.double() + pre
        ops.linear(A, W, b, out=res, residual=res, tile=tile)
        torch.testing.assert_close(res.double(), want, **_tol(dtype, torch.float32))
        return
    flag = {"gelu": ops.EPI_GELU, "relu": ops.EPI_RELU, "none": 0}[epi]
    out = ops.linear(A, W, b, epilogue=flag, tile=tile)
    ref = {"gelu": F.gelu, "relu": F.relu, "none": lambda t: t}[epi](pre)
    torch.testing.assert_close(out.double(), ref, **_tol(dtype, dtype))


@pytest.mark.parametrize("tile,M,N,K,epi", [
    (19, 768, 576, 512, "none"), (18, 768, 576, 512, "gelu"),            # one tile per workgroup, shortest K
    (19, 12288, 2304, 768, "none"), (18, 12288, 3072, 768, "gelu"),      # the ViT-B layers: 2 / 3 tiles per workgroup
    (19, 7680, 2304, 768, "relu"), (18, 7680, 2304, 1024, "nobias"),     # 320 / 360 tiles: workgroups with 1 and with 2 tiles
    (19, 192 * 67, 288 * 5, 768, "gelu"), (18, 256 * 35, 192 * 9, 3072, "none"),
    (20, 768, 512, 512, "gelu"), (20, 12288, 4096, 1024, "gelu"), (20, 192 * 40, 256 * 9, 640, "relu")])
def test_gemm_quad_stream_forms(ops, tile, M, N, K, epi):
    """Tiles 18 / 19 / 20 (pp_gemm_quad.hip): four waves, 128x96 / 96x144 / 96x128 wave tiles, as a persistent stream: a
    finished tile's packed registers are stored from the next tile's first K-tiles (before a workgroup's first tile those
    stores carry an out-of-range offset), one / several tiles per workgroup, shortest and long K."""
    dtype = torch.bfloat16
    A, W = _rand((M, K), dtype, 1), _rand((N, K), dtype, 2, K ** -0.5)
    b = None if epi == "nobias" else _rand((N,), torch.float32, 3)
    flag = {"gelu": ops.EPI_GELU, "relu": ops.EPI_RELU}.get(epi, 0)
    out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
    for _ in range(2):                                          # twice: nothing depends on what the first launch left behind
        ops.linear(A, W, b, out=out, epilogue=flag, tile=tile)
    same = ops.linear(A, W, b, epilogue=flag, tile=3)
    assert not torch.isnan(out).any()
    torch.testing.assert_close(out.float(), same.float(), rtol=2 ** -7, atol=1e-3)
    if M * N <= 1 << 24:
        pre = A.double() @ W.double().t() + (0 if b is None else b.double())
        ref = {"gelu": F.gelu, "relu": F.relu}.get(epi, lambda t: t)(pre)
        torch.testing.assert_close(out.double(), ref, **_tol(dtype, dtype))


@pytest.mark.parametrize("tile,B,N,heads,hd", [(19, 4, 192, 12, 64), (18, 4, 192, 12, 64), (19, 4, 432, 16, 72), (18, 16, 432, 16, 80), (20, 4, 432, 16, 80)])
def test_headmajor_qkv_projection_quad_stream(ops, tile, B, N, heads, hd):
    C, M = heads * hd, B * N
    x = _rand((M, C), torch.bfloat16, 1)
    W = _rand((3 * C, C), torch.bfloat16, 2, C ** -0.5)
    b = _rand((3 * C,), torch.float32, 3)
    rowmajor = ops.linear(x, W, b, tile=3)
    hmaj = torch.full((M, 3 * C), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.linear(x, W, b, out=hmaj, tile=tile, headmajor=(heads, hd))
    want = rowmajor.reshape(M, 3, heads, hd).permute(1, 2, 0, 3).contiguous()        # [3][heads][M][hd]
    torch.testing.assert_close(hmaj.reshape(3, heads, M, hd).float(), want.float(), rtol=2 ** -7, atol=1e-3)


def test_gemm_quad_forms_refuse_what_they_do_not_serve(ops):
    from probpose_pytorch_amd import _lib
    A, W = _rand((768, 512), torch.bfloat16, 1), _rand((768, 512), torch.bfloat16, 2)
    with pytest.raises(_lib.HipExtensionError, match="lab builds"):
        ops.linear(A, W, None, tile=16)                                         # the per-launch form is not in the shipped library
    res = torch.zeros((768, 768), dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.HipExtensionError, match="tiles 15"):
        ops.linear(A, W, None, out=res, residual=res, tile=18)                  # f32 residual stream
    with pytest.raises(_lib.HipExtensionError, match="tiles 15"):
        ops.linear(A.float(), W.float(), None, tile=19)                         # fp32 parity mode
    A, W = _rand((300, 512), torch.bfloat16, 1), _rand((576, 512), torch.bfloat16, 2)
    with pytest.raises(_lib.HipExtensionError, match="tiles 18"):
        ops.linear(A, W, None, tile=19)                                         # M not a whole number of tiles


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(ops, dtype):
    M, N, K = 384, 256, 256
    A, W = _rand((M, K), dtype, 1), _rand((N, K), dtype, 2, K ** -0.5)
    b = _rand((N,), torch.float32, 3)
    pre = A.double() @ W.double().t() + b.double()
    out = ops.linear(A, W, b, epilogue=ops.EPI_GELU)
    torch.testing.assert_close(out.double(), F.gelu(pre), **_tol(dtype, dtype))
    out = ops.linear(A, W, b, epilogue=ops.EPI_RELU)
    torch.testing.assert_close(out.double(), F.relu(pre), **_tol(dtype, dtype))
    res = _rand((M, N), torch.float32, 4)
    want = res.double() + pre
    ops.linear(A, W, b, out=res, residual=res)                 # in place on the fp32 residual stream
    torch.testing.assert_close(res.double(), want, **_tol(dtype, torch.float32))
    # row bias with period (pos_embed): out[m] += rb[m % P]
    P = 96
    rb = _rand((P, N), torch.float32, 5)
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm(A, W, out, M=M, N=N, Kd=K, lda=K, ldw=K, ldc=N, bias=b, rowbias=rb, rowbias_period=P,
             epilogue=ops.EPI_OUT_F32)
    want = pre + rb.double().repeat(M // P, 1)
    torch.testing.assert_close(out.double(), want, **_tol(dtype, torch.float32))


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_exact_integer_data_catches_layout_bugs(ops, dtype):
    """A = I-like / asymmetric small-integer operands: exact in both dtypes."""
    M, N, K = 256, 256, 128
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-3, 4, (M, K), generator=g).to(dtype).cuda()
    W = torch.randint(-3, 4, (N, K), generator=g).to(dtype).cuda()
    for tile in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10):
        if tile == 8 and dtype == torch.float32:
            continue
        out = ops.linear(A, W, out_dtype=torch.float32, tile=tile)
        assert torch.equal(out.double(), A.double() @ W.double().t())
    # auto-selection picks the 192x96 tile when it saves a round of workgroups (M = 16 crops x 192)
    A2 = torch.randint(-3, 4, (3072, K), generator=g).to(dtype).cuda()
    W2 = torch.randint(-3, 4, (3072, K), generator=g).to(dtype).cuda()
    assert torch.equal(ops.linear(A2, W2, out_dtype=torch.float32).double(), A2.double() @ W2.double().t())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_conv3x3_gather_batched_branches(ops, dtype):
    """Aux-branch stage >= 1: 4 branches, each a 3x3 conv on its C-wide slice of 4C-wide rows."""
    from probpose_pytorch_amd import pack
    B, h, w, C = 3, 4, 4, 64
    x = _rand((B * h * w, 4 * C), dtype, 1)
    wt = _rand((4, C, C, 3, 3), torch.float32, 2, (9 * C) ** -0.5)
    bias = _rand((4, C), torch.float32, 3)
    Wp = torch.stack([pack.conv_taps_major(wt[i].cpu()) for i in range(4)]).to(dtype).cuda()
    ro = pack.conv_gather_table(B, h, w, 3, 3, 1, 1, 4 * C).cuda()
    out = torch.empty((B * h * w, 4 * C), dtype=dtype, device="cuda")
    for tile in (1, 2, 3, 4, 5, 6, 7, 8, 10):
        if tile == 8 and dtype == torch.float32:
            continue
        out.zero_()
        ops.gemm(x, Wp, out, M=B * h * w, N=C, Kd=9 * C, lda=4 * C, ldw=9 * C, ldc=4 * C, bias=bias, rowoff=ro,
                 seg_len=C, batch=4, strideA=C, strideW=C * 9 * C, strideC=C, strideBias=C, tile=tile)
        _check_branches(x, Wp, bias, out, B, h, w, C, dtype)


def _check_branches(x, Wp, bias, out, B, h, w, C, dtype):
    for i in range(4):
        xi = x[:, i * C:(i + 1) * C].double().reshape(B, h, w, C).permute(0, 3, 1, 2)
        wi = Wp[i].double().reshape(C, 3, 3, C).permute(0, 3, 1, 2)
        ref = F.conv2d(xi, wi, bias[i].double(), padding=1).permute(0, 2, 3, 1).reshape(B * h * w, C)
        torch.testing.assert_close(out[:, i * C:(i + 1) * C].double(), ref, **_tol(dtype, dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("k", [4, 3, 2])
def test_gemm_deconv_parities_scatter(ops, dtype, k, tile):
    from probpose_pytorch_amd import pack
    if tile in (8, 9) and dtype == torch.float32:
        pytest.skip("the 256-wide tiles are bf16-only")
    B, h, w, Cin, Cout = 2, 8, 6, 64, 128
    x = _rand((B * h * w, Cin), dtype, 1)
    wt = _rand((Cin, Cout, k, k), torch.float32, 2, (4 * Cin) ** -0.5)
    bias = _rand((Cout,), torch.float32, 3)
    Wp = pack.pack_deconv_parities(wt.cpu(), k).to(dtype).cuda()
    ro, rm = pack.deconv_tables(B, h, w, k, Cin)
    M = B * h * w
    out = torch.full((4 * M, Cout), float("nan"), dtype=dtype, device="cuda")
    ops.gemm(x, Wp, out, M=M, N=Cout, Kd=4 * Cin, lda=Cin, ldw=4 * Cin, ldc=Cout, bias=bias, rowoff=ro.cuda(),
             seg_len=Cin, out_rowmap=rm.cuda(), batch=4, strideW=Cout * 4 * Cin, strideRowoff=4 * M,
             strideRowmap=M, epilogue=ops.EPI_RELU, tile=tile)
    pad, op = pack.deconv_geometry(k)
    # reference from the same (rounded) packed weights: rebuild the deconv weight from Wp
    xi = x.double().reshape(B, h, w, Cin).permute(0, 3, 1, 2)
    wr = wt.to(dtype).double()
    ref = F.relu(F.conv_transpose2d(xi, wr, bias.double(), stride=2, padding=pad, output_padding=op))
    ref = ref.permute(0, 2, 3, 1).reshape(4 * M, Cout)
    torch.testing.assert_close(out.double(), ref, **_tol(dtype, dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_heatmap_epilogue(ops, dtype):
    B, HW, Cin, K = 2, 64 * 48, 256, 17
    x = _rand((B * HW, Cin), dtype, 1)
    W = _rand((K, Cin), dtype, 2, Cin ** -0.5)
    b = _rand((K,), torch.float32, 3, 0.1)
    out = torch.empty((B, K, HW), dtype=torch.float32, device="cuda")
    ops.gemm(x, W, out, M=B * HW, N=K, Kd=Cin, lda=Cin, ldw=Cin, ldc=K, bias=b, heatmap=(K, HW, 0.5))
    pre = (x.double() @ W.double().t() + b.double()).reshape(B, HW, K).permute(0, 2, 1)
    ref = torch.clamp(pre / 0.5, 0, 1)
    torch.testing.assert_close(out.double(), ref, **_tol(dtype, torch.float32))
    assert out.min() == 0 and out.max() == 1


def test_gemm_retired_layernorm_fusion_flags_are_refused(ops):
    """The LayerNorm-fusion epilogues of round 1 (flag bits 128 / 256) were removed: the library says so."""
    from probpose_pytorch_amd import _lib
    A, W = _rand((192, 64), torch.bfloat16, 1), _rand((192, 64), torch.bfloat16, 2)
    out = torch.empty((192, 192), dtype=torch.bfloat16, device="cuda")
    for bit in (128, 256):
        with pytest.raises(_lib.HipExtensionError, match="removed"):
            ops.gemm(A, W, out, M=192, N=192, Kd=64, lda=64, ldw=64, ldc=192, epilogue=bit)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,Cin,K", [(2, 64 * 48, 256, 17), (1, 96 * 72, 256, 133), (3, 50, 64, 5)])
def test_final_heatmap_kernel(ops, dtype, B, HW, Cin, K):
    x = _rand((B * HW, Cin), dtype, 1)
    W = _rand((K, Cin), dtype, 2, Cin ** -0.5)
    b = _rand((K,), torch.float32, 3, 0.1)
    out = torch.empty((B, K, HW), dtype=torch.float32, device="cuda")
    if 64 * (Cin * x.element_size() + 16) + K * Cin * x.element_size() > 160 * 1024:
        from probpose_pytorch_amd import _lib
        with pytest.raises(_lib.HipExtensionError, match="LDS"):   # the engine routes this case to pp_gemm
            ops.final_heatmap(x, W, b, out, B, HW, Cin, K, 0.5)
        return
    ops.final_heatmap(x, W, b, out, B, HW, Cin, K, 0.5)
    pre = (x.double() @ W.double().t() + b.double()).reshape(B, HW, K).permute(0, 2, 1)
    torch.testing.assert_close(out.double(), torch.clamp(pre / 0.5, 0, 1), **_tol(dtype, torch.float32))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(384, 768), (5, 384), (64, 1280), (3, 100), (2, 4096)])
def test_layernorm(ops, dtype, rows, C):
    x = _rand((rows, C), torch.float32, 1, 3.0) + 0.5
    g, b = _rand((C,), torch.float32, 2), _rand((C,), torch.float32, 3)
    out = torch.empty((rows, C), dtype=dtype, device="cuda")
    ops.layernorm(x, g, b, 1e-6, out)
    ref = F.layer_norm(x.double(), (C,), g.double(), b.double(), 1e-6)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=2e-2)
    torch.testing.assert_close(out.double(), ref, **tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,heads,hd", [(2, 192, 12, 64), (1, 192, 12, 32), (2, 432, 2, 80), (3, 12, 2, 64),
                                          (1, 70, 3, 32), (1, 432, 3, 64), (2, 200, 2, 32), (1, 97, 2, 80),
                                          (1, 1000, 1, 80), (2, 577, 2, 64), (1, 193, 1, 80), (2, 5, 1, 80)])
def test_attention(ops, dtype, B, N, heads, hd):
    C = heads * hd
    qkv = _rand((B * N, 3 * C), dtype, 1)
    out = torch.empty((B * N, C), dtype=dtype, device="cuda")
    ops.attention(qkv, out, B, N, heads, hd)
    q, k, v = qkv.double().reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * N, C)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2 ** -6, atol=2e-2)
    torch.testing.assert_close(out.double(), ref, **tol)


@pytest.mark.parametrize("B,N,heads,hd", [(2, 192, 12, 64), (2, 432, 2, 80), (1, 577, 2, 64), (1, 300, 2, 32)])
def test_attention_output_only_8_byte_aligned(ops, B, N, heads, hd):
    """The bf16 kernels store whole output lines 16 bytes per lane when `out` is 16-byte aligned; an `out` that is only
    8-byte aligned takes the narrower forms: same numbers either way."""
    C = heads * hd
    qkv = _rand((B * N, 3 * C), torch.bfloat16, 1)
    aligned = torch.empty((B * N, C), dtype=torch.bfloat16, device="cuda")
    ops.attention(qkv, aligned, B, N, heads, hd)
    buf = torch.full((B * N * C + 8,), float("nan"), dtype=torch.bfloat16, device="cuda")
    shifted = buf[4:4 + B * N * C].view(B * N, C)
    assert shifted.data_ptr() % 16 == 8
    ops.attention(qkv, shifted, B, N, heads, hd)
    torch.testing.assert_close(shifted.float(), aligned.float(), rtol=2 ** -6, atol=2e-2)
    assert torch.isnan(buf[:4]).all() and torch.isnan(buf[4 + B * N * C:]).all()       # nothing written outside


@pytest.mark.parametrize("tile", [0, 2, 3, 6, 7, 10, 13])
@pytest.mark.parametrize("B,N,heads,hd", [(2, 432, 16, 80), (3, 192, 12, 32), (1, 50, 4, 64), (2, 433, 4, 80)])
def test_headmajor_qkv_projection_and_attention(ops, tile, B, N, heads, hd):
    """The qkv projection written head-major ([3][heads][B*N][hd], PP_EPI_HEADMAJOR) by every tile form that takes the
    flag, then the streaming attention kernel reading that layout: (a) the head-major buffer holds exactly the
    row-major projection's numbers, (b) attention on it == attention of torch on the same q / k / v."""
    C, M = heads * hd, B * N
    x = _rand((M, C), torch.bfloat16, 1)
    W = _rand((3 * C, C), torch.bfloat16, 2, C ** -0.5)
    b = _rand((3 * C,), torch.float32, 3)
    rowmajor = ops.linear(x, W, b, tile=tile)
    hmaj = torch.full((M, 3 * C), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.linear(x, W, b, out=hmaj, tile=tile, headmajor=(heads, hd))
    want = rowmajor.reshape(M, 3, heads, hd).permute(1, 2, 0, 3).contiguous()        # [3][heads][M][hd]
    assert torch.equal(hmaj.reshape(3, heads, M, hd), want)
    out = torch.empty((M, C), dtype=torch.bfloat16, device="cuda")
    ops.attention(hmaj, out, B, N, heads, hd, headmajor=True)
    q, k, v = want.double().reshape(3, heads, B, N, hd).permute(0, 2, 1, 3, 4).unbind(0)   # (B, heads, N, hd)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(M, C)
    torch.testing.assert_close(out.double(), ref, rtol=2 ** -6, atol=2e-2)
    # and the same numbers as the row-major path of the same kernel family
    out_rm = torch.empty_like(out)
    ops.attention(rowmajor, out_rm, B, N, heads, hd)
    torch.testing.assert_close(out.float(), out_rm.float(), rtol=2 ** -6, atol=2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_patchify(ops, dtype):
    B, H, W, p = 2, 64, 48, 16
    x = torch.rand((B, 3, H, W), generator=torch.Generator().manual_seed(0)).cuda()
    out = torch.empty((B * (H // p) * (W // p), 3 * p * p), dtype=dtype, device="cuda")
    ops.patchify(x, out, p)
    ref = F.unfold(x, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, 3 * p * p)   # k = c*p*p + py*p + px
    assert torch.equal(out, ref.to(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("h,w,kh,kw", [(16, 12, 4, 3), (4, 4, 2, 2), (2, 2, 2, 2), (24, 18, 4, 3), (6, 6, 3, 3), (5, 5, 2, 2)])
def test_maxpool_relu(ops, dtype, h, w, kh, kw):
    B, C = 2, 64
    x = _rand((B * h * w, C), dtype, 1)
    out = torch.empty((B * (h // kh) * (w // kw), C), dtype=dtype, device="cuda")
    ops.maxpool_relu(x, out, B, h, w, C, kh, kw)
    xi = x.float().reshape(B, h, w, C).permute(0, 3, 1, 2)
    ref = F.relu(F.max_pool2d(xi, (kh, kw))).permute(0, 2, 3, 1).reshape(-1, C)
    assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_aux_tail(ops, dtype):
    B, C, K = 5, 384, 17
    x = _rand((B, 4 * C), dtype, 1)
    w = _rand((4, K, C), dtype, 2, C ** -0.5)
    b = _rand((4, K), torch.float32, 3)
    out = torch.empty((4, B, K), dtype=torch.float32, device="cuda")
    ops.aux_tail(x, w, b, out, B, C, K)
    for i in range(4):
        pre = x[:, i * C:(i + 1) * C].double() @ w[i].double().t() + b[i].double()
        ref = torch.sigmoid(pre) if i < 3 else F.relu(pre)
        torch.testing.assert_close(out[i].double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_transposes(ops, dtype):
    B, N, C = 3, 192, 100
    t = _rand((B * N, C), dtype, 1)
    out = torch.empty((B, C, N), dtype=torch.float32, device="cuda")
    ops.tokens_to_nchw(t, out, B, N, C)
    assert torch.equal(out, t.float().reshape(B, N, C).permute(0, 2, 1))
    back = torch.empty((B * N, C), dtype=dtype, device="cuda")
    ops.nchw_to_tokens(out, back, B, C, N)
    assert torch.equal(back, t)


def test_bad_arguments_raise(ops):
    from probpose_pytorch_amd import _lib
    A = torch.zeros((128, 40), device="cuda")
    W = torch.zeros((128, 40), device="cuda")
    with pytest.raises(_lib.HipExtensionError, match="multiple"):
        ops.linear(A, W)                       # K not a multiple of the tile depth
    with pytest.raises(_lib.HipExtensionError, match="head_dim"):
        ops.attention(torch.zeros((4, 3 * 48), device="cuda"), torch.zeros((4, 48), device="cuda"), 1, 4, 1, 48)
    with pytest.raises(TypeError):
        ops.linear(A.half(), W.half())


# ---------------------------------------------------------------------------
# fp8 (OCP e4m3) GEMM / LayerNorm: BASELINE config 5.  Reference = float64 product of the SAME e4m3
# operands (torch's casts), so only the fp32 accumulation order and the output rounding differ.
# ---------------------------------------------------------------------------
def _fp8_case(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((M, K), generator=g)
    w = torch.randn((N, K), generator=g) * K ** -0.5
    sa = float(a.abs().max()) / 448.0
    a8 = (a / sa).to(torch.float8_e4m3fn)
    from probpose_pytorch_amd import ops as _ops
    w8, sw = _ops.quantize_rows_fp8(w)
    bias = torch.randn((N,), generator=g)
    cs = (sa * sw).contiguous()
    ref = (a8.double() @ w8.double().T) * cs.double()[None, :] + bias.double()[None, :]
    return a8.cuda(), w8.cuda(), cs.cuda(), bias.cuda(), ref


@pytest.mark.parametrize("tile", [0, 2, 3, 10])
@pytest.mark.parametrize("M,N,K", [(192, 192, 128), (500, 384, 768), (1000, 96, 256), (77, 776, 384)])
def test_gemm_fp8_bf16_out(ops, M, N, K, tile):
    a8, w8, cs, bias, ref = _fp8_case(M, N, K, 11)
    out = ops.linear(a8, w8, bias, colscale=cs, tile=tile)
    assert out.dtype == torch.bfloat16
    torch.testing.assert_close(out.double().cpu(), ref, rtol=2 ** -8, atol=1e-3)


def test_gemm_fp8_gelu_fp8_out_and_residual(ops):
    M, N, K = 384, 768, 256
    a8, w8, cs, bias, ref = _fp8_case(M, N, K, 12)
    # fc1-style: GELU, output quantised to e4m3 with a static scale
    act = F.gelu(ref)
    so = float(act.abs().max()) / 448.0
    out8 = torch.empty((M, N), dtype=torch.float8_e4m3fn, device="cuda")
    ops.linear(a8, w8, bias, out=out8, epilogue=ops.EPI_GELU, colscale=cs, out_scale=so)
    got = out8.float().cpu().double() * so
    want8 = (act / so).float().to(torch.float8_e4m3fn).double() * so
    # equal up to one e4m3 step (2^-3 relative) where fp32-vs-fp64 accumulation lands across a rounding boundary
    err = (got - want8).abs()
    assert float((err > 0).double().mean()) < 0.02, "more than 2 % of the outputs differ from the f64 rounding"
    assert bool((err <= act.abs() * 2 ** -3 + so * 2 ** -9).all())
    # fc2-style: fp32 residual stream updated in place
    res = torch.randn((M, N), generator=torch.Generator().manual_seed(5)).cuda()
    want = res.double().cpu() + ref
    ops.linear(a8, w8, bias, out=res, residual=res, colscale=cs)
    # fp32 accumulation of e4m3 products: partial sums reach ~5e7 (ulp 4) before the 5e-6 dequantisation scale
    torch.testing.assert_close(res.double().cpu(), want, rtol=1e-5, atol=1e-3)


def test_gemm_fp8_argument_checks(ops):
    a8, w8, cs, bias, _ = _fp8_case(192, 192, 128, 1)
    with pytest.raises(TypeError):
        ops.linear(a8, w8, bias)                      # no column scales
    with pytest.raises(ops._lib.HipExtensionError):
        ops.linear(a8[:, :64], w8[:, :64], bias, colscale=cs)     # K must be a multiple of 128
    with pytest.raises(ops._lib.HipExtensionError):
        ops.linear(a8, w8, bias, colscale=cs, tile=5)


def test_layernorm_fp8(ops):
    rows, C = 300, 1024
    g = torch.Generator().manual_seed(3)
    x = (torch.randn((rows, C), generator=g) * 3 + 1).cuda()
    gam, bet = torch.randn((C,), generator=g).cuda(), torch.randn((C,), generator=g).cuda()
    ref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-6).cpu()
    scale = float(ref.abs().max()) / 448.0
    out = torch.empty((rows, C), dtype=torch.float8_e4m3fn, device="cuda")
    ops.layernorm(x, gam, bet, 1e-6, out, out_scale=scale)
    got = out.float().cpu().double() * scale
    want = (ref / scale).float().to(torch.float8_e4m3fn).double() * scale
    err = (got - want).abs()
    assert float((err > 0).double().mean()) < 0.01
    assert bool((err <= ref.abs() * 2 ** -3 + scale * 2 ** -9).all())
    # saturation instead of NaN: values beyond the scale clamp to +-448
    ops.layernorm(x, gam, bet, 1e-6, out, out_scale=scale / 4)
    assert torch.isfinite(out.float()).all() and float(out.float().abs().max()) == 448.0


@pytest.mark.parametrize("B,N,heads,hd", [(2, 192, 4, 64), (1, 432, 2, 80), (1, 192, 3, 32), (1, 70, 2, 64)])
def test_attention_fp8_output(ops, B, N, heads, hd):
    """fp8 mode: same attention, output stored as e4m3(o / scale) == the bf16-path result quantised the same way
    (up to one e4m3 step where the bf16 rounding of the reference path moved a value across a boundary)."""
    C = heads * hd
    qkv = _rand((B * N, 3 * C), torch.bfloat16, 1)
    ref = torch.empty((B * N, C), dtype=torch.bfloat16, device="cuda")
    ops.attention(qkv, ref, B, N, heads, hd)
    scale = float(ref.float().abs().max()) / 448.0
    out8 = torch.empty((B * N, C), dtype=torch.float8_e4m3fn, device="cuda")
    ops.attention(qkv, out8, B, N, heads, hd, out_scale=scale)
    got = out8.float() * scale
    err = (got - ref.float()).abs()
    assert bool((err <= ref.float().abs() * 2 ** -3 + scale * 2 ** -8).all())
    assert float(err.mean()) < float(ref.float().abs().mean()) * 0.05


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("split", [3, 9])
def test_gemm_splitk_aux_conv_and_pool_sum(ops, dtype, split):
    """Aux-branch stages >= 1 as split-K launches (taps split over workgroups, f32 partials) + the pooling kernel that
    sums the partials, adds the bias, pools and applies ReLU: equals conv3x3 + bias -> MaxPool -> ReLU of torch."""
    from probpose_pytorch_amd import pack
    B, h, w, C = 3, 4, 4, 64
    x = _rand((B * h * w, 4 * C), dtype, 1)
    wt = _rand((4, C, C, 3, 3), torch.float32, 2, (9 * C) ** -0.5)
    bias = _rand((4, C), torch.float32, 3)
    Wp = torch.stack([pack.conv_taps_major(wt[i].cpu()) for i in range(4)]).to(dtype).cuda()
    ro = pack.conv_gather_table(B, h, w, 3, 3, 1, 1, 4 * C).cuda()
    M = B * h * w
    taps = 9 // split
    for tile in (0, 2, 3, 4, 10):
        parts = torch.full((split, M, 4 * C), float("nan"), dtype=torch.float32, device="cuda")
        ops.gemm(x, Wp, parts, M=M, N=C, Kd=taps * C, lda=4 * C, ldw=9 * C, ldc=4 * C, rowoff=ro, seg_len=C, batch=4,
                 strideA=C, strideW=C * 9 * C, strideC=C, epilogue=ops.EPI_OUT_F32, splitk=split, strideW_k=taps * C,
                 strideRowoff_k=taps * M, strideC_k=M * 4 * C, tile=tile)
        conv = parts.sum(0) + bias.reshape(-1)
        out = torch.empty_like(conv).to(dtype)
        out.copy_(conv)
        _check_branches(x, Wp, bias, out, B, h, w, C, dtype)
        pooled = torch.empty((B * 2 * 2, 4 * C), dtype=dtype, device="cuda")
        ops.maxpool_relu_sum(parts, bias.reshape(-1).contiguous(), pooled, B, h, w, 4 * C, 2, 2)
        ref = F.relu(F.max_pool2d(conv.reshape(B, h, w, 4 * C).permute(0, 3, 1, 2), 2)).permute(0, 2, 3, 1)
        # the kernel adds the partials in split order, torch.sum pairwise: equal up to fp32 rounding of the sum
        torch.testing.assert_close(pooled.float().reshape(B, 2, 2, 4 * C), ref.to(dtype).float(),
                                   rtol=2e-6 if dtype == torch.float32 else 2 ** -7, atol=1e-6 if dtype == torch.float32 else 1e-2)
    with pytest.raises(Exception):      # split-K launches carry no bias / activation
        ops.gemm(x, Wp, parts, M=M, N=C, Kd=taps * C, lda=4 * C, ldw=9 * C, ldc=4 * C, rowoff=ro, seg_len=C, batch=4,
                 strideA=C, strideW=C * 9 * C, strideC=C, bias=bias, epilogue=ops.EPI_OUT_F32, splitk=split,
                 strideW_k=taps * C, strideRowoff_k=taps * M, strideC_k=M * 4 * C)
