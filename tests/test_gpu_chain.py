"""Forward-chain kernels (csrc/chain.hip) through the C ABI: fused TTLinearM chain and the single products of the
TTConv2dM / TKConv2dC chains, fp32 (three-plane bf16 split) and bf16, against fp64 torch references.

Tolerances: the fp32 mode must be as accurate as an fp32 GEMM -- error <= 2e-6 of the largest |output| at K <= 1536
(measured ~3e-7; an fp32 `torch.mm` on the CPU gives ~6e-7 on the same data); the bf16 mode is bounded by the bf16
rounding of inputs, the intermediate and the output (2^-8 each): 2e-2 of the largest |output|."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from tadmm import ops
    return ops


def _ref_fused(x, win, wout, bias):
    y = (x.double() @ win.double().t()) @ wout.double().t()
    return y + (bias.double() if bias is not None else 0)


@pytest.mark.parametrize("T,kin,r,nout,tile", [
    (12608, 384, 256, 1152, 0),      # DeiT-S qkv at the bench's token count
    (12608, 384, 256, 1152, 64),
    (197, 384, 96, 384, 0),          # ragged token tile, middle rank not a multiple of 64
    (50, 72, 20, 40, 0),             # Kin, R, Nout all need padding (R padded to 64 by weight_planes)
    (1, 1536, 256, 384, 0),          # one token, long K (DeiT-S fc2)
])
def test_fused_fp32_matches_fp64(T, kin, r, nout, tile):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(T + kin)
    x = torch.randn(T, kin, generator=g).cuda()
    win = (torch.randn(r, kin, generator=g) / kin ** 0.5).cuda()
    wout = (torch.randn(nout, r, generator=g) / r ** 0.5).cuda()
    bias = torch.randn(nout, generator=g).cuda()
    wp_in = ops.weight_planes(win, 3, pad_rows=64)
    wp_out = ops.weight_planes(wout, 3, pad_cols=64)
    assert wp_out.shape[2] * 32 == wp_in.shape[1] * 16
    # the three planes reproduce the fp32 weight exactly
    assert torch.equal(ops.unpack_planes(wp_in).float().sum(0)[:r, :kin], win)
    y = ops.chain_fused(x, wp_in, wp_out, bias, nout, tile_tokens=tile)
    ref = _ref_fused(x, win, wout, bias)
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    assert y.shape == (T, nout) and err < 2e-6, err
    y2 = ops.chain_fused(x, wp_in, wp_out, None, nout, tile_tokens=tile)
    torch.testing.assert_close(y2 + bias, y, atol=1e-6 * ref.abs().max().item(), rtol=0)


@pytest.mark.parametrize("T,kin,r,nout,tile", [(12608, 384, 256, 1152, 0), (12608, 384, 256, 1152, 32), (77, 64, 32, 24, 0)])
def test_fused_bf16(T, kin, r, nout, tile):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(T, kin, generator=g).cuda().bfloat16()
    win = (torch.randn(r, kin, generator=g) / kin ** 0.5).cuda()
    wout = (torch.randn(nout, r, generator=g) / r ** 0.5).cuda()
    bias = torch.randn(nout, generator=g).cuda()
    wp_in, wp_out = ops.weight_planes(win, 1, pad_rows=64), ops.weight_planes(wout, 1, pad_cols=64)
    y = ops.chain_fused(x, wp_in, wp_out, bias, nout, tile_tokens=tile)
    assert y.dtype == torch.bfloat16
    ref = _ref_fused(x.float(), ops.unpack_planes(wp_in)[0, :r, :kin].float(), ops.unpack_planes(wp_out)[0, :nout, :r].float(), bias)
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-2, err


@pytest.mark.parametrize("B,c,hw,n,img_out", [
    (4, 64, (14, 14), 40, True),       # vectorised image loads (196 % 4 == 0)
    (3, 48, (7, 7), 36, True),         # 49 pixels: scalar loads, images straddle token tiles
    (2, 512, (7, 7), 300, False),      # more than 256 output features: two feature blocks
])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_single_product_on_images(B, c, hw, n, img_out, dtype):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(B * c + n)
    x = torch.randn(B, c, *hw, generator=g).cuda().to(dtype)
    w = (torch.randn(n, c, generator=g) / c ** 0.5).cuda()
    bias = torch.randn(n, generator=g).cuda()
    planes = 3 if dtype == torch.float32 else 1
    wp = ops.weight_planes(w, planes)
    y = ops.chain_single(x, wp, bias, n, image_out=img_out)
    wq = ops.unpack_planes(wp).float().sum(0)[:n, :c].double()
    ref = torch.einsum("bchw,nc->bnhw", x.double(), wq) + bias.double().view(1, -1, 1, 1)
    if not img_out:
        ref = ref.permute(0, 2, 3, 1).reshape(-1, n)
    assert y.shape == ref.shape
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (2e-6 if dtype == torch.float32 else 1e-2), err


def test_single_product_rows_to_rows_and_errors():
    ops = _ops()
    from tadmm._cabi import TadmmError
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(300, 96, generator=g).cuda()
    w = torch.randn(520, 96, generator=g).cuda()
    wp = ops.weight_planes(w, 3)
    y = ops.chain_single(x, wp, None, 520, entry="tadmm_tucker_1x1")
    ref = x.double() @ w.double().t()
    assert (y.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6
    # fused middle rank above the LDS-resident limit is refused, loudly
    big = ops.weight_planes(torch.randn(320, 96, generator=g).cuda(), 3, pad_rows=64)
    out = ops.weight_planes(torch.randn(64, 320, generator=g).cuda(), 3, pad_cols=64)
    with pytest.raises(TadmmError):
        ops.chain_fused(x, big, out, None, 64)
    with pytest.raises(TadmmError):
        ops.chain_fused(x.cpu(), wp, wp, None, 64)


# ------------------------------------------------------------------ the layers run on these kernels
class _HP:
    pass


def _deit_qkv():
    hp = _HP()
    hp.tt_shapes = {"qkv.weight": [36, 32, 16, 24]}
    hp.ranks = {"qkv.weight": [1, 25, 256, 18, 1]}
    return hp


def test_ttlinearm_uses_the_fused_entry_and_matches_the_per_core_chain(monkeypatch):
    from tadmm import ops, tt_layers
    torch.manual_seed(0)
    lin = tt_layers.TTLinearM(384, 1152, bias=True, hp_dict=_deit_qkv(), name="qkv.weight").cuda()
    with torch.no_grad():
        lin.bias.normal_()
    x = torch.randn(64, 197, 384, device="cuda")
    calls = []
    real = ops.chain_fused
    monkeypatch.setattr(ops, "chain_fused", lambda *a, **k: (calls.append(k.get("entry", "tadmm_ttlinear_fwd")), real(*a, **k))[1])
    with torch.no_grad():
        y = lin(x)
        ref = lin._forward_chain(x)                    # TTLinear.py:79-86 product by product on the fp32 GEMM
    assert calls == ["tadmm_ttlinear_fwd"]
    scale = ref.abs().max().item()
    assert (y - ref).abs().max().item() < 5e-6 * scale
    # bf16 inference: same kernel, one plane; bounded by bf16 rounding of x, H and y
    with torch.no_grad():
        yb = lin(x.bfloat16())
    assert yb.dtype == torch.bfloat16 and calls[-1] == "tadmm_ttlinear_fwd"
    assert (yb.float() - ref).abs().max().item() < 3e-2 * scale
    # the packed factors follow in-place updates of the cores (version counters)
    with torch.no_grad():
        lin.tt_cores[1].mul_(2.0)
        y2 = lin(x)
        ref2 = lin._forward_chain(x)
    assert (y2 - ref2).abs().max().item() < 5e-6 * ref2.abs().max().item()
    assert (y2 - y).abs().max().item() > 1e-3 * scale


def test_ttlinearm_fused_backward_matches_fp64_autograd():
    from tadmm import tt_layers
    torch.manual_seed(1)
    hp = _HP()
    hp.tt_shapes = {"w": [8, 6, 4, 12]}
    hp.ranks = {"w": [1, 6, 40, 5, 1]}
    lin = tt_layers.TTLinearM(48, 48, bias=True, hp_dict=hp, name="w").cuda()
    x = torch.randn(70, 48, device="cuda", requires_grad=True)
    y = lin(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    # fp64 reference: dense weight rebuilt from the cores (ttd.py:39-40), autograd through torch
    cores = [c.detach().double().requires_grad_(True) for c in lin.tt_cores]
    w = cores[0].reshape(-1, cores[0].shape[-1])
    for c in cores[1:]:
        w = w.reshape(-1, c.shape[0]) @ c.reshape(c.shape[0], -1)
    xd = x.detach().double().requires_grad_(True)
    bd = lin.bias.detach().double().requires_grad_(True)
    yd = xd @ w.reshape(48, 48).t() + bd
    yd.backward(gy.double())
    assert (y.double() - yd).abs().max().item() < 1e-5 * yd.abs().max().item()
    assert (x.grad.double() - xd.grad).abs().max().item() < 1e-5 * xd.grad.abs().max().item()
    assert (lin.bias.grad.double() - bd.grad).abs().max().item() < 1e-5 * bd.grad.abs().max().item()
    for c, cd in zip(lin.tt_cores, cores):
        assert (c.grad.double() - cd.grad).abs().max().item() < 2e-5 * cd.grad.abs().max().item()


# ------------------------------------------------------------------ fused factorised convolution (csrc/convchain.hip)
@pytest.mark.parametrize("B,C,H,W,r1,r2,O,k,stride,pad,dil", [
    (8, 64, 8, 8, 23, 25, 64, 3, 1, 1, 1),       # TKConv2dC ResNet-32 layer3 (CIFAR)
    (5, 96, 7, 7, 40, 36, 80, 3, 1, 1, 1),       # 49 pixels: scalar loads, ragged ranks
    (3, 32, 8, 8, 16, 20, 24, 3, 2, 1, 1),       # stride 2: 8x8 -> 4x4
    (2, 48, 6, 6, 24, 24, 40, 5, 1, 2, 1),       # 5x5 taps
    (2, 32, 7, 7, 12, 12, 16, 3, 1, 2, 2),       # dilation 2
    (4, 40, 5, 5, 9, 11, 12, 1, 1, 0, 1),        # 1x1 core
    (3, 64, 14, 14, 40, 36, 48, 3, 1, 1, 1),     # 14x14: four output rows per workgroup, halo of six input rows
    (2, 32, 28, 28, 24, 24, 40, 3, 1, 1, 1),     # 28x28: two output rows per workgroup
    (2, 16, 56, 56, 12, 16, 24, 3, 1, 1, 1),     # 56x56: one output row per workgroup, 3 x 56 pixels of halo
    (2, 48, 14, 14, 20, 20, 32, 3, 2, 1, 1),     # stride 2: 14x14 -> 7x7 in one tile, halo = the whole input plane (196 px: NT > 3 -> smaller tile)
    (2, 24, 20, 12, 16, 16, 24, 5, 1, 2, 1),     # non-square plane, 5x5 taps
    (4, 128, 7, 7, 220, 220, 96, 3, 1, 1, 1),    # TT ranks: in fp32 the three planes only fit with 32 pixels per workgroup
    (3, 64, 14, 14, 138, 138, 80, 3, 1, 1, 1),   # ResNet-18 layer3 ranks (fp32: 32-pixel tiles, two rows each)
])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_chain_matches_three_torch_convs(B, C, H, W, r1, r2, O, k, stride, pad, dil, dtype):
    import torch.nn.functional as F
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(B * C + k)
    x = torch.randn(B, C, H, W, generator=g).cuda().to(dtype)
    w1 = (torch.randn(r1, C, generator=g) / C ** 0.5).cuda()
    core = (torch.randn(r2, r1, k, k, generator=g) / (r1 * k * k) ** 0.5).cuda()
    w3 = (torch.randn(O, r2, generator=g) / r2 ** 0.5).cuda()
    bias = torch.randn(O, generator=g).cuda()
    n = 3 if dtype == torch.float32 else 1
    ks, st, pd, dl = (k, k), (stride, stride), (pad, pad), (dil, dil)
    assert ops.conv_chain_fits(x, r1, r2, ks, st, pd, dl)
    p1, p2, p3 = ops.weight_planes(w1, n, pad_rows=32), ops.conv_core_planes(core, n), ops.weight_planes(w3, n)
    y = ops.conv_chain(x, p1, p2, p3, bias, O, ks, st, pd, dl)
    # reference in fp64 on the weights the kernel sees (bf16 mode: rounded to bf16)
    q = (lambda t: t.double()) if dtype == torch.float32 else (lambda t: t.to(torch.bfloat16).double())
    xd = x.double()
    ref = F.conv2d(xd, q(w1).view(r1, C, 1, 1))
    ref = F.conv2d(ref, q(core), None, st, pd, dl)
    ref = F.conv2d(ref, q(w3).view(O, r2, 1, 1), bias.double())
    assert y.shape == ref.shape and y.dtype == dtype
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (3e-6 if dtype == torch.float32 else 3e-2), err


def test_small_image_conv_layers_take_the_fused_launch(monkeypatch):
    """TTConv2dM / TKConv2dC / TKConv2dM in inference: one `tadmm_ttconv_fused` launch (whole plane or row tiles), equal to
    the three-launch path (the path training and very wide planes take)."""
    from tadmm import ops, tk_layers, tt_layers
    torch.manual_seed(3)
    hp = _HP()
    hp.tt_shapes = {"c.weight": [8, 8, 9, 8, 8]}
    hp.ranks = {"c.weight": [1, 8, 40, 40, 8, 1]}
    conv = tt_layers.TTConv2dM(64, 64, 3, padding=1, bias=True, hp_dict=hp, name="c.weight").cuda()
    with torch.no_grad():
        conv.bias.normal_()
    hk = _HP()
    hk.ranks = {"k.weight": [25, 23]}
    tkc = tk_layers.TKConv2dC(64, 64, 3, padding=1, bias=True, hp_dict=hk, name="k.weight").cuda()
    tkm = tk_layers.TKConv2dM(64, 64, 3, padding=1, bias=True, hp_dict=hk, name="k.weight").cuda()
    x = torch.randn(16, 64, 8, 8, device="cuda")
    calls = []
    real = ops.conv_chain
    monkeypatch.setattr(ops, "conv_chain", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for layer in (conv, tkc, tkm):
        n0 = len(calls)
        with torch.no_grad():
            y = layer(x)
        assert len(calls) == n0 + 1
        y3 = layer(x)                                        # grad mode: three launches, differentiable
        assert len(calls) == n0 + 1 and y3.requires_grad
        assert (y - y3.detach()).abs().max().item() < 5e-6 * y3.abs().max().item()
        big = torch.randn(2, 64, 14, 14, device="cuda")      # 196 pixels = four row tiles per image
        with torch.no_grad():
            layer(big)                                       # fp32: fits, but does not pay -> three launches
        assert len(calls) == n0 + 1
        with torch.no_grad():
            yb = layer(big.bfloat16())                       # bf16: one launch
        assert len(calls) == n0 + 2
        yb3 = layer(big)
        assert (yb.float() - yb3.detach()).abs().max().item() < 3e-2 * yb3.abs().max().item()
        wide = torch.randn(1, 64, 4, 80, device="cuda")      # rows of 80 pixels: three launches
        with torch.no_grad():
            layer(wide)
        assert len(calls) == n0 + 2


def test_tklinearm_runs_on_the_fused_chain_and_is_differentiable(monkeypatch):
    from tadmm import ops, tk_layers
    torch.manual_seed(5)
    hk = _HP()
    hk.ranks = {"k.weight": [40, 24]}
    lin = tk_layers.TKLinearM(96, 128, bias=True, hp_dict=hk, name="k.weight").cuda()
    with torch.no_grad():
        lin.bias.normal_()
    x = torch.randn(333, 96, device="cuda")
    calls = []
    real = ops.chain_fused
    monkeypatch.setattr(ops, "chain_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        y = lin(x)
    assert len(calls) == 1
    W = (lin.last_factor.double() @ lin.core_tensor.double() @ lin.first_factor.double())
    ref = x.double() @ W.t() + lin.bias.double()
    assert (y.double() - ref).abs().max().item() < 5e-6 * ref.abs().max().item()
    xg = x.clone().requires_grad_(True)
    yg = lin(xg)
    gy = torch.randn_like(yg)
    yg.backward(gy)
    params = [p.detach().double().requires_grad_(True) for p in (lin.first_factor, lin.core_tensor, lin.last_factor, lin.bias)]
    xd = x.double().requires_grad_(True)
    yd = xd @ (params[2] @ params[1] @ params[0]).t() + params[3]
    yd.backward(gy.double())
    assert (xg.grad.double() - xd.grad).abs().max().item() < 1e-5 * xd.grad.abs().max().item()
    for p, pd in zip((lin.first_factor, lin.core_tensor, lin.last_factor, lin.bias), params):
        assert (p.grad.double() - pd.grad).abs().max().item() < 2e-5 * pd.grad.abs().max().item()
