"""Kernel-level parity of the MFMA implicit-GEMM conv / wgrad kernels (through include/dvg_dev.h)
against torch's CPU convolutions, for every weight mode, with and without the fused upsample / quad-sum."""
import pytest
import torch
import torch.nn.functional as F

from image_generation_amd import dev

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.fixture(params=["reg", "dma"])
def staging(request):
    """Both staging forms of the float32 GEMM kernels (options igemm_dma, wgrad_dma; the product runs the LDS-DMA forms):
    register-staged with ds_write, and LDS-DMA (forward / data-gradient with K-major packed weights; 3x3 weight gradient)."""
    from image_generation_amd import _lib
    v = 1 if request.param == "dma" else 0
    with _lib.option_scope(igemm_dma=v, wgrad_dma=v):
        yield request.param


@pytest.mark.parametrize("N,Cin,Cout,side", [(3, 32, 64, 16), (5, 64, 128, 8), (7, 128, 96, 4), (2, 64, 32, 8), (33, 32, 32, 4),
                                             (256, 128, 128, 4), (64, 128, 128, 4),  # these two take the split-K path
                                             (1024, 128, 128, 8), (600, 64, 64, 16),  # 128x128 and 128x64 tiles
                                             # position-major tiles (whole row blocks of images, no split-K): 8x8 above, 4x4, 2x2
                                             (1024, 32, 64, 4), (8192, 32, 64, 2)])
def test_conv2d_fwd_dgrad_wgrad(N, Cin, Cout, side, staging):
    torch.manual_seed(N)
    x = torch.randn(N, Cin, side, side); w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    x.requires_grad_(True); w.requires_grad_(True)
    y = F.conv2d(x, w, b, padding=1)
    gy = torch.randn_like(y)
    y.backward(gy)
    L, M = side.bit_length() - 1, N * side * side
    xm = dev.nchw_to_morton(x.detach()).cuda()
    out, st = dev.conv_igemm(xm, w.detach().cuda(), 0, M, Cin, Cout, L, bias=b.cuda(), stats=True)
    assert _rel(dev.morton_to_nchw(out.cpu(), N, Cout, side), y.detach()) < 2e-6
    s = st.sum(0).cpu()
    assert _rel(s[:, 0], y.detach().sum((0, 2, 3))) < 1e-4 and _rel(s[:, 1], (y.detach() ** 2).sum((0, 2, 3))) < 1e-5
    gym = dev.nchw_to_morton(gy).cuda()
    dx = dev.conv_igemm(gym, w.detach().cuda(), 1, M, Cout, Cin, L)
    assert _rel(dev.morton_to_nchw(dx.cpu(), N, Cin, side), x.grad) < 2e-6
    gw = dev.conv_wgrad(xm, gym, 0, w.shape, M, Cin, Cout, L)
    assert _rel(gw.cpu(), w.grad) < 3e-6


@pytest.mark.parametrize("N,Cin,Cout,side", [(64, 32, 64, 16), (256, 64, 128, 8), (512, 128, 512, 4), (512, 512, 128, 4),
                                             (128, 128, 64, 8), (64, 64, 32, 16), (2, 32, 64, 16), (32, 32, 32, 4)])
def test_winograd_form_of_the_3x3_layers(N, Cin, Cout, side):
    """The Winograd F(2x2,3x3) form (csrc/conv_wino.hip; the encoder's layers 1-3 from 256 workgroups up) against float64
    convolutions: outputs and data gradients within 3x the direct float32 kernel's own bar (its transforms add a few
    roundings per product sum), BatchNorm partial sums as the direct kernel's; and against the direct kernel itself."""
    from image_generation_amd import _lib
    torch.manual_seed(N + Cin)
    x = torch.randn(N, Cin, side, side); w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    gy = torch.randn(N, Cout, side, side)
    y64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    dx64 = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    L, M = side.bit_length() - 1, N * side * side
    with _lib.option_scope(enc_wino=1):
        assert dev.conv_wino_ok(M, Cin, Cout, L) and dev.conv_wino_ok(M, Cout, Cin, L)
        xm, gym = dev.nchw_to_morton(x).cuda(), dev.nchw_to_morton(gy).cuda()
        out, st = dev.conv_wino(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda(), stats=True)
        dx = dev.conv_wino(gym, w.cuda(), 1, M, Cout, Cin, L)
    out_d = dev.conv_igemm(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda())
    e_w = _rel(dev.morton_to_nchw(out.cpu(), N, Cout, side).double(), y64)
    e_d = _rel(dev.morton_to_nchw(out_d.cpu(), N, Cout, side).double(), y64)
    assert e_w < 2e-6 and e_w < 4 * e_d + 1e-8, (e_w, e_d)
    dx_d = dev.conv_igemm(gym, w.cuda(), 1, M, Cout, Cin, L)
    g_w = _rel(dev.morton_to_nchw(dx.cpu(), N, Cin, side).double(), dx64)
    g_d = _rel(dev.morton_to_nchw(dx_d.cpu(), N, Cin, side).double(), dx64)
    assert g_w < 2e-6 and g_w < 4 * g_d + 1e-8, (g_w, g_d)
    s = st.sum(0).cpu().double()
    assert _rel(s[:, 0], y64.sum((0, 2, 3))) < 1e-4 and _rel(s[:, 1], (y64 ** 2).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("N,Cin,Cout,side", [(64, 32, 64, 16), (256, 64, 128, 8), (512, 128, 512, 4), (512, 512, 128, 4),
                                             (128, 128, 64, 8), (64, 64, 32, 16), (4, 32, 64, 16), (64, 32, 32, 4), (16, 32, 32, 8), (48, 96, 160, 8)])
def test_winograd_f4x4_form_of_the_3x3_layers(N, Cin, Cout, side):
    """The Winograd F(4x4,3x3) form (csrc/conv_wino4.hip: 36 position GEMMs per 4x4 output tile on the 16x16x4 MFMA,
    lane-local output transform) against float64 convolutions -- forward with bias and BatchNorm partial sums, and the
    data gradient -- for every image size it serves (4x4: one tile per image, the halo known at compile time; 8x8; 16x16),
    with the F(2x2,3x3) kernel's and the direct kernel's own distances from float64 beside it.  Its transform constants are
    1/24 .. 8 where F(2x2) has +-1, 1/2: the bar is float32 summation noise of the LARGER intermediate values (measured
    on the CPU before the kernel existed, profiles/r05_wino_f43_accuracy_cpu.txt: rms 2e-6 of the output's rms)."""
    from image_generation_amd import _lib
    torch.manual_seed(N + Cin)
    x = torch.randn(N, Cin, side, side); w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    gy = torch.randn(N, Cout, side, side)
    y64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    dx64 = F.conv_transpose2d(gy.double(), w.double(), padding=1)
    L, M = side.bit_length() - 1, N * side * side
    assert dev.conv_wino4_shape(M, Cin, Cout, L) and dev.conv_wino4_shape(M, Cout, Cin, L)
    xm, gym = dev.nchw_to_morton(x).cuda(), dev.nchw_to_morton(gy).cuda()
    out, st = dev.conv_wino4(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda(), stats=True)
    dx = dev.conv_wino4(gym, w.cuda(), 1, M, Cout, Cin, L)
    rms = lambda a, ref: float((a.double() - ref).norm() / ref.norm())  # noqa: E731
    o4, g4 = dev.morton_to_nchw(out.cpu(), N, Cout, side), dev.morton_to_nchw(dx.cpu(), N, Cin, side)
    out_d = dev.conv_igemm(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda())
    dx_d = dev.conv_igemm(gym, w.cuda(), 1, M, Cout, Cin, L)
    od, gd = dev.morton_to_nchw(out_d.cpu(), N, Cout, side), dev.morton_to_nchw(dx_d.cpu(), N, Cin, side)
    row = {"fwd max": (_rel(o4, y64), _rel(od, y64)), "fwd rms": (rms(o4, y64), rms(od, y64)),
           "dgrad max": (_rel(g4, dx64), _rel(gd, dx64)), "dgrad rms": (rms(g4, dx64), rms(gd, dx64))}
    with _lib.option_scope(enc_wino=1):
        if dev.conv_wino_ok(M, Cin, Cout, L) and dev.conv_wino_ok(M, Cout, Cin, L):
            o2 = dev.morton_to_nchw(dev.conv_wino(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda()).cpu(), N, Cout, side)
            g2 = dev.morton_to_nchw(dev.conv_wino(gym, w.cuda(), 1, M, Cout, Cin, L).cpu(), N, Cin, side)
            row["F(2x2) fwd max/rms, dgrad max/rms"] = (_rel(o2, y64), rms(o2, y64), _rel(g2, dx64), rms(g2, dx64))
    print(f"F(4x4) vs float64 [N={N} {Cin}->{Cout} @{side}] (F(4x4), direct):", {k: tuple(f"{v:.2e}" for v in vs) for k, vs in row.items()})
    assert row["fwd max"][0] < 3e-5 and row["fwd rms"][0] < 8e-6, row
    assert row["dgrad max"][0] < 3e-5 and row["dgrad rms"][0] < 8e-6, row
    s = st.sum(0).cpu().double()
    assert _rel(s[:, 0], y64.sum((0, 2, 3))) < 1e-4 and _rel(s[:, 1], (y64 ** 2).sum((0, 2, 3))) < 1e-5
    # twice the same launch: same bits (dynamic tile deal, fixed arithmetic per tile)
    out2 = dev.conv_wino4(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda())
    assert torch.equal(out, out2)


@pytest.mark.parametrize("N,Cin,Cout,side,ups", [(64, 32, 64, 16, 0), (256, 64, 128, 8, 0), (512, 128, 512, 4, 0),  # the encoder's layers 1-3
                                                 (37, 64, 64, 8, 0), (5, 32, 64, 4, 0),                             # ragged image counts
                                                 (256, 128, 64, 8, 1), (64, 64, 32, 16, 1), (24, 64, 64, 4, 1)])    # behind the x2 upsample
def test_winograd_weight_gradient_kernel(N, Cin, Cout, side, ups):
    """conv_wino_wgrad8_kernel (csrc/conv_wino_wgrad.hip) alone, all six instantiations
    (<2,2>, <1,2>, <2,1> tiles, plain and behind the Upsample(x2): 9 of 16 positions) against a float64
    weight gradient of the same layer: within 2x the direct kernel's own distance (+ 1e-7: a handful of extra roundings
    in the transforms), i.e. float32 summation noise -- the whole-network tests hold these kernels to 1e-4 .. 5e-3 only."""
    from image_generation_amd import _lib
    torch.manual_seed(N + Cin + side)
    L, M = side.bit_length() - 1, N * side * side
    gy = torch.randn(N, Cout, side, side)
    if ups:  # ConvTranspose2d(Cin, Cout, 3, padding 1) of the nearest-upsampled map (side = OUTPUT resolution)
        xs = torch.randn(N, Cin, side // 2, side // 2)
        shape = (Cin, Cout, 3, 3)
        w = torch.zeros(shape, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose2d(F.interpolate(xs.double(), scale_factor=2, mode="nearest"), w, padding=1)
        ref, = torch.autograd.grad(y, w, gy.double())
        xm, mode = dev.nchw_to_morton(xs).cuda(), 2
    else:
        x = torch.randn(N, Cin, side, side)
        shape = (Cout, Cin, 3, 3)
        ref = torch.nn.grad.conv2d_weight(x.double(), shape, gy.double(), padding=1)
        xm, mode = dev.nchw_to_morton(x).cuda(), 0
    gym = dev.nchw_to_morton(gy).cuda()
    gw = dev.conv_wino_wgrad(xm, gym, mode, shape, M, Cin, Cout, L, ups=ups)
    assert gw is not None
    gd = dev.conv_wgrad(xm, gym, mode, shape, M, Cin, Cout, L, ups=ups)
    rel = lambda a: float((a.cpu().double() - ref).norm() / ref.norm())
    e_w, e_d = rel(gw), rel(gd)
    assert e_w < 2 * e_d + 1e-7 and e_w < 2e-6, (e_w, e_d)
    # twice the same launch: the slab sums are fixed-order, so the bits repeat
    gw2 = dev.conv_wino_wgrad(xm, gym, mode, shape, M, Cin, Cout, L, ups=ups)
    assert torch.equal(gw, gw2)


@pytest.mark.parametrize("N,Cin,Cout,side", [(256, 128, 64, 4), (64, 64, 32, 8), (64, 32, 32, 4), (16, 96, 64, 8), (128, 512, 128, 4)])
def test_winograd_f4x4_behind_the_upsample(N, Cin, Cout, side):
    """The decoder's Upsample(x2) + ConvTranspose2d 3x3 forward in the F(4x4,3x3) form with 25 of the 36 positions
    (csrc/conv_wino4.hip, UM = 1: the input transform of the repeated source pixels has a vanishing row) against float64, with
    the F(2x2)-behind-the-upsample kernel's distance beside it; side = OUTPUT resolution, the input lives at side / 2."""
    from image_generation_amd import _lib
    torch.manual_seed(N + Cin + side)
    xs = torch.randn(N, Cin, side // 2, side // 2)
    w = torch.randn(Cin, Cout, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    y64 = F.conv_transpose2d(F.interpolate(xs.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    L, M = side.bit_length() - 1, N * side * side
    xm = dev.nchw_to_morton(xs).cuda()
    out, st = dev.conv_wino4(xm, w.cuda(), 2, M, Cin, Cout, L, bias=b.cuda(), stats=True, um=1)
    o4 = dev.morton_to_nchw(out.cpu(), N, Cout, side)
    rms = lambda a, ref: float((a.double() - ref).norm() / ref.norm())  # noqa: E731
    out_d = dev.conv_igemm(xm, w.cuda(), 2, M, Cin, Cout, L, ups=1, bias=b.cuda())
    od = dev.morton_to_nchw(out_d.cpu(), N, Cout, side)
    print(f"F(4x4) behind the upsample vs float64 [N={N} {Cin}->{Cout} @{side}]: max {_rel(o4, y64):.2e} rms {rms(o4, y64):.2e}; "
          f"direct max {_rel(od, y64):.2e} rms {rms(od, y64):.2e}")
    assert _rel(o4, y64) < 3e-5 and rms(o4, y64) < 8e-6
    s = st.sum(0).cpu().double()
    assert _rel(s[:, 0], y64.sum((0, 2, 3))) < 1e-4 and _rel(s[:, 1], (y64 ** 2).sum((0, 2, 3))) < 1e-5
    out2 = dev.conv_wino4(xm, w.cuda(), 2, M, Cin, Cout, L, bias=b.cuda(), um=1)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("N,Cin,Cout,side", [(256, 128, 64, 4), (64, 64, 32, 8), (64, 32, 32, 4), (16, 96, 64, 8), (128, 512, 128, 4)])
def test_winograd_f4x4_data_gradient_behind_the_upsample(N, Cin, Cout, side):
    """The data gradient of the same layer onto the source map (csrc/conv_wino4.hip, UM = 2: the fine-grid gradient's patches,
    rows / columns {0,1,3,4,5} of the input transform, and the 2x2 sum of the upsample's adjoint folded into the output
    transform) against float64, with the direct kernel's distance beside it."""
    torch.manual_seed(N + Cin + side + 1)
    w = torch.randn(Cin, Cout, 3, 3) / (3 * Cout**0.5)
    gy = torch.randn(N, Cout, side, side)
    # d/d(source) = the 2x2 sums of the transposed convolution's adjoint (a plain convolution with the same weight)
    dfine = F.conv2d(gy.double(), w.double(), padding=1)
    ref = dfine.reshape(N, Cin, side // 2, 2, side // 2, 2).sum((3, 5))
    L, M = side.bit_length() - 1, N * side * side
    gym = dev.nchw_to_morton(gy).cuda()
    dx = dev.conv_wino4(gym, w.cuda(), 3, M, Cout, Cin, L, um=2)
    d4 = dev.morton_to_nchw(dx.cpu(), N, Cin, side // 2)
    dd = dev.morton_to_nchw(dev.conv_igemm(gym, w.cuda(), 3, M, Cout, Cin, L, poolsum=1).cpu(), N, Cin, side // 2)
    rms = lambda a: float((a.double() - ref).norm() / ref.norm())  # noqa: E731
    print(f"F(4x4) data gradient behind the upsample vs float64 [N={N} {Cout}->{Cin} @{side}]: max {_rel(d4, ref):.2e} rms {rms(d4):.2e}; "
          f"direct max {_rel(dd, ref):.2e} rms {rms(dd):.2e}")
    assert _rel(d4, ref) < 3e-5 and rms(d4) < 8e-6
    assert torch.equal(dx, dev.conv_wino4(gym, w.cuda(), 3, M, Cout, Cin, L, um=2))


@pytest.mark.parametrize("N,Cin,Cout,side,cus", [(64, 32, 64, 16, 0), (256, 64, 128, 8, 0), (512, 128, 512, 4, 0),   # the encoder's layers 1-3
                                                 (36, 64, 64, 8, 0), (4, 32, 64, 4, 0), (8, 64, 32, 16, 0),          # few images (no split), the 64 x 32 tile on 16x16
                                                 (512, 128, 512, 4, 256), (64, 32, 64, 16, 256)])                   # the whole chip's split
def test_winograd_f4x4_weight_gradient_kernel(N, Cin, Cout, side, cus):
    """conv_wino4_wgrad_kernel (csrc/conv_wino4_wgrad.hip: 36 position GEMMs over the 4x4 output tiles on the 16x16x4 MFMA,
    both operands transformed in the workgroup from coalesced global loads) against a float64 weight gradient of the same
    layer, with the F(2x2,3x3) kernel's and the direct kernel's distances beside it; both channel tiles (64 x 32, 32 x 64),
    every image size, split and unsplit launches; and twice the same launch for the bits."""
    torch.manual_seed(N + Cin + side)
    L, M = side.bit_length() - 1, N * side * side
    x = torch.randn(N, Cin, side, side)
    gy = torch.randn(N, Cout, side, side)
    shape = (Cout, Cin, 3, 3)
    ref = torch.nn.grad.conv2d_weight(x.double(), shape, gy.double(), padding=1)
    xm, gym = dev.nchw_to_morton(x).cuda(), dev.nchw_to_morton(gy).cuda()
    gw = dev.conv_wino4_wgrad(xm, gym, 0, shape, M, Cin, Cout, L, cus=cus)
    assert gw is not None
    gd = dev.conv_wgrad(xm, gym, 0, shape, M, Cin, Cout, L)
    rel = lambda a: float((a.cpu().double() - ref).norm() / ref.norm())  # noqa: E731
    e_4, e_d = rel(gw), rel(gd)
    g2 = dev.conv_wino_wgrad(xm, gym, 0, shape, M, Cin, Cout, L)
    print(f"F(4x4) weight gradient vs float64 [N={N} {Cin}->{Cout} @{side} cus={cus}]: F(4x4) {e_4:.2e}  direct {e_d:.2e}  F(2x2) " +
          (f"{rel(g2):.2e}" if g2 is not None else "n/a"))
    assert e_4 < 8e-6, (e_4, e_d)
    gw2 = dev.conv_wino4_wgrad(xm, gym, 0, shape, M, Cin, Cout, L, cus=cus)
    assert torch.equal(gw, gw2)


@pytest.mark.parametrize("N,Cin,Cout,side", [(1024, 32, 64, 4), (8192, 32, 64, 2), (1024, 128, 128, 8)])
def test_position_major_tiles_are_bit_identical_to_pixel_major_tiles(N, Cin, Cout, side):
    """Position-major tiles (conv.h: ConvArgs.posmajor) skip the taps that fall outside the image -- multiplications by
    zero padding -- and keep the order of the rest: the same bits as pixel-major tiles (option igemm_posmajor = 0)."""
    import os
    torch.manual_seed(N + side)
    L, M = side.bit_length() - 1, N * side * side
    x = torch.randn(M, Cin, device="cuda"); w = (torch.randn(Cout, Cin, 3, 3) / (3 * Cin**0.5)).cuda(); b = torch.randn(Cout).cuda()
    outs = {}
    from image_generation_amd import _lib
    for flag in ("1", "0"):  # "1": pixel-major tiles
        with _lib.option_scope(igemm_posmajor=0 if flag == "1" else 1):
            outs[flag] = dev.conv_igemm(x, w, 0, M, Cin, Cout, L, bias=b, stats=True)
    assert torch.equal(outs["0"][0], outs["1"][0])
    # (the BatchNorm partial rows group other pixels: their totals agree to float32 rounding)
    assert _rel(outs["0"][1].sum(0), outs["1"][1].sum(0)) < 1e-5


@pytest.mark.parametrize("N,Cin,Cout,side", [(4, 128, 64, 4), (3, 64, 32, 8), (2, 96, 128, 2)])
def test_convtranspose_with_fused_upsample(N, Cin, Cout, side, staging):
    """side = OUTPUT resolution; the input lives at side/2 and is nearest-upsampled inside the gather;
    the data-gradient sums each 2x2 quad in the epilogue."""
    torch.manual_seed(side)
    xs = torch.randn(N, Cin, side // 2, side // 2, requires_grad=True)
    w = (torch.randn(Cin, Cout, 3, 3) / (3 * Cin**0.5)).requires_grad_(True)
    b = torch.randn(Cout)
    y = F.conv_transpose2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, b, padding=1)
    gy = torch.randn_like(y)
    y.backward(gy)
    L, M = side.bit_length() - 1, N * side * side
    xm = dev.nchw_to_morton(xs.detach()).cuda()
    out = dev.conv_igemm(xm, w.detach().cuda(), 2, M, Cin, Cout, L, ups=1, bias=b.cuda())
    assert _rel(dev.morton_to_nchw(out.cpu(), N, Cout, side), y.detach()) < 2e-6
    gym = dev.nchw_to_morton(gy).cuda()
    dx = dev.conv_igemm(gym, w.detach().cuda(), 3, M, Cout, Cin, L, poolsum=1)
    assert _rel(dev.morton_to_nchw(dx.cpu(), N, Cin, side // 2), xs.grad) < 3e-6
    gw = dev.conv_wgrad(xm, gym, 2, w.shape, M, Cin, Cout, L, ups=1)
    assert _rel(gw.cpu(), w.grad) < 3e-6


@pytest.mark.parametrize("mode,M,Cin,Cout,side,ntaps", [(4, 1024, 512, 2048, 1, 1), (0, 64 * 16, 128, 1024, 4, 9), (2, 64 * 16, 512, 128, 4, 9)])
def test_slab_sums_of_few_slabs_of_a_large_weight_are_the_same_bits(mode, M, Cin, Cout, side, ntaps):
    """wgrad_reduce_tile_kernel (at most 8 slabs, at least 256 32 x 32 tiles: coalesced reads and writes) against the
    8-lanes-per-element kernel on the same slabs: Linear, Conv2d and ConvTranspose2d checkpoint layouts."""
    from image_generation_amd import _lib
    torch.manual_seed(M + Cin)
    L = side.bit_length() - 1
    x = torch.randn(M, Cin, device="cuda"); dy = torch.randn(M, Cout, device="cuda")
    shape = {4: (Cout, Cin), 0: (Cout, Cin, 3, 3), 2: (Cin, Cout, 3, 3)}[mode]
    got = {}
    for form in (0, 1):
        with _lib.option_scope(wgrad_reduce_tiled=form):
            got[form] = dev.conv_wgrad(x, dy, mode, shape, M, Cin, Cout, L, ntaps=ntaps)
    assert torch.equal(got[0], got[1])
    assert got[1].abs().max().item() > 0


@pytest.mark.parametrize("N,n", [(37, 64), (700, 128), (16384, 128)])  # (n = 128: the 128 x 128-tile 1-tap weight-gradient kernel)
def test_linear_as_one_tap_gemm(N, n, staging):
    torch.manual_seed(0)
    x = torch.randn(N, n, requires_grad=True); w = (torch.randn(4 * n, n) / n**0.5).requires_grad_(True)
    y = F.linear(x, w)  # (N, 4n), column c*4+p
    gy = torch.randn_like(y)
    y.backward(gy)
    out = dev.conv_igemm(x.detach().cuda(), w.detach().cuda(), 4, N, n, 4 * n, 0, ntaps=1)  # column p*n + c
    want = y.detach().reshape(N, n, 4).permute(0, 2, 1).reshape(N, 4 * n)
    assert _rel(out.cpu(), want) < 2e-6
    gyp = gy.reshape(N, n, 4).permute(0, 2, 1).reshape(N, 4 * n).contiguous().cuda()
    dx = dev.conv_igemm(gyp, w.detach().cuda(), 5, N, 4 * n, n, 0, ntaps=1)
    assert _rel(dx.cpu(), x.grad) < 2e-6
    gw = dev.conv_wgrad(x.detach().cuda(), gyp, 4, w.shape, N, n, 4 * n, 0, ntaps=1)
    assert _rel(gw.cpu(), w.grad) < 3e-6


# ----------------------------------------------------------------------------------------------------------------
# "bf16 GEMM inputs, f32 accumulate" (dvg_set_conv_precision(DVG_PRECISION_BF16_INPUTS)): the forward and data-gradient
# GEMMs must equal a float32 convolution of the bf16-ROUNDED operands (exact products, f32 sums) to summation-order
# rounding; against the unrounded float32 result they sit at bf16's 2^-9 relative operand error.

class _bf16_inputs:
    def __enter__(self):
        from image_generation_amd import _lib
        self.lib = _lib.lib()
        assert self.lib.dvg_set_conv_precision(1) == 0 and self.lib.dvg_get_conv_precision() == 1

    def __exit__(self, *exc):
        assert self.lib.dvg_set_conv_precision(0) == 0


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("N,Cin,Cout,side", [(3, 32, 64, 16), (5, 64, 128, 8), (7, 128, 96, 4), (2, 64, 32, 8),
                                             (256, 128, 128, 4), (64, 128, 128, 4), (40, 128, 128, 8)])
def test_bf16_inputs_conv2d_fwd_dgrad(N, Cin, Cout, side):
    torch.manual_seed(N)
    x = torch.randn(N, Cin, side, side); w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    gy = torch.randn(N, Cout, side, side)
    y_r = F.conv2d(_r(x), _r(w), b, padding=1)                      # bf16-rounded operands, f32 arithmetic
    dx_r = torch.nn.grad.conv2d_input(x.shape, _r(w), _r(gy), padding=1)
    y_f = F.conv2d(x, w, b, padding=1)
    L, M = side.bit_length() - 1, N * side * side
    with _bf16_inputs():
        out, st = dev.conv_igemm(dev.nchw_to_morton(x).cuda(), w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda(), stats=True)
        dx = dev.conv_igemm(dev.nchw_to_morton(gy).cuda(), w.cuda(), 1, M, Cout, Cin, L)
    got = dev.morton_to_nchw(out.cpu(), N, Cout, side)
    assert _rel(got, y_r) < 3e-6
    assert 1e-5 < _rel(got, y_f) < 2e-2                             # it really ran on rounded operands
    s = st.sum(0).cpu()
    assert _rel(s[:, 0], y_r.sum((0, 2, 3))) < 1e-4 and _rel(s[:, 1], (y_r ** 2).sum((0, 2, 3))) < 1e-5
    assert _rel(dev.morton_to_nchw(dx.cpu(), N, Cin, side), dx_r) < 3e-6


@pytest.mark.parametrize("N,Cin,Cout,side", [(4, 128, 64, 4), (3, 64, 32, 8), (2, 96, 128, 2)])
def test_bf16_inputs_convtranspose_with_fused_upsample(N, Cin, Cout, side):
    torch.manual_seed(side)
    xs = torch.randn(N, Cin, side // 2, side // 2); w = torch.randn(Cin, Cout, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    gy = torch.randn(N, Cout, side, side)
    xr = _r(xs).requires_grad_(True)
    y_r = F.conv_transpose2d(F.interpolate(xr, scale_factor=2, mode="nearest"), _r(w), b, padding=1)
    y_r.backward(_r(gy))
    L, M = side.bit_length() - 1, N * side * side
    with _bf16_inputs():
        out = dev.conv_igemm(dev.nchw_to_morton(xs).cuda(), w.cuda(), 2, M, Cin, Cout, L, ups=1, bias=b.cuda())
        dx = dev.conv_igemm(dev.nchw_to_morton(gy).cuda(), w.cuda(), 3, M, Cout, Cin, L, poolsum=1)
    assert _rel(dev.morton_to_nchw(out.cpu(), N, Cout, side), y_r.detach()) < 3e-6
    assert _rel(dev.morton_to_nchw(dx.cpu(), N, Cin, side // 2), xr.grad) < 5e-6


def test_bf16_inputs_linear_as_one_tap_gemm():
    torch.manual_seed(0)
    N, n = 37, 64
    x = torch.randn(N, n); w = torch.randn(4 * n, n) / n**0.5
    gy = torch.randn(N, 4 * n)
    y_r = F.linear(_r(x), _r(w))
    gyp = gy.reshape(N, n, 4).permute(0, 2, 1).reshape(N, 4 * n).contiguous()
    dx_r = _r(gy) @ _r(w)
    with _bf16_inputs():
        out = dev.conv_igemm(x.cuda(), w.cuda(), 4, N, n, 4 * n, 0, ntaps=1)
        dx = dev.conv_igemm(gyp.cuda(), w.cuda(), 5, N, 4 * n, n, 0, ntaps=1)
    assert _rel(out.cpu(), y_r.reshape(N, n, 4).permute(0, 2, 1).reshape(N, 4 * n)) < 3e-6
    assert _rel(dx.cpu(), dx_r) < 3e-6


# ----------------------------------------------------------------------------------------------------------------
# DVG_PRECISION_F32_SPLIT3: float32 operands as three bf16 pieces, six piece products on the bf16 MFMA, float32
# accumulation.  The bar is the float32 kernel's own: against float64 it must be as close as the f32-MFMA form is.

class _split3:
    """(the split form serves every tile configuration of the LDS-DMA kernel)"""
    def __enter__(self):
        from image_generation_amd import _lib
        self.lib = _lib.lib()
        assert self.lib.dvg_set_conv_precision(2) == 0 and self.lib.dvg_get_conv_precision() == 2

    def __exit__(self, *exc):
        assert self.lib.dvg_set_conv_precision(0) == 0


@pytest.mark.parametrize("N,Cin,Cout,side", [(3, 32, 64, 16), (5, 64, 128, 8), (7, 128, 96, 4), (2, 64, 32, 8),
                                             (256, 128, 128, 4), (64, 128, 128, 4), (40, 128, 128, 8), (512, 128, 256, 8)])
def test_split3_conv2d_fwd_dgrad_is_float32_class(N, Cin, Cout, side):
    torch.manual_seed(N)
    x = torch.randn(N, Cin, side, side); w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    gy = torch.randn(N, Cout, side, side)
    y64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    L, M = side.bit_length() - 1, N * side * side
    xm, gm = dev.nchw_to_morton(x).cuda(), dev.nchw_to_morton(gy).cuda()
    out_f, st_f = dev.conv_igemm(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda(), stats=True)
    dx_f = dev.conv_igemm(gm, w.cuda(), 1, M, Cout, Cin, L)
    with _split3():
        out, st = dev.conv_igemm(xm, w.cuda(), 0, M, Cin, Cout, L, bias=b.cuda(), stats=True)
        dx = dev.conv_igemm(gm, w.cuda(), 1, M, Cout, Cin, L)
    e_s = _rel(dev.morton_to_nchw(out.cpu(), N, Cout, side).double(), y64)
    e_f = _rel(dev.morton_to_nchw(out_f.cpu(), N, Cout, side).double(), y64)
    # float32-class: well inside the 2e-6 bar the float32 kernel itself is held to against torch's float32 convolution
    # (the bf16 MFMA's float32 accumulate is a little coarser than an fmaf chain: measured 2-5e-7 against 1e-7)
    assert e_s < max(1e-6, 1.5 * e_f), (e_s, e_f)
    d_s = _rel(dev.morton_to_nchw(dx.cpu(), N, Cin, side).double(), dx64)
    d_f = _rel(dev.morton_to_nchw(dx_f.cpu(), N, Cin, side).double(), dx64)
    assert d_s < max(1e-6, 1.5 * d_f), (d_s, d_f)   # (K = 2304: both forms sit at 2e-6 of the largest entry)
    s = st.sum(0).cpu()
    assert _rel(s[:, 0].double(), y64.sum((0, 2, 3))) < 1e-4 and _rel(s[:, 1].double(), (y64 ** 2).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("N,Cin,Cout,side", [(4, 128, 64, 4), (3, 64, 32, 8), (2, 96, 128, 2), (128, 128, 64, 8)])
def test_split3_convtranspose_with_fused_upsample_and_fold(N, Cin, Cout, side):
    torch.manual_seed(side)
    xs = torch.randn(N, Cin, side // 2, side // 2); w = torch.randn(Cin, Cout, 3, 3) / (3 * Cin**0.5); b = torch.randn(Cout)
    gy = torch.randn(N, Cout, side, side)
    x64 = xs.double().requires_grad_(True)
    y64 = F.conv_transpose2d(F.interpolate(x64, scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    y64.backward(gy.double())
    L, M = side.bit_length() - 1, N * side * side
    with _split3():
        out = dev.conv_igemm(dev.nchw_to_morton(xs).cuda(), w.cuda(), 2, M, Cin, Cout, L, ups=1, bias=b.cuda())
        dx = dev.conv_igemm(dev.nchw_to_morton(gy).cuda(), w.cuda(), 3, M, Cout, Cin, L, poolsum=1)
    assert _rel(dev.morton_to_nchw(out.cpu(), N, Cout, side).double(), y64.detach()) < 1e-6
    assert _rel(dev.morton_to_nchw(dx.cpu(), N, Cin, side // 2).double(), x64.grad) < 1e-6


def test_split3_linear_as_one_tap_gemm():
    torch.manual_seed(0)
    N, n = 37, 64
    x = torch.randn(N, n); w = torch.randn(4 * n, n) / n**0.5
    gy = torch.randn(N, 4 * n)
    y64 = F.linear(x.double(), w.double())
    gyp = gy.reshape(N, n, 4).permute(0, 2, 1).reshape(N, 4 * n).contiguous()
    with _split3():
        out = dev.conv_igemm(x.cuda(), w.cuda(), 4, N, n, 4 * n, 0, ntaps=1)
        dx = dev.conv_igemm(gyp.cuda(), w.cuda(), 5, N, 4 * n, n, 0, ntaps=1)
    assert _rel(out.cpu().double(), y64.reshape(N, n, 4).permute(0, 2, 1).reshape(N, 4 * n)) < 1e-6
    assert _rel(dx.cpu().double(), gy.double() @ w.double()) < 1e-6
