"""GPU parity of the fused encoder / decoder (forward, backward, BN running stats) against the
CPU oracle and against the fixtures generated from the reference modules."""
import os

import numpy as np
import pytest
import torch

import gen
from image_generation_amd import _lib
from image_generation_amd.modules import Decoder, Encoder
from oracle import nets

pytestmark = pytest.mark.gpu


def _load(module, params):
    module.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in params.items()})
    return module.cuda()


def _oracle_params(params):
    out = {}
    for k, v in params.items():
        t = torch.from_numpy(np.array(v))
        if t.dtype == torch.float32 and "running" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def _close(got, want, rtol, name, atol_scale=1e-5):
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want).max()
    assert err <= rtol * scale + atol_scale * 1e-3, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(scope="module")
def fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "enc_dec_n64.npz")))


def test_encoder_matches_reference_fixture(fx):
    n, B = int(fx["n"]), int(fx["B"])
    enc = _load(Encoder(n), gen.make_params(n, "encoder", 101)).train()
    x = torch.from_numpy(gen.make_images(B, 202)).cuda()
    gl = torch.from_numpy(np.random.default_rng(303).standard_normal((B, n)).astype(np.float32)).cuda()
    logits = enc(x)
    _close(logits.detach().cpu(), fx["enc_train_logits"], 2e-5, "logits")
    (logits * gl).sum().backward()
    for name, prm in enc.named_parameters():
        g = prm.grad.cpu().numpy()
        want_sub = fx[f"enc_grad_sub/{name}"]
        l2 = fx[f"enc_grad_norm/{name}"][1]
        # conv biases in front of a BatchNorm have a zero true gradient (rounding noise only)
        assert np.abs(gen.subsample(g) - want_sub).max() <= 1e-4 * max(l2, 1.0) / np.sqrt(max(g.size, 1)) * 30 + 3e-4 * np.abs(want_sub).max() + 2e-5, name
    sd = enc.state_dict()
    for k in fx:
        if k.startswith("enc_after/"):
            _close(sd[k[len("enc_after/"):]].cpu(), fx[k], 1e-5, k)
    enc.eval()
    enc.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in gen.make_params(n, "encoder", 101).items()})
    _close(enc(x).detach().cpu(), fx["enc_eval_logits"], 2e-5, "eval logits")


def test_encoder_layer0_recomputed_equals_stored():
    """Layer 0 recomputed by every pass (option enc_l0_fused = 1, the default: csrc/special.hip enc_l0_kernel) against the
    stored form of rounds 1-2.  Evaluation mode: the same arithmetic in the same order, bit-identical logits.  Training
    mode: the recomputed form takes the layer's batch statistics from the first and second moments of the 3x3 input patches
    (double; the statistics of the exact layer output) where the stored form sums its float32 output, so mean and variance
    agree to float32 rounding, not bit for bit; the backward sums run over another partition of the pixels and take
    sum zhat (x) in_t from the same moments."""
    n, B = 64, 96
    params = gen.make_params(n, "encoder", 121)
    x = torch.from_numpy(gen.make_images(B, 232)).cuda()
    gl = torch.from_numpy(np.random.default_rng(343).standard_normal((B, n)).astype(np.float32)).cuda()
    out = {}
    for fused in (0, 1):
        with _lib.option_scope(enc_l0_fused=fused):
            enc = _load(Encoder(n), params).eval()
            ev = enc(x).detach().cpu()
            enc = _load(Encoder(n), params).train()
            lg = enc(x)
            (lg * gl).sum().backward()
            out[fused] = (ev, lg.detach().cpu(), {k: v.grad.detach().cpu() for k, v in enc.named_parameters()},
                          {k: v.detach().cpu() for k, v in enc.state_dict().items() if "running" in k or "num_batches" in k})
    assert torch.equal(out[0][0], out[1][0])
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert rel(out[1][1], out[0][1]) < 2e-6, rel(out[1][1], out[0][1])
    for k, v in out[0][3].items():
        if "num_batches" in k:
            assert torch.equal(v, out[1][3][k]), k
        else:
            assert float((v.double() - out[1][3][k].double()).abs().max()) <= 2e-7 * float(v.double().abs().max()) + 1e-9, k
    for k, g in out[0][2].items():
        if k.startswith("conv.") and k.endswith(".bias"):
            continue  # a conv bias in front of a BatchNorm: zero true gradient, rounding noise only
        assert rel(out[1][2][k], g) < 2e-5, (k, rel(out[1][2][k], g))


def test_decoder_tail_fused_equals_separate_passes():
    """The 8x8x32 stage's BatchNorm / Dropout2d / LeakyReLU inside the 32 -> 1 layer's kernels (option dec_tail_fused = 1,
    the default: csrc/special.hip DecActIn) against the separate passes of rounds 1-2: the forward is the same arithmetic
    element for element (bit-identical outputs, training with injected masks and evaluation); the backward sums run over
    another partition of the pixels."""
    n, B, R = 64, 12, 4
    params = gen.make_params(n, "decoder", 414)
    spins0 = torch.from_numpy(gen.make_spins(B, R, n, 515)).cuda()
    masks = [torch.from_numpy(m).cuda() for m in gen.make_masks(B * R, 616)]
    go = torch.from_numpy(np.random.default_rng(717).standard_normal((B, R, 1, 32, 32)).astype(np.float32)).cuda()
    out = {}
    for fused in (0, 1):
        with _lib.option_scope(dec_tail_fused=fused):
            dec = _load(Decoder(n), params).eval()
            ev = dec(spins0).detach().cpu()
            dec = _load(Decoder(n), params).train()
            dec.inject_dropout_masks(masks)
            sp = spins0.clone().requires_grad_(True)
            y = dec(sp)
            (y * go).sum().backward()
            out[fused] = (ev, y.detach().cpu(), sp.grad.detach().cpu(), {k: v.grad.detach().cpu() for k, v in dec.named_parameters()})
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert rel(out[1][2], out[0][2]) < 2e-5
    for k, g in out[0][3].items():
        if k.startswith("conv") and k.endswith(".bias") and not k.startswith("conv.4"):
            continue  # conv biases in front of a BatchNorm: zero true gradient, rounding noise only
        assert rel(out[1][3][k], g) < 5e-5, (k, rel(out[1][3][k], g))


@pytest.mark.parametrize("n,B,R", [(64, 12, 4), (128, 300, 8), (64, 5, 1)])
def test_decoder_with_the_mse_fused_behind_it_equals_the_three_separate_calls(n, B, R):
    """``Decoder.forward_mse`` (dvg_decoder_fwd_mse_ex / dvg_decoder_bwd_mse_ex: the reconstruction and its gradient are
    never written) against ``Decoder.forward`` + ``replicated_mse_loss`` + the decoder's backward: the loss to the
    rounding of its double partial sums (they are grouped by image instead of by stride), the gradient wrt the spins and
    EVERY parameter gradient bit for bit -- each value is formed by the arithmetic of the kernel it replaces, in its
    order.  Injected and device-drawn dropout masks; B R above and below the 2048 blocks of the tail's grid."""
    from image_generation_amd import functional as F

    params = gen.make_params(n, "decoder", 414 + n)
    spins0 = torch.from_numpy(gen.make_spins(B, R, n, 515)).cuda()
    images = torch.from_numpy(gen.make_images(B, 818)).cuda()
    for injected in (True, False):
        masks = [torch.from_numpy(m).cuda() for m in gen.make_masks(B * R, 616)] if injected else None
        out = []
        for fused in (False, True):
            dec = _load(Decoder(n), params).train()
            dec.dropout_seed = 4242
            if masks is not None:
                dec.inject_dropout_masks(masks)
            sp = spins0.clone().requires_grad_(True)
            if fused:
                loss = dec.forward_mse(sp, images)
            else:
                loss = F.replicated_mse_loss(dec(sp), images)
            loss.backward()
            out.append((float(loss.detach()), sp.grad.detach().cpu(), {k: v.grad.detach().cpu() for k, v in dec.named_parameters()},
                        {k: v.detach().cpu() for k, v in dec.state_dict().items() if "running" in k}))
        assert abs(out[1][0] - out[0][0]) <= 2e-7 * abs(out[0][0]), (out[0][0], out[1][0])
        assert torch.equal(out[1][1], out[0][1]), "gradient wrt the spins"
        for k, g in out[0][2].items():
            assert torch.equal(out[1][2][k], g), k
        for k, v in out[0][3].items():
            assert torch.equal(out[1][3][k], v), k


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_decoder_with_the_mse_fused_behind_it_in_the_side_operand_modes(mode):
    """The fused tail is float32 arithmetic in every operand mode (its kernels are not GEMM launches): under f32x3 and under
    bf16 GEMM inputs ``forward_mse`` still equals the three separate calls bit for bit in every gradient."""
    from image_generation_amd import functional as F

    n, B, R = 64, 40, 4
    params = gen.make_params(n, "decoder", 515)
    spins0 = torch.from_numpy(gen.make_spins(B, R, n, 6)).cuda()
    images = torch.from_numpy(gen.make_images(B, 9)).cuda()
    _lib.set_conv_precision(mode)
    try:
        out = []
        for fused in (False, True):
            dec = _load(Decoder(n), params).train()
            dec.dropout_seed = 11
            sp = spins0.clone().requires_grad_(True)
            loss = dec.forward_mse(sp, images) if fused else F.replicated_mse_loss(dec(sp), images)
            loss.backward()
            out.append((float(loss.detach()), sp.grad.detach().cpu(), {k: v.grad.detach().cpu() for k, v in dec.named_parameters()}))
    finally:
        _lib.set_conv_precision("f32")
    assert abs(out[1][0] - out[0][0]) <= 2e-7 * abs(out[0][0])
    assert torch.equal(out[1][1], out[0][1])
    for k, g in out[0][2].items():
        assert torch.equal(out[1][2][k], g), k


def test_decoder_forward_mse_fails_loudly_outside_its_domain():
    dec = Decoder(64).cuda().eval()
    sp = torch.from_numpy(gen.make_spins(2, 2, 64, 1)).cuda()
    im = torch.from_numpy(gen.make_images(2, 2)).cuda()
    with pytest.raises(_lib.DvgError):
        dec.forward_mse(sp, im)  # evaluation mode
    dec.train()
    with pytest.raises(ValueError):
        dec.forward_mse(sp, im[:1])  # one image for two rows of spins
    with _lib.option_scope(dec_tail_fused=0):
        with pytest.raises(_lib.DvgError):
            dec.forward_mse(sp, im)  # the fused tail is built on the dec_tail_fused kernels


def test_encoder_winograd_form_and_direct_form_match_the_float64_oracle():
    """The encoder's 3x3 layers in the Winograd form (csrc/conv_wino.hip, conv_wino_wgrad.hip) and as the direct implicit
    GEMM, on whole networks at B = 1024 -- evaluation-mode forward and training-mode forward + backward -- EACH against the
    float64 oracle (oracle/nets.py), under the three policies: never (enc_wino = 0), the default (-1: evaluation launches
    of 256 workgroups' worth or more, training launches of 512 or more: at this size the first two 3x3 layers) and every
    launch (1).  Bars: logits 3e-5 of their range; gradients 5e-3 relative L2, the bar of the full-size step test (the first
    layers' gradients pass through three BatchNorm backward passes, and at 1024 x 341 pooling windows no batch is free of
    float32 near-ties whose routing is rounding noise on either side) -- and no form farther from float64 than 1.5x the
    direct form + 1e-4."""
    n, B = 128, 1024
    params = gen.make_params(n, "encoder", 111)
    x = torch.from_numpy(gen.make_images(B, 222))
    gl = torch.from_numpy(np.random.default_rng(333).standard_normal((B, n)).astype(np.float32))
    p = {k: (torch.from_numpy(np.array(v)).double().requires_grad_("running" not in k) if np.array(v).dtype == np.float32
             else torch.from_numpy(np.array(v))) for k, v in params.items()}
    with torch.no_grad():
        p_eval = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in p.items()}
        want_ev = nets.encoder_forward(p_eval, x.double(), training=False)
    want = nets.encoder_forward(p, x.double(), training=True)
    (want * gl.double()).sum().backward()
    out = {}
    for mode in (0, -1, 1):
        with _lib.option_scope(enc_wino=mode):
            enc = _load(Encoder(n), params).eval()
            ev = enc(x.cuda()).detach().cpu()
            enc = _load(Encoder(n), params).train()
            lg = enc(x.cuda())
            (lg * gl.cuda()).sum().backward()
            out[mode] = (ev, lg.detach().cpu(), {k: v.grad.detach().cpu() for k, v in enc.named_parameters()})
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert not torch.equal(out[-1][0], out[0][0])            # (the default DID take the other kernel for the evaluation call)
    assert not torch.equal(out[-1][1], out[0][1])            # ... and for the large launches of the training call
    assert not torch.equal(out[-1][1], out[1][1])            # ... but not for all of them
    for mode in (0, -1, 1):
        _close(out[mode][0], want_ev, 3e-5, f"evaluation logits, enc_wino={mode}")
        _close(out[mode][1], want.detach(), 3e-5, f"training logits, enc_wino={mode}")
        for k, g in out[mode][2].items():
            if k.startswith("conv") and k.endswith("bias"):
                continue  # conv biases in front of a BatchNorm: zero true gradient, rounding noise only
            e, e0 = rel(g, p[k].grad), rel(out[0][2][k], p[k].grad)
            assert e < 5e-3 and e < 1.5 * e0 + 1e-4, (mode, k, e, e0)


def test_winograd_dynamic_tile_deal_is_bit_identical():
    """Option wino_dynamic (default 1): the Winograd forward / data-gradient launches hand their tile blocks out through an
    atomic counter instead of round-robin.  Which workgroup computes a block must not change a bit: outputs, BatchNorm
    statistics (through the logits) and every gradient equal the static deal's exactly, with all launches switched
    (enc_wino = 1), at a size where workgroups take several blocks each and at one where most get none."""
    from image_generation_amd import _lib
    for n, B in ((128, 1024), (64, 96)):
        params = gen.make_params(n, "encoder", 111)
        x = torch.from_numpy(gen.make_images(B, 222)).cuda()
        gl = torch.from_numpy(np.random.default_rng(333).standard_normal((B, n)).astype(np.float32)).cuda()
        out = {}
        for dyn in (0, 1):
            with _lib.option_scope(enc_wino=1, wino_dynamic=dyn):
                for rep in range(2):  # (twice: the second run starts from the counters the first one left behind)
                    enc = _load(Encoder(n), params).train()
                    lg = enc(x)
                    (lg * gl).sum().backward()
                    torch.cuda.synchronize()
                out[dyn] = (lg.detach().clone(), {k: v.grad.detach().clone() for k, v in enc.named_parameters()})
        assert torch.equal(out[0][0], out[1][0]), (n, B)
        for k, g in out[0][1].items():
            assert torch.equal(g, out[1][1][k]), (n, B, k)


def test_decoder_matches_reference_fixture(fx):
    n, B, R = int(fx["n"]), int(fx["B"]), int(fx["R"])
    dec = _load(Decoder(n), gen.make_params(n, "decoder", 404)).train()
    spins = torch.from_numpy(gen.make_spins(B, R, n, 505)).cuda().requires_grad_(True)
    dec.inject_dropout_masks([torch.from_numpy(m).cuda() for m in gen.make_masks(B * R, 606)])
    go = torch.from_numpy(np.random.default_rng(707).standard_normal((B, R, 1, 32, 32)).astype(np.float32)).cuda()
    y = dec(spins)
    _close(y.detach().cpu(), fx["dec_train_out"], 2e-5, "decoder output")
    (y * go).sum().backward()
    _close(spins.grad.cpu(), fx["dec_grad_spins"], 1e-4, "grad spins")
    for name, prm in dec.named_parameters():
        g = prm.grad.cpu().numpy()
        want_sub = fx[f"dec_grad_sub/{name}"]
        assert np.abs(gen.subsample(g) - want_sub).max() <= 3e-4 * np.abs(want_sub).max() + 5e-5, name
    sd = dec.state_dict()
    for k in fx:
        if k.startswith("dec_after/"):
            _close(sd[k[len("dec_after/"):]].cpu(), fx[k], 1e-5, k)
    dec.eval()
    dec.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in gen.make_params(n, "decoder", 404).items()})
    _close(dec(spins.detach()).detach().cpu(), fx["dec_eval_out"], 2e-5, "eval output")


def _tie_free_images(params, n, B, min_gap=3e-6):
    """Continuous-valued images whose max-pool windows have no near-ties (checked in float64):
    a near-tie's argmax is decided by float32 rounding noise, on the CPU as much as on the GPU."""
    p64 = {k: (torch.from_numpy(np.array(v)).double() if np.array(v).dtype == np.float32 else torch.from_numpy(np.array(v)))
           for k, v in params.items()}
    for seed in range(200):
        x = torch.rand(B, 1, 32, 32, generator=torch.Generator().manual_seed(1000 + seed))
        cap = []
        with torch.no_grad():
            nets.encoder_forward({k: v.clone() for k, v in p64.items()}, x.double(), training=True, capture_prepool=cap)
        if nets.min_pool_gap(cap) > min_gap:
            return x
    raise RuntimeError("no tie-free batch found")


@pytest.mark.parametrize("n,B", [(128, 32), (256, 9), (64, 130), (192, 16), (320, 12), (448, 8)])  # last three: UI sizes
def test_encoder_matches_oracle_full_gradients(n, B):
    params = gen.make_params(n, "encoder", 11 + n)
    enc = _load(Encoder(n), params).train()
    p = {k: (torch.from_numpy(np.array(v)).double().requires_grad_("running" not in k) if np.array(v).dtype == np.float32
             else torch.from_numpy(np.array(v))) for k, v in params.items()}
    x = _tie_free_images(params, n, B, min_gap=3e-6 if B <= 32 else 1e-6)
    gl = torch.randn(B, n, generator=torch.Generator().manual_seed(1))
    want = nets.encoder_forward(p, x.double(), training=True)  # float64 oracle = the exact answer
    (want * gl.double()).sum().backward()
    got = enc(x.cuda())
    (got * gl.cuda()).sum().backward()
    _close(got.detach().cpu(), want.detach(), 3e-5, "logits")
    for name, prm in enc.named_parameters():
        if name.startswith("conv") and name.endswith("bias") and int(name.split(".")[1]) % 4 == 0:
            assert prm.grad.abs().max().item() < 2e-3  # zero true gradient (bias feeding a BatchNorm)
            continue
        _close(prm.grad.cpu(), p[name].grad, 1e-4, name)
    for l in range(4):
        for stat in ("running_mean", "running_var"):
            _close(enc.state_dict()[f"conv.{4*l+1}.{stat}"].cpu(), p[f"conv.{4*l+1}.{stat}"], 1e-5, stat)
        assert int(enc.state_dict()[f"conv.{4*l+1}.num_batches_tracked"]) == int(p[f"conv.{4*l+1}.num_batches_tracked"])


@pytest.mark.parametrize("gamma_edge", [False, True])
def test_encoder_bn_backward_sums_from_the_pooled_activations(gamma_edge):
    """The BatchNorm/pool backward's (sum dz, sum dz zhat) pass reads the pooled activations the forward kept -- zhat of a
    window's maximum is (lrelu^-1(pooled) - beta) / gamma -- instead of the window's four pre-BatchNorm outputs
    (csrc/elementwise.hip, enc_bn_pool_bwd_reduce_p_kernel; dev option enc_bn_reduce_pooled): against the float64 oracle, and
    against the window form on the same inputs.  gamma_edge: some BatchNorm weights zero, tiny and negative -- those channel
    quads take the window form inside the kernel (a zero weight makes the window's first element the maximum, whatever zhat)."""
    n, B = 64, 24
    params = gen.make_params(n, "encoder", 31)
    if gamma_edge:
        for l, idx in ((1, 5), (2, 17), (3, 40)):
            w = np.array(params[f"conv.{4*l+1}.weight"], dtype=np.float32)
            w[idx] = 0.0; w[idx + 1] = 1e-6; w[idx + 8] = -0.7
            params[f"conv.{4*l+1}.weight"] = w
    p = {k: (torch.from_numpy(np.array(v)).double().requires_grad_("running" not in k) if np.array(v).dtype == np.float32
             else torch.from_numpy(np.array(v))) for k, v in params.items()}
    # (a zero weight ties a whole window: no tie-free images exist there, and near-ties may route differently in float64 --
    # the edge case is held against the window form only)
    x = ((torch.rand(B, 1, 32, 32, generator=torch.Generator().manual_seed(3)) < 0.3).float() if gamma_edge
         else _tie_free_images(params, n, B, min_gap=3e-6))
    gl = torch.randn(B, n, generator=torch.Generator().manual_seed(1))
    if not gamma_edge:
        want = nets.encoder_forward(p, x.double(), training=True)
        (want * gl.double()).sum().backward()
    grads = {}
    for form in (1, 0):
        with _lib.option_scope(enc_bn_reduce_pooled=form):
            enc = _load(Encoder(n), params).train()
            (enc(x.cuda()) * gl.cuda()).sum().backward()
            grads[form] = {k: v.grad.clone().cpu() for k, v in enc.named_parameters()}
    for name, g1 in grads[1].items():
        if name.startswith("conv") and name.endswith("bias") and int(name.split(".")[1]) % 4 == 0:
            continue  # (zero true gradient: a bias feeding a BatchNorm)
        if not gamma_edge:
            _close(g1, p[name].grad, 1e-4, name)
        _close(g1, grads[0][name], 2e-5, name + " (pooled form against window form)")


def test_encoder_layer0_moment_statistics_on_real_valued_images():
    """Layer 0's BatchNorm statistics come from the first / second moments of the 3x3 input patches (csrc/special.hip,
    enc_l0_moments_kernel) and its backward takes sum zhat (x) in_t from the same moments: nothing in that algebra needs
    the {0, 1} images of the other tests.  Real-valued images with a large common offset (mean 3, deviation 0.5: the
    variance is a small difference of large moments) against the float64 oracle: batch statistics through the running
    buffers, logits, and every gradient of the stage."""
    n, B = 64, 24
    params = gen.make_params(n, "encoder", 77)
    enc = _load(Encoder(n), params).train()
    p = {k: (torch.from_numpy(np.array(v)).double().requires_grad_("running" not in k) if np.array(v).dtype == np.float32
             else torch.from_numpy(np.array(v))) for k, v in params.items()}
    g = torch.Generator().manual_seed(5)
    x = (3.0 + 0.5 * torch.randn(B, 1, 32, 32, generator=g)).float()
    gl = torch.randn(B, n, generator=g)
    want = nets.encoder_forward(p, x.double(), training=True)
    (want * gl.double()).sum().backward()
    got = enc(x.cuda())
    (got * gl.cuda()).sum().backward()
    for stat in ("running_mean", "running_var"):
        _close(enc.state_dict()[f"conv.1.{stat}"].cpu(), p[f"conv.1.{stat}"], 2e-6, stat)
    _close(got.detach().cpu(), want.detach(), 3e-5, "logits")
    for name in ("conv.0.weight", "conv.1.weight", "conv.1.bias", "conv.4.weight", "projection.weight"):
        _close(dict(enc.named_parameters())[name].grad.cpu(), p[name].grad, 2e-4, name)


def test_encoder_forward_large_batch_folds_bn_partials():
    """B = 288 gives the first layer >= 1024 per-block BatchNorm partial rows, which launch_bn_finalize folds in a
    first pass: forward logits and running statistics against the oracle (forward only: at this many pooling windows
    no batch is free of float32 near-ties, whose gradient routing is rounding noise on either side)."""
    n, B = 64, 288
    params = gen.make_params(n, "encoder", 21)
    enc = _load(Encoder(n), params).train()
    p = _oracle_params(params)
    x = torch.rand(B, 1, 32, 32, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = nets.encoder_forward(p, x, training=True)
        got = enc(x.cuda())
    _close(got.cpu(), want, 5e-5, "logits")
    for l in range(4):
        for stat in ("running_mean", "running_var"):
            _close(enc.state_dict()[f"conv.{4*l+1}.{stat}"].cpu(), p[f"conv.{4*l+1}.{stat}"], 2e-5, stat)


def test_encoder_maxpool_exact_ties_route_to_first_element():
    """Constant images make every interior pooling window an exact tie; torch gives the gradient to the
    first window element (scan order).  A different tie rule changes the weight gradients at O(1)."""
    n, B = 64, 16
    params = gen.make_params(n, "encoder", 5)
    enc = _load(Encoder(n), params).train()
    p = _oracle_params(params)
    vals = torch.linspace(0.0, 1.0, B)
    x = vals[:, None, None, None].expand(B, 1, 32, 32).contiguous()
    gl = torch.randn(B, n, generator=torch.Generator().manual_seed(4))
    want = nets.encoder_forward(p, x, training=True)
    (want * gl).sum().backward()
    got = enc(x.cuda())
    (got * gl.cuda()).sum().backward()
    _close(got.detach().cpu(), want.detach(), 1e-4, "logits")
    for name in ("conv.0.weight", "conv.4.weight", "conv.8.weight", "conv.12.weight", "conv.5.weight"):
        _close(dict(enc.named_parameters())[name].grad.cpu(), p[name].grad, 2e-3, name)


@pytest.mark.parametrize("n,B,R", [(128, 8, 8), (64, 3, 1), (256, 5, 2), (192, 6, 2), (320, 4, 4), (448, 3, 2)])  # last three: UI sizes
def test_decoder_matches_oracle_full_gradients(n, B, R):
    _decoder_vs_oracle(n, B, R)


@pytest.mark.parametrize("n,B,R", [(128, 8, 8), (256, 16, 2), (64, 32, 4)])
def test_decoder_winograd_forms_match_oracle_full_gradients(n, B, R):
    """The Upsample(x2) + ConvTranspose2d 3x3 layers in the Winograd form -- forward, data gradient and weight gradient
    with 9 of the 16 transform positions (csrc/conv_wino.hip UM = 1 / 2, conv_wino_wgrad.hip UPS): the same float64
    oracle comparison as the direct forms, same bars (option dec_wino = 1 forces the forms at these
    sizes; by default they serve decoder batches of 8192 rows or more)."""
    from image_generation_amd import _lib
    with _lib.option_scope(dec_wino=1):
        _decoder_vs_oracle(n, B, R)


def _decoder_vs_oracle(n, B, R):
    params = gen.make_params(n, "decoder", 21 + n)
    dec = _load(Decoder(n), params).train()
    p = _oracle_params(params)
    spins = torch.from_numpy(gen.make_spins(B, R, n, 7)).requires_grad_(True)
    masks = [torch.from_numpy(m) for m in gen.make_masks(B * R, 8)]
    go = torch.randn(B, R, 1, 32, 32, generator=torch.Generator().manual_seed(2))
    want = nets.decoder_forward(p, spins, training=True, dropout_masks=masks)
    (want * go).sum().backward()
    sg = spins.detach().cuda().requires_grad_(True)
    dec.inject_dropout_masks([m.cuda() for m in masks])
    got = dec(sg)
    (got * go.cuda()).sum().backward()
    _close(got.detach().cpu(), want.detach(), 3e-5, "output")
    _close(sg.grad.cpu(), spins.grad, 2e-4, "grad spins")
    for name, prm in dec.named_parameters():
        if name.startswith("convtrans") and name.endswith("bias") and name.split(".")[1] in ("0", "5", "10", "15"):
            assert prm.grad.abs().max().item() < 1e-3
            continue
        _close(prm.grad.cpu(), p[name].grad, 2e-4, name)
    for l in range(4):
        for stat in ("running_mean", "running_var"):
            _close(dec.state_dict()[f"convtrans.{5*l+1}.{stat}"].cpu(), p[f"convtrans.{5*l+1}.{stat}"], 1e-5, stat)


@pytest.mark.parametrize("n,B,R", [(128, 8, 8), (256, 40, 4)])
def test_decoder_gradients_do_not_depend_on_the_staging_form(n, B, R):
    """The LDS-DMA forms of the weight-gradient kernels (3x3, folded upsample + 3x3) keep the register-staged forms'
    accumulation order: bit-identical gradients.  (The Linear layer's 1-tap form uses another tile and K split: float32
    rounding apart; the forward / data-gradient kernels' two forms order their sums differently: 1e-5.)"""
    params = gen.make_params(n, "decoder", 5 + n)
    spins = torch.from_numpy(gen.make_spins(B, R, n, 11)).cuda()
    masks = [torch.from_numpy(m).cuda() for m in gen.make_masks(B * R, 9)]
    go = torch.randn(B, R, 1, 32, 32, generator=torch.Generator().manual_seed(5)).cuda()
    grads = {}
    for form in ("0", "1"):
        with _lib.option_scope(wgrad_dma=int(form)):
            dec = _load(Decoder(n), params).train()
            dec.inject_dropout_masks(masks)
            sg = spins.clone().requires_grad_(True)
            (dec(sg) * go).sum().backward()
            grads[form] = {k: v.grad.clone() for k, v in dec.named_parameters()}
    for k in grads["0"]:
        if k == "increase_latent_dim.weight" and n % 128 == 0:
            _close(grads["1"][k].cpu(), grads["0"][k].cpu(), 2e-6, k)
        else:
            assert torch.equal(grads["0"][k], grads["1"][k]), k


@pytest.mark.parametrize("n,B,R", [(1024, 32, 8), (512, 64, 4)])
def test_decoder_gradients_do_not_depend_on_the_slab_sum_form(n, B, R):
    """c5's shapes (1024 latent spins): the slab sums of the Linear layer's and the dense 2x2 layer's weight gradients -- a few
    slabs of a LARGE weight -- run as tiled one-thread-per-element kernels (csrc/conv_wgrad.hip: wgrad_reduce_tile_kernel,
    wgrad_d22_reduce_pair_kernel); they add the slabs in the 8-lanes-per-element kernels' order: bit-identical gradients."""
    params = gen.make_params(n, "decoder", 5 + n)
    spins = torch.from_numpy(gen.make_spins(B, R, n, 11)).cuda()
    masks = [torch.from_numpy(m).cuda() for m in gen.make_masks(B * R, 9)]
    go = torch.randn(B, R, 1, 32, 32, generator=torch.Generator().manual_seed(5)).cuda()
    grads = {}
    for form in (0, 1):
        with _lib.option_scope(wgrad_reduce_tiled=form):
            dec = _load(Decoder(n), params).train()
            dec.inject_dropout_masks(masks)
            sg = spins.clone().requires_grad_(True)
            (dec(sg) * go).sum().backward()
            grads[form] = {k: v.grad.clone() for k, v in dec.named_parameters()}
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


@pytest.mark.parametrize("n,B,R,gtol", [(32, 7, 2, 2e-5), (64, 5, 3, 2e-5), (128, 16, 8, 2e-5), (256, 33, 4, 2e-5), (128, 64, 8, 2e-3)])
def test_decoder_first_layers_dense_and_composed_forms_equal_the_9_tap_form(n, B, R, gtol):
    """ConvTranspose2d 3x3 on the 2x2 images behind the Linear layer runs as one dense map per image (only the 4 of 9
    taps that land inside the image: 16/36 of the FLOPs; option dec_d22 = 0 keeps the 9-tap implicit GEMM), and for large
    batches the Linear layer and that map are composed into ONE linear map per image whose weight is formed per step,
    the gradients of both original weights following by the chain rule in weight space (option dec_lc0 = 1 forces it on small
    batches, dec_lc0 = 0 keeps the two GEMMs).  Same output, same gradients of every parameter (summation order apart),
    same BatchNorm statistics.  (Gradient bar 2e-5 on the small cases; the 512-image case has enough activations --
    ~10^6 -- that one of them sits within the forms' 1e-6 forward difference of LeakyReLU's kink and takes the other
    slope in the backward pass: measured 3e-4 there, between ANY two forms, so its bar is 2e-3.)"""
    params = gen.make_params(n, "decoder", 77 + n)
    spins = torch.from_numpy(gen.make_spins(B, R, n, 13)).cuda()
    masks = [torch.from_numpy(m).cuda() for m in gen.make_masks(B * R, 10)]
    go = torch.randn(B, R, 1, 32, 32, generator=torch.Generator().manual_seed(6)).cuda()
    forms = {"9tap": dict(dec_d22=0, dec_lc0=0), "dense": dict(dec_lc0=0), "composed": dict(dec_lc0=1)}
    res = {}
    for name, opts in forms.items():
        with _lib.option_scope(**opts):
            dec = _load(Decoder(n), params).train()
            dec.inject_dropout_masks(masks)
            sg = spins.clone().requires_grad_(True)
            out = dec(sg)
            (out * go).sum().backward()
            res[name] = (out.detach().cpu(), sg.grad.cpu(), {k: v.grad.cpu() for k, v in dec.named_parameters()},
                         {k: v.cpu() for k, v in dec.state_dict().items() if "running" in k})
            dec.eval()
            with torch.no_grad():
                res[name] += (dec(spins).cpu(),)
    for name in ("dense", "composed"):
        _close(res[name][0], res["9tap"][0], 1e-5, name + " output")
        _close(res[name][4], res["9tap"][4], 1e-5, name + " eval output")
        _close(res[name][1], res["9tap"][1], gtol, name + " grad spins")
        for k in res["9tap"][2]:
            if k.startswith("convtrans") and k.endswith("bias") and k.split(".")[1] in ("0", "5", "10", "15"):
                continue  # (zero true gradient in front of a BatchNorm: rounding noise only)
            _close(res[name][2][k], res["9tap"][2][k], gtol, name + " " + k)
        for k in res["9tap"][3]:
            _close(res[name][3][k], res["9tap"][3][k], 1e-6, name + " " + k)


def test_position_major_tiles_equal_pixel_major_tiles():
    """Launches whose images fill whole row blocks run position-major tiles (a tile = one pixel position of 64..128
    images; the K loop skips the taps that fall outside the image: conv_igemm.hip) -- plain 3x3 and both folded forms.
    option igemm_posmajor = 0 keeps pixel-major tiles.  Same networks, 1024 images: outputs to 1e-5 (the BatchNorm partial sums
    group other rows), gradients to 2e-3 (LeakyReLU kink, see the composed-form test)."""
    n, B, R = 64, 128, 8
    dparams, eparams = gen.make_params(n, "decoder", 9), gen.make_params(n, "encoder", 10)
    spins = torch.from_numpy(gen.make_spins(B, R, n, 3)).cuda()
    masks = [torch.from_numpy(m).cuda() for m in gen.make_masks(B * R, 4)]
    go = torch.randn(B, R, 1, 32, 32, generator=torch.Generator().manual_seed(8)).cuda()
    imgs = torch.from_numpy(gen.make_images(1024, 5)).cuda()
    gl = torch.randn(1024, n, generator=torch.Generator().manual_seed(9)).cuda()
    res = {}
    for flag in ("1", "0"):  # "1": pixel-major tiles (igemm_posmajor = 0), "0": the default
        with _lib.option_scope(igemm_posmajor=0 if flag == "1" else 1):
            dec = _load(Decoder(n), dparams).train()
            dec.inject_dropout_masks(masks)
            sg = spins.clone().requires_grad_(True)
            out = dec(sg)
            (out * go).sum().backward()
            enc = _load(Encoder(n), eparams).train()
            logits = enc(imgs)
            (logits * gl).sum().backward()
            res[flag] = (out.detach().cpu(), logits.detach().cpu(), sg.grad.cpu(),
                         {"dec." + k: v.grad.cpu() for k, v in dec.named_parameters()} |
                         {"enc." + k: v.grad.cpu() for k, v in enc.named_parameters()})
    _close(res["0"][0], res["1"][0], 1e-5, "decoder output")
    _close(res["0"][1], res["1"][1], 1e-5, "encoder logits")
    _close(res["0"][2], res["1"][2], 2e-3, "grad spins")
    for k in res["0"][3]:
        if k.endswith("bias") and ("convtrans" in k or "conv." in k) and not k.endswith("16.bias") and "increase" not in k:
            continue  # (conv biases in front of a BatchNorm: zero true gradient)
        _close(res["0"][3][k], res["1"][3][k], 2e-3, k)


def test_decoder_device_dropout_is_per_sample_channel_and_reproducible():
    n, B, R = 64, 16, 4
    dec = _load(Decoder(n), gen.make_params(n, "decoder", 3)).train()
    spins = torch.from_numpy(gen.make_spins(B, R, n, 1)).cuda()
    dec.dropout_seed = 99
    a = dec(spins).detach()
    dec._dropout_calls = 0
    b = dec(spins).detach()
    c = dec(spins).detach()
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_cpu_tensors_fail_loudly():
    from image_generation_amd._lib import DvgError
    enc = Encoder(64)
    with pytest.raises(DvgError):
        enc(torch.zeros(2, 1, 32, 32))


def _cos(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def _r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _ConvTBf16(torch.autograd.Function):
    """ConvTranspose2d with "bf16 GEMM inputs, f32 accumulate" in forward and data-gradient, float32 weight gradient."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return torch.nn.functional.conv_transpose2d(_r16(x), _r16(w), b, stride=1, padding=1)

    @staticmethod
    def backward(ctx, gy):
        import torch.nn.functional as F
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xr = _r16(x).detach().requires_grad_(True)
            dx, = torch.autograd.grad(F.conv_transpose2d(xr, _r16(w), None, stride=1, padding=1), xr, _r16(gy))
            wf = w.detach().requires_grad_(True)
            dw, = torch.autograd.grad(F.conv_transpose2d(x.detach(), wf, None, stride=1, padding=1), wf, gy)
        return dx, dw, (gy.sum((0, 2, 3)) if ctx.has_b else None)


def _fold_class_weight(w, pa, pb):
    """The 3x3 kernel a folded layer effectively applies to output pixels of parity class (pa, pb): taps that read the
    same SOURCE pixel of the x2-upsampled input are summed in float32, the sum is rounded to bf16 and sits on one tap
    of the group (the others are zero) -- the arithmetic of the library's WM_CONVT_FOLD_* packs."""
    groups = lambda par: ([[0, 1], [2]] if par == 0 else [[0], [1, 2]])  # noqa: E731  (kernel rows / columns per source row)
    out = torch.zeros_like(w)
    for gr in groups(pa):
        for gc in groups(pb):
            acc = torch.zeros_like(w[:, :, 0, 0])
            for kh in gr:
                for kw in gc:
                    acc = acc + w[:, :, kh, kw]
            out[:, :, gr[0], gc[0]] = _r16(acc)
    return out


class _ConvTFoldBf16(torch.autograd.Function):
    """ConvTranspose2d on a x2-nearest-upsampled map, as the folded bf16-input layers compute it."""

    @staticmethod
    def forward(ctx, x, w, b):
        import torch.nn.functional as F
        ctx.save_for_backward(x, w)
        y = None
        for pa in (0, 1):
            for pb in (0, 1):
                yc = F.conv_transpose2d(_r16(x), _fold_class_weight(w, pa, pb), None, stride=1, padding=1)
                y = torch.zeros_like(yc) if y is None else y
                y[..., pa::2, pb::2] = yc[..., pa::2, pb::2]
        return y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, gy):
        import torch.nn.functional as F
        x, w = ctx.saved_tensors
        gyr = _r16(gy)
        dx = torch.zeros_like(x)
        for pa in (0, 1):
            for pb in (0, 1):
                g = torch.zeros_like(gyr)
                g[..., pa::2, pb::2] = gyr[..., pa::2, pb::2]
                dx = dx + F.conv2d(g, _fold_class_weight(w, pa, pb), None, stride=1, padding=1)
        with torch.enable_grad():
            wf = w.detach().requires_grad_(True)
            dw, = torch.autograd.grad(F.conv_transpose2d(x.detach(), wf, None, stride=1, padding=1), wf, gy)
        return dx, dw, gy.sum((0, 2, 3))


class _LinBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(_r16(x), _r16(w), b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        return _r16(gy) @ _r16(w), gy.flatten(0, -2).t() @ x.flatten(0, -2), gy.flatten(0, -2).sum(0)


def test_bf16_input_mode_decoder_against_an_emulating_oracle(monkeypatch):
    """dvg_set_conv_precision(DVG_PRECISION_BF16_INPUTS): the decoder's Linear and its GEMM-shaped transposed
    convolutions take bf16 operands in forward and data-gradient (f32 accumulate, f32 weight gradients); the 32->1 and
    1->1 layers, BatchNorm and everything else stay float32.  The oracle is run with exactly that arithmetic emulated
    (operands rounded to bf16, float32 convolutions; for the layers the library folds over the x2 upsample, the
    pre-summed taps are what gets rounded).  Against the plain float32 oracle the same run sits at bf16's operand
    precision -- neither equal nor garbage."""
    import types
    import torch.nn.functional as F
    from image_generation_amd import _lib
    L = _lib.lib()
    n, B, R = 128, 16, 4
    params = gen.make_params(n, "decoder", 21 + n)
    masks = [torch.from_numpy(m) for m in gen.make_masks(B * R, 8)]
    go = torch.randn(B, R, 1, 32, 32, generator=torch.Generator().manual_seed(2))

    def oracle(emulate):
        p = _oracle_params(params)
        spins = torch.from_numpy(gen.make_spins(B, R, n, 7)).requires_grad_(True)
        fs = types.SimpleNamespace(**{k: getattr(F, k) for k in dir(F) if not k.startswith("_")})
        if emulate:
            fs.conv_transpose2d = lambda x, w, b, stride=1, padding=1: (  # 128->128 at 2x2; folded 128->64, 64->32; rest f32
                F.conv_transpose2d(x, w, b, stride=stride, padding=padding) if w.shape[1] < 32
                else _ConvTBf16.apply(x, w, b) if x.shape[-1] == 2 else _ConvTFoldBf16.apply(x, w, b))
            fs.linear = lambda x, w, b: _LinBf16.apply(x, w, b)
        monkeypatch.setattr(nets, "F", fs)
        y = nets.decoder_forward(p, spins, training=True, dropout_masks=masks)
        (y * go).sum().backward()
        return y.detach(), spins.grad, {k: v.grad for k, v in p.items() if v.requires_grad}

    y32, gs32, _ = oracle(False)
    y16, gs16, gp16 = oracle(True)
    assert L.dvg_set_conv_precision(1) == 0
    try:
        dec = _load(Decoder(n), params).train()
        sg = torch.from_numpy(gen.make_spins(B, R, n, 7)).cuda().requires_grad_(True)
        dec.inject_dropout_masks([m.cuda() for m in masks])
        got = dec(sg)
        (got * go.cuda()).sum().backward()
        torch.cuda.synchronize()
    finally:
        assert L.dvg_set_conv_precision(0) == 0
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    # (not tighter: an activation within float32 summation noise of a bf16 rounding boundary lands on the other side)
    assert rel(got.detach().cpu(), y16) < 1.5e-3
    assert 1e-4 < rel(got.detach().cpu(), y32) < 3e-2
    assert _cos(sg.grad.cpu(), gs16) > 0.9999 and _cos(sg.grad.cpu(), gs32) < 0.999
    for name, prm in dec.named_parameters():
        if name.endswith("bias") and name.startswith("convtrans") and name.split(".")[1] in ("0", "5", "10", "15"):
            continue                                   # zero true gradient (bias feeding a BatchNorm)
        assert _cos(prm.grad.cpu(), gp16[name]) > 0.9999, name


class _ConvBf16(torch.autograd.Function):
    """Conv2d with bf16 operands in forward and data-gradient, float32 weight gradient."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.conv2d(_r16(x), _r16(w), b, stride=1, padding=1)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        dx = torch.nn.grad.conv2d_input(x.shape, _r16(w), _r16(gy), stride=1, padding=1)
        dw = torch.nn.grad.conv2d_weight(x, w.shape, gy, stride=1, padding=1)
        return dx, dw, gy.sum((0, 2, 3))


def test_bf16_input_mode_encoder_against_an_emulating_oracle(monkeypatch):
    """Encoder counterpart: the 32->64, 64->128 and 128->n convolutions take bf16 operands in forward and data-gradient;
    the 1->32 convolution, the projection, BatchNorm, pooling and the weight gradients stay float32."""
    import types
    import torch.nn.functional as F
    from image_generation_amd import _lib
    L = _lib.lib()
    n, B = 128, 24
    params = gen.make_params(n, "encoder", 11 + n)
    x = (torch.rand(B, 1, 32, 32, generator=torch.Generator().manual_seed(3)) < 0.2).float()
    gl = torch.randn(B, n, generator=torch.Generator().manual_seed(1))

    def oracle(emulate):
        p = _oracle_params(params)
        fs = types.SimpleNamespace(**{k: getattr(F, k) for k in dir(F) if not k.startswith("_")})
        if emulate:
            fs.conv2d = lambda xx, w, b, stride=1, padding=1: (
                _ConvBf16.apply(xx, w, b) if w.shape[1] >= 32 else F.conv2d(xx, w, b, stride=stride, padding=padding))
        monkeypatch.setattr(nets, "F", fs)
        y = nets.encoder_forward(p, x, training=True)
        (y * gl).sum().backward()
        return y.detach(), {k: v.grad for k, v in p.items() if v.requires_grad}

    y32, _ = oracle(False)
    y16, gp16 = oracle(True)
    assert L.dvg_set_conv_precision(1) == 0
    try:
        enc = _load(Encoder(n), params).train()
        got = enc(x.cuda())
        (got * gl.cuda()).sum().backward()
        torch.cuda.synchronize()
    finally:
        assert L.dvg_set_conv_precision(0) == 0
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    assert rel(got.detach().cpu(), y16) < 3e-3 and 1e-4 < rel(got.detach().cpu(), y32) < 5e-2
    for name, prm in enc.named_parameters():
        if name.startswith("conv") and name.endswith("bias") and int(name.split(".")[1]) % 4 == 0:
            continue  # zero true gradient (bias feeding a BatchNorm)
        # (a max-pool window whose two largest bf16-path values swap order between the two float32 summation orders
        # re-routes one gradient element: rare, and the reason this is a cosine rather than an element-wise bound)
        assert _cos(prm.grad.cpu(), gp16[name]) > 0.999, (name, _cos(prm.grad.cpu(), gp16[name]))


def test_operand_mode_must_not_change_between_forward_and_backward():
    from image_generation_amd import _lib
    L = _lib.lib()
    n = 64
    enc = _load(Encoder(n), gen.make_params(n, "encoder", 5)).train()
    x = (torch.rand(4, 1, 32, 32, generator=torch.Generator().manual_seed(3)) < 0.2).float().cuda()
    y = enc(x)
    assert L.dvg_set_conv_precision(1) == 0
    try:
        with pytest.raises(_lib.DvgError, match="operand mode"):
            y.sum().backward()
    finally:
        assert L.dvg_set_conv_precision(0) == 0


def test_wide_tile_configuration_in_a_child_process():
    """The 128x128 convolution tile serves launches of >= 512 such blocks (c3-scale batches); option igemm_thr128 = 1 sends
    every 128-multiple layer of the small fixtures through it.  The whole selection of tests re-runs under it in a child."""
    import os
    import subprocess
    import sys
    if os.environ.get("DVG_TEST_OPTIONS"):
        pytest.skip("already inside the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DVG_TEST_OPTIONS="igemm_thr128=1")  # (tests/conftest.py applies it through dvg_set_option)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_nets.py"),
                        os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-m", "gpu", "-k",
                        "decoder_matches_oracle_full_gradients or matches_reference_fixture or bf16 or conv2d_fwd_dgrad_wgrad"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1200)  # (f32 and bf16-input forms)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("options", ["enc_l0_fused=0,dec_tail_fused=0", "enc_wino=1"])
def test_alternative_layer_forms_in_a_child_process(options):
    """The forms round 3 replaced or made optional stay correct: encoder layer 0 and the decoder's 8x8x32 stage as stored
    tensors with separate passes (rounds 1-2), and the Winograd form of the encoder's 3x3 layers on every launch.  The
    fixture / oracle tests of both networks re-run under the options in a child."""
    import os
    import subprocess
    import sys
    if os.environ.get("DVG_TEST_OPTIONS"):
        pytest.skip("already inside the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DVG_TEST_OPTIONS=options)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_nets.py"), "-q", "-m", "gpu", "-k",
                        "matches_oracle_full_gradients or matches_reference_fixture"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
