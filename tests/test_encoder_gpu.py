"""GPU: the fused inference encoder (include/mapf_dqn.h: mapf_encoder_pack / mapf_encoder_forward,
csrc/mapf_encoder.hip) against plain PyTorch fp32 math of the same network (reference model.py:147-162).

Two references:
  * `ref_fp32`: the encoder in fp32 -- the tolerance is the bf16 one stated in SURVEY 8(c): 2e-2 * max(1, |x|);
  * `ref_emulated`: fp32 convolutions of bf16-rounded weights with the layer outputs rounded to bf16 at exactly
    the kernel's rounding points; what is left is fp32 summation order, i.e. occasional 1-ulp bf16 flips.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _net(seed, bias_scale=0.1):
    from mapf_rl_amd.model import Network

    torch.manual_seed(seed)
    net = Network().cuda()
    with torch.no_grad():  # the reference initialises biases to zero; give them values so that they are tested
        for m in net.obs_encoder.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.bias.uniform_(-bias_scale, bias_scale)
    return net


def _convs(net):
    from mapf_rl_amd.fused import encoder_convs

    return encoder_convs(net.obs_encoder)


def ref_fp32(net, obs):
    x = obs.float()
    c = _convs(net)
    w = [m.weight.detach().float().contiguous() for m in c]
    b = [m.bias.detach().float() for m in c]
    h = F.relu(F.conv2d(x, w[0], b[0]))
    for i in (1, 3, 5):
        t = F.relu(F.conv2d(h, w[i], b[i], padding=1))
        h = F.relu(F.conv2d(t, w[i + 1], b[i + 1], padding=1) + h)
    return F.relu(F.conv2d(h, w[7], b[7])).flatten(1)


def ref_fp32_autograd(net, obs):
    x = obs.float()
    c = _convs(net)
    h = F.relu(F.conv2d(x, c[0].weight, c[0].bias))
    for i in (1, 3, 5):
        t = F.relu(F.conv2d(h, c[i].weight, c[i].bias, padding=1))
        h = F.relu(F.conv2d(t, c[i + 1].weight, c[i + 1].bias, padding=1) + h)
    return F.relu(F.conv2d(h, c[7].weight, c[7].bias)).flatten(1)


def ref_emulated(net, obs):
    """the kernels' rounding points: weights and every layer output in f16, the encoder's output in bf16"""
    r = lambda t: t.to(torch.float16).float()  # noqa: E731
    x = obs.float()
    c = _convs(net)
    w = [r(m.weight.detach().float().contiguous()) for m in c]
    b = [m.bias.detach().float() for m in c]
    h = r(F.relu(F.conv2d(x, w[0], b[0])))
    for i in (1, 3, 5):
        t = r(F.relu(F.conv2d(h, w[i], b[i], padding=1)))
        h = r(F.relu(F.conv2d(t, w[i + 1], b[i + 1], padding=1) + h))
    return F.relu(F.conv2d(h, w[7], b[7])).to(torch.bfloat16).float().flatten(1)


def _run(net, obs):
    from mapf_rl_amd.fused import PackedEncoder, encoder_forward

    wp, bp = PackedEncoder().get(net.obs_encoder)
    return encoder_forward(obs, wp, bp)


@pytest.fixture(autouse=True)
def _fp32_reference_math():
    old = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    yield
    torch.backends.cudnn.allow_tf32 = old


@pytest.mark.parametrize("M", [1, 7, 8, 9, 16, 63, 200, 1029])
def test_encoder_u8_against_fp32(M):
    net = _net(M)
    g = torch.Generator(device="cuda").manual_seed(M)
    obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) < 0.35).to(torch.uint8)
    out = _run(net, obs)
    assert out.shape == (M, 784) and out.dtype == torch.bfloat16
    ref = ref_fp32(net, obs)
    err = (out.float() - ref).abs()
    tol = 2e-2 * torch.clamp(ref.abs(), min=1.0)
    assert bool((err <= tol).all()), "max err %.4g (ref max %.4g)" % (float(err.max()), float(ref.abs().max()))
    emu = ref_emulated(net, obs)
    e2 = (out.float() - emu).abs()
    # same rounding points: almost every element identical; the rest are single-ulp flips (2^-11 relative) of an
    # intermediate f16 activation carried through the later layers into a different bf16 rounding of the output -- bounded here by 4 bf16 ulps
    assert float((e2 > 0).float().mean()) < 0.25
    assert bool((e2 <= 2.0 ** -6 * torch.clamp(emu.abs(), min=0.25)).all()), float(e2.max())
    assert float(ref.abs().max()) > 0.05  # the test is not vacuous


def test_encoder_layout_is_position_and_channel_exact():
    """Weights that make every stage a pure copy/shift, so a transposed tap, a swapped channel or a wrong
    flatten order shows up as an exact mismatch (all values are small integers: exact in bf16)."""
    net = _net(3, bias_scale=0.0)
    c = _convs(net)
    with torch.no_grad():
        for m in c:
            m.weight.zero_()
            m.bias.zero_()
        # conv0: channel co <- input channel (co % 6) at tap (co // 6) % 9 (asymmetric in kh/kw), weight 1 + co % 3
        for co in range(128):
            t = (co // 6) % 9
            c[0].weight[co, co % 6, t // 3, t % 3] = 1 + co % 3
        # res layers: block1 = shift by tap (1,2) from channel (co+1)%128; block2 = tap (2,0) from channel (co+5)%128
        for i in (1, 3, 5):
            for co in range(128):
                c[i].weight[co, (co + 1) % 128, 1, 2] = 1
                c[i + 1].weight[co, (co + 5) % 128, 2, 0] = 1
        for co in range(16):
            c[7].weight[co, (7 * co + 3) % 128, 0, 0] = 1
            c[7].bias[co] = co
    g = torch.Generator(device="cuda").manual_seed(5)
    obs = (torch.rand((37, 6, 9, 9), device="cuda", generator=g) < 0.5).to(torch.uint8)
    out = _run(net, obs)
    ref = ref_fp32(net, obs)
    assert float(ref.max()) <= 256  # integers that bf16 holds exactly
    assert torch.equal(out.float(), ref)


def test_encoder_stays_finite_beyond_the_f16_range():
    """The kernels keep activations in f16: a layer output beyond 65504 is clamped there instead of becoming inf (and NaN one
    layer later through inf * 0 or inf - inf), in the inference and the training forward alike."""
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import PackedEncoder, encoder_forward

    net = _net(5)
    with torch.no_grad():
        for c in _convs(net)[:3]:
            c.weight.mul_(60.0)  # conv0's outputs reach ~10^2, the first block's ~10^5-10^6
    g = torch.Generator(device="cuda").manual_seed(5)
    M = 37
    obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) < 0.5).to(torch.uint8)
    ref = ref_fp32(net, obs)
    assert float(ref.max()) > 65504.0  # the fp32 network does leave the range
    wp, bp = PackedEncoder().get(net.obs_encoder)
    out = encoder_forward(obs, wp, bp)
    assert torch.isfinite(out.float()).all()
    lat = torch.empty((M, 784), dtype=torch.bfloat16, device="cuda")
    acts = torch.empty((7, M, 49, 128), dtype=torch.float16, device="cuda")
    bits = torch.empty((7, M, 49, 4), dtype=torch.int32, device="cuda")
    check(lib.mapf_encoder_forward_save(obs.data_ptr(), 0, M, wp.data_ptr(), bp.data_ptr(), lat.data_ptr(), acts.data_ptr(), bits.data_ptr(), None),
          "mapf_encoder_forward_save")
    assert torch.isfinite(acts.float()).all() and float(acts.float().max()) == 65504.0 and torch.equal(lat, out)


def test_encoder_bf16_input_and_model_path():
    """bf16 observations (the replay gather's output type) and the `Network.encode` switch: without autograd under
    bf16 autocast the fused kernel runs; with autograd the MIOpen path runs; both agree within bf16 tolerance."""
    from mapf_rl_amd.model import Network

    net = _net(11)
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = (torch.rand((300, 6, 9, 9), device="cuda", generator=g) < 0.3)
    a = _run(net, obs.to(torch.uint8))
    b = _run(net, obs.to(torch.bfloat16))
    c = _run(net, obs)  # bool
    assert torch.equal(a, b) and torch.equal(a, c)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.no_grad():
            fused = net.encode(obs.to(torch.uint8))
        unfused = net.encode(obs.to(torch.uint8))
    assert torch.equal(fused, a)
    assert unfused.requires_grad and not fused.requires_grad
    assert torch.allclose(fused.float(), unfused.float(), rtol=3e-2, atol=3e-2)
    # weight updates are picked up (version counter of the parameters)
    with torch.no_grad():
        net.obs_encoder[5].bias.add_(1.0)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fused2 = net.encode(obs.to(torch.uint8))
    d = fused2.float() - fused.float()  # relu(x + 1) - relu(x) lies in [0, 1] and is 1 wherever x > 0
    assert float(d.min()) >= 0 and float(d.max()) <= 1.01 and float(d[fused > 0].min()) > 0.98
    # switch off -> MIOpen path also without autograd
    Network.FUSED_INFERENCE = False
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            plain = net.encode(obs.to(torch.uint8))
    finally:
        Network.FUSED_INFERENCE = True
    assert torch.allclose(fused2.float(), plain.float(), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("M", [300, 301, 2])
def test_encoder_training_path_gradients(M):
    """Fused forward with saved activations + layer-wise backward (autograd.Function) against fp32 autograd of the
    same network: latent within the bf16 tolerance, every parameter gradient as close to the fp32 one as the
    layer-by-layer bf16 path gets (M not a multiple of the kernels' 4 observations per workgroup included)."""
    from mapf_rl_amd.model import Network

    net = _net(21)
    g = torch.Generator(device="cuda").manual_seed(2)
    obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.uint8)
    gl = torch.randn((M, 784), device="cuda", generator=g)
    # fp32 reference
    net.zero_grad()
    ref = ref_fp32_autograd(net, obs)
    (ref * gl).sum().backward()
    gref = {k: p.grad.detach().clone() for k, p in net.obs_encoder.named_parameters()}
    net.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lat = net.encode(obs)
    assert lat.requires_grad and lat.dtype == torch.bfloat16
    (lat.float() * gl).sum().backward()
    err = (lat.float() - ref).abs()
    assert bool((err <= 2e-2 * torch.clamp(ref.abs(), min=1.0)).all())
    gfused = {k: p.grad.detach().clone() for k, p in net.obs_encoder.named_parameters()}
    # the layer-by-layer MIOpen path at the same precision: the yardstick for what bf16 costs on these gradients
    net.zero_grad()
    Network.FUSED_TRAINING = False
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lat2 = net.encode(obs)
        (lat2.float() * gl).sum().backward()
    finally:
        Network.FUSED_TRAINING = True
    assert torch.allclose(lat.float(), lat2.float(), rtol=3e-2, atol=3e-2)
    for k, p in net.obs_encoder.named_parameters():
        a, b, c = gfused[k], gref[k], p.grad
        assert a.shape == b.shape and a.dtype == torch.float32
        rel_fused = float((a - b).norm()) / (float(b.norm()) + 1e-12)
        rel_plain = float((c - b).norm()) / (float(b.norm()) + 1e-12)
        # f16 activations/gradients through up to 8 layers (ReLU masks flip near zero): better than the unfused
        # bf16 path (3 more mantissa bits), and below 5 % of the fp32 gradient's norm in any case
        print(k, "fused %.3e  miopen-bf16 %.3e" % (rel_fused, rel_plain))
        assert rel_fused <= max(rel_plain, 0.01), (k, rel_fused, rel_plain)
        assert rel_fused <= 0.05, (k, rel_fused)


@pytest.mark.parametrize("M", [5, 64, 203])
def test_forward_save_outputs(M):
    """mapf_encoder_forward_save: the latent equals the inference kernel's bit for bit, the saved activations are the
    layer outputs (last one reproduces the latent through the 1x1 layer), and relu_bits are exactly their sign bits."""
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import PackedEncoder, encoder_forward

    net = _net(M)
    g = torch.Generator(device="cuda").manual_seed(M)
    obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) < 0.35).to(torch.uint8)
    wp, bp = PackedEncoder().get(net.obs_encoder)
    lat = torch.empty((M, 784), dtype=torch.bfloat16, device="cuda")
    acts = torch.full((7, M, 49, 128), float("nan"), dtype=torch.float16, device="cuda")
    bits = torch.full((7, M, 49, 4), -1, dtype=torch.int32, device="cuda")
    check(lib.mapf_encoder_forward_save(obs.data_ptr(), 0, M, wp.data_ptr(), bp.data_ptr(), lat.data_ptr(), acts.data_ptr(),
                                        bits.data_ptr(), None), "mapf_encoder_forward_save")
    assert torch.equal(lat, encoder_forward(obs, wp, bp))
    assert torch.isfinite(acts.float()).all() and float(acts.float().min()) >= 0
    # sign bits (include/mapf_dqn.h): channel 32 w + 16 a + 4 h + r <-> bit 8 h + 4 a + r of word w
    pos = (acts.float() > 0).view(7, M, 49, 4, 32).to(torch.int64)
    c = torch.arange(32, device="cuda")
    bitpos = 8 * ((c >> 2) & 3) + 4 * (c >> 4) + (c & 3)
    words = (pos << bitpos).sum(-1)
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    assert torch.equal(bits, words)
    # the layer outputs against the fp32 network, layer by layer (f16 activations: 2^-11 per rounding, a few layers deep)
    c = _convs(net)
    x = obs.float()
    h = F.relu(F.conv2d(x, c[0].weight.float(), c[0].bias))
    outs = [h]
    for i in (1, 3, 5):
        t = F.relu(F.conv2d(h, c[i].weight.float(), c[i].bias, padding=1))
        h = F.relu(F.conv2d(t, c[i + 1].weight.float(), c[i + 1].bias, padding=1) + h)
        outs += [t, h]
    for k, ref in enumerate(outs):
        got = acts[k].float().view(M, 7, 7, 128).permute(0, 3, 1, 2)
        assert bool(((got - ref).abs() <= 4e-3 * torch.clamp(ref.abs(), min=1.0)).all()), k


@pytest.mark.parametrize("M", [1, 2, 3, 161, 1000])
def test_wgrad_kernel_against_fp32(M):
    """mapf_encoder_wgrad on random f16 operands against the fp32 weight gradient of a 3x3 pad-1 convolution
    (the products are exact in fp32, so only the summation order differs); with a loss-scale word the partial sums come
    back multiplied by the float in its second element."""
    from mapf_rl_amd._lib import check, lib

    g = torch.Generator(device="cuda").manual_seed(M)
    gz = (torch.randn((M, 7, 7, 128), device="cuda", generator=g) * (torch.rand((M, 7, 7, 128), device="cuda", generator=g) < 0.5)).to(torch.float16)
    a = torch.relu(torch.randn((M, 7, 7, 128), device="cuda", generator=g)).to(torch.float16)
    ws = torch.full((128, 128, 3, 3, 128), float("nan"), dtype=torch.float32, device="cuda")  # MAPF_ENC_WGRAD_PARTS slabs
    check(lib.mapf_encoder_wgrad(gz.data_ptr(), a.data_ptr(), M, None, ws.data_ptr(), None), "mapf_encoder_wgrad")
    got = ws.sum(0).permute(0, 3, 1, 2)                                         # [co, ci, ky, kx]
    ref = torch.nn.grad.conv2d_weight(a.float().permute(0, 3, 1, 2), (128, 128, 3, 3), gz.float().permute(0, 3, 1, 2), padding=1)
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())
    scale = torch.tensor([0.0, 2.0 ** -9], dtype=torch.float32, device="cuda")
    ws2 = torch.full_like(ws, float("nan"))
    check(lib.mapf_encoder_wgrad(gz.data_ptr(), a.data_ptr(), M, scale.data_ptr(), ws2.data_ptr(), None), "mapf_encoder_wgrad")
    assert torch.equal(ws2, ws * 2.0 ** -9)


@pytest.mark.parametrize("M,layers,parts", [(1, 6, 21), (2, 6, 21), (161, 6, 21), (1000, 6, 21), (3267, 6, 21), (700, 3, 42), (50, 1, 128), (333, 8, 16)])
def test_wgrad_multi_layer_launch_against_fp32_and_the_single_layer_launch(M, layers, parts):
    """mapf_encoder_wgrad_multi -- `layers` 3x3 layers in ONE launch, `parts` observation partitions per layer -- against the fp32
    weight gradient of every layer and against mapf_encoder_wgrad on the same operands (with parts = 128 and one layer: the same
    launch, bit for bit); layers a stride apart that is larger than the arrays (the rows_buffer bucket); the loss-scale word."""
    from mapf_rl_amd._lib import ERR_INVALID_ARG, check, lib

    g = torch.Generator(device="cuda").manual_seed(7 * M + layers)
    pad = 3  # observations of slack between two layers' arrays
    S = (M + pad) * 6272
    gz = (torch.randn((layers, M + pad, 7, 7, 128), device="cuda", generator=g) * (torch.rand((layers, M + pad, 7, 7, 128), device="cuda", generator=g) < 0.5)).to(torch.float16)
    a = torch.relu(torch.randn((layers, M + pad, 7, 7, 128), device="cuda", generator=g)).to(torch.float16)
    ws = torch.full((layers, parts, 128, 3, 3, 128), float("nan"), dtype=torch.float32, device="cuda")
    check(lib.mapf_encoder_wgrad_multi(gz.data_ptr(), S, a.data_ptr(), S, layers, parts, M, None, None, ws.data_ptr(), None), "mapf_encoder_wgrad_multi")
    assert torch.isfinite(ws).all()
    one = torch.empty((128, 128, 3, 3, 128), dtype=torch.float32, device="cuda")
    for l in range(layers):
        got = ws[l].sum(0).permute(0, 3, 1, 2)                                  # [co, ci, ky, kx]
        ref = torch.nn.grad.conv2d_weight(a[l, :M].float().permute(0, 3, 1, 2), (128, 128, 3, 3), gz[l, :M].float().permute(0, 3, 1, 2), padding=1)
        assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), (l, float((got - ref).abs().max()))
        check(lib.mapf_encoder_wgrad(gz[l].data_ptr(), a[l].data_ptr(), M, None, one.data_ptr(), None), "mapf_encoder_wgrad")
        if parts == 128:
            assert torch.equal(ws[l], one)
        else:
            assert float((ws[l].sum(0) - one.sum(0)).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    scale = torch.tensor([0.0, 2.0 ** -9], dtype=torch.float32, device="cuda")
    ws2 = torch.full_like(ws, float("nan"))
    check(lib.mapf_encoder_wgrad_multi(gz.data_ptr(), S, a.data_ptr(), S, layers, parts, M, None, scale.data_ptr(), ws2.data_ptr(), None), "mapf_encoder_wgrad_multi")
    assert torch.equal(ws2, ws * 2.0 ** -9)
    # argument checks: layers that would overlap, a stride that breaks the 16-byte alignment, counts out of range
    if layers > 1:
        assert lib.mapf_encoder_wgrad_multi(gz.data_ptr(), M * 6272 - 8, a.data_ptr(), S, layers, parts, M, None, None, ws.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_wgrad_multi(gz.data_ptr(), S + 4, a.data_ptr(), S, layers, parts, M, None, None, ws.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_wgrad_multi(gz.data_ptr(), S, a.data_ptr(), S, 9, parts, M, None, None, ws.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_wgrad_multi(gz.data_ptr(), S, a.data_ptr(), S, layers, 129, M, None, None, ws.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_wgrad_multi(gz.data_ptr(), S, a.data_ptr(), S, layers, 0, M, None, None, ws.data_ptr(), None) == ERR_INVALID_ARG


@pytest.mark.parametrize("M,valid", [(64, 64), (64, 37), (64, 1), (64, 0), (1000, 555), (4096, 3267), (13, 20)])
def test_bounded_entry_points_equal_the_plain_ones_on_the_valid_rows(M, valid):
    """The `_bounded` encoder entry points (row count read from device memory; the learner's bucket-sized launches) against the plain
    ones run on exactly `valid` rows: forward / forward_save give the same latents, saved activations and ReLU words on the valid rows
    and zeros in the latents and in the last saved layer behind them; backward gives the same gz / gz7 on the valid rows, zeros in gz7
    behind them and the same bias sums; the weight-gradient kernels give the plain kernels' sums -- with every row behind the bound
    poisoned (NaN operands) to prove that nothing reads them."""
    from mapf_rl_amd._lib import ERR_INVALID_ARG, check, lib
    from mapf_rl_amd import fused as F

    net = _net(5)
    wp, bp = F.PackedEncoder().get(net.obs_encoder)
    wpt = F.pack_encoder_backward(net.obs_encoder)
    g = torch.Generator(device="cuda").manual_seed(M * 7 + valid)
    V = min(valid, M)
    obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.bfloat16)
    cnt = torch.tensor([valid], dtype=torch.int32, device="cuda")
    nan16 = lambda shape: torch.full(shape, float("nan"), dtype=torch.float16, device="cuda")
    lat_b = torch.full((M, 784), float("nan"), dtype=torch.bfloat16, device="cuda")
    acts_b, bits_b = nan16((7, M, 49, 128)), torch.full((7, M, 49, 4), -1, dtype=torch.int32, device="cuda")
    check(lib.mapf_encoder_forward_save_bounded(obs.data_ptr(), 1, M, cnt.data_ptr(), wp.data_ptr(), bp.data_ptr(), lat_b.data_ptr(), acts_b.data_ptr(),
                                                bits_b.data_ptr(), None), "mapf_encoder_forward_save_bounded")
    lat_i = torch.full((M, 784), float("nan"), dtype=torch.bfloat16, device="cuda")
    check(lib.mapf_encoder_forward_bounded(obs.data_ptr(), 1, M, cnt.data_ptr(), wp.data_ptr(), bp.data_ptr(), lat_i.data_ptr(), None), "mapf_encoder_forward_bounded")
    assert torch.equal(lat_b[V:], torch.zeros_like(lat_b[V:])) and torch.equal(lat_i[V:], torch.zeros_like(lat_i[V:]))
    assert torch.equal(acts_b[6, V:], torch.zeros_like(acts_b[6, V:]))
    if V:
        lat_p = torch.empty((V, 784), dtype=torch.bfloat16, device="cuda")
        acts_p, bits_p = torch.empty((7, V, 49, 128), dtype=torch.float16, device="cuda"), torch.empty((7, V, 49, 4), dtype=torch.int32, device="cuda")
        check(lib.mapf_encoder_forward_save(obs.data_ptr(), 1, V, wp.data_ptr(), bp.data_ptr(), lat_p.data_ptr(), acts_p.data_ptr(), bits_p.data_ptr(), None), "save")
        assert torch.equal(lat_b[:V], lat_p) and torch.equal(lat_i[:V], lat_p)
        assert torch.equal(acts_b[:, :V], acts_p) and torch.equal(bits_b[:, :V], bits_p)
    # backward + weight gradients: gradient rows behind the bound are NaN too (the learner's are zero: nothing may depend on it)
    gl = torch.randn((M, 784), device="cuda", generator=g).to(torch.bfloat16)
    gl[V:] = float("nan")
    nblk = -(-M // F.ENC_OBS_PER_BLOCK)
    gz_b, gz7_b = nan16((7, M, 49, 128)), nan16((M * 49, 16))
    gb_b, gb7_b = torch.full((7, nblk, 128), float("nan"), device="cuda"), torch.full((4 * nblk, 16), float("nan"), device="cuda")
    sc_b = torch.zeros(2, dtype=torch.int32, device="cuda")
    lat_q = lat_b.clone()
    gl_q = gl.clone()
    gl_q[V:] = 0  # (the loss-scale reduction in front of the chain reads all M rows of the incoming gradient: the caller's padding is zero)
    check(lib.mapf_encoder_backward_bounded(gl_q.data_ptr(), lat_q.data_ptr(), M, cnt.data_ptr(), bits_b.data_ptr(), wpt.data_ptr(), gz_b.data_ptr(),
                                            gb_b.data_ptr(), gz7_b.data_ptr(), gb7_b.data_ptr(), sc_b.data_ptr(), None), "mapf_encoder_backward_bounded")
    assert torch.isfinite(gb_b).all() and torch.isfinite(gb7_b).all()
    assert torch.equal(gz7_b[V * 49:], torch.zeros_like(gz7_b[V * 49:]))
    ws_m = torch.full((6, 21, 128, 3, 3, 128), float("nan"), device="cuda")
    check(lib.mapf_encoder_wgrad_multi(gz_b[1].data_ptr(), M * 6272, acts_b[0].data_ptr(), M * 6272, 6, 21, M, cnt.data_ptr(), sc_b.data_ptr(), ws_m.data_ptr(), None),
          "mapf_encoder_wgrad_multi")
    ws0 = torch.full((F.ENC_WGRAD0_PARTS, 128, 64), float("nan"), device="cuda")
    check(lib.mapf_encoder_wgrad0_bounded(gz_b[0].data_ptr(), obs.data_ptr(), 1, M, cnt.data_ptr(), sc_b.data_ptr(), ws0.data_ptr(), None), "mapf_encoder_wgrad0_bounded")
    assert torch.isfinite(ws_m).all() and torch.isfinite(ws0).all()
    if V:
        gz_p, gz7_p = torch.empty((7, V, 49, 128), dtype=torch.float16, device="cuda"), torch.empty((V * 49, 16), dtype=torch.float16, device="cuda")
        nb = -(-V // F.ENC_OBS_PER_BLOCK)
        gb_p, gb7_p = torch.empty((7, nb, 128), device="cuda"), torch.empty((4 * nb, 16), device="cuda")
        sc_p = torch.zeros(2, dtype=torch.int32, device="cuda")
        check(lib.mapf_encoder_backward(gl[:V].contiguous().data_ptr(), lat_p.data_ptr(), V, bits_p.data_ptr(), wpt.data_ptr(), gz_p.data_ptr(), gb_p.data_ptr(),
                                        gz7_p.data_ptr(), gb7_p.data_ptr(), sc_p.data_ptr(), None), "mapf_encoder_backward")
        assert torch.equal(sc_b, sc_p) and torch.equal(gz_b[:, :V], gz_p) and torch.equal(gz7_b[:V * 49], gz7_p)
        assert torch.allclose(gb_b.sum(1), gb_p.sum(1), rtol=1e-5, atol=1e-6) and torch.allclose(gb7_b.sum(0), gb7_p.sum(0), rtol=1e-5, atol=1e-6)
        ws_p = torch.empty((6, 21, 128, 3, 3, 128), device="cuda")
        check(lib.mapf_encoder_wgrad_multi(gz_p[1].data_ptr(), V * 6272, acts_p[0].data_ptr(), V * 6272, 6, 21, V, None, sc_p.data_ptr(), ws_p.data_ptr(), None), "multi")
        assert torch.equal(ws_m, ws_p)  # (the valid rows are what is partitioned: the same partitions, the same sums)
        ws0_p = torch.empty((F.ENC_WGRAD0_PARTS, 128, 64), device="cuda")
        check(lib.mapf_encoder_wgrad0(gz_p[0].data_ptr(), obs.data_ptr(), 1, V, sc_p.data_ptr(), ws0_p.data_ptr(), None), "wgrad0")
        assert torch.equal(ws0, ws0_p)
    else:
        assert float(ws_m.abs().sum()) == 0.0 and float(ws0.abs().sum()) == 0.0 and float(gb_b.abs().sum()) == 0.0
    # argument checks
    assert lib.mapf_encoder_forward_bounded(obs.data_ptr(), 1, M, None, wp.data_ptr(), bp.data_ptr(), lat_i.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_wgrad0_bounded(gz_b[0].data_ptr(), obs.data_ptr(), 1, M, None, sc_b.data_ptr(), ws0.data_ptr(), None) == ERR_INVALID_ARG


def test_plan_totals():
    from mapf_rl_amd._lib import check, lib

    c = torch.randint(0, 1000, (6, 777), dtype=torch.int32, device="cuda")
    t = torch.full((8,), -1, dtype=torch.int32, device="cuda")
    check(lib.mapf_plan_totals(c.data_ptr(), 6, 777, t.data_ptr(), None), "mapf_plan_totals")
    assert torch.equal(t[:6].long(), c.long().sum(1)) and int(t[6]) == -1


def test_training_step_is_bitwise_repeatable():
    """No atomics anywhere in the fused training path (bias and weight gradients are per-workgroup / per-partition
    partial sums added in a fixed order): two runs give bit-identical latents and parameter gradients."""
    net = _net(31)
    g = torch.Generator(device="cuda").manual_seed(3)
    obs = (torch.rand((777, 6, 9, 9), device="cuda", generator=g) < 0.3).to(torch.uint8)
    gl = torch.randn((777, 784), device="cuda", generator=g)
    runs = []
    for _ in range(2):
        net.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lat = net.encode(obs)
        (lat.float() * gl).sum().backward()
        runs.append((lat.detach().clone(), [p.grad.detach().clone() for p in net.obs_encoder.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)


def test_encoder_argument_checks():
    from mapf_rl_amd._lib import ERR_INVALID_ARG, lib

    assert lib.mapf_encoder_forward(None, 0, 8, None, None, None, None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_forward(None, 0, 0, None, None, None, None) == ERR_INVALID_ARG  # weights are always required
    w = torch.zeros(894976, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(912, dtype=torch.float32, device="cuda")
    o = torch.zeros((8, 6, 9, 9), dtype=torch.uint8, device="cuda")
    out = torch.zeros((8, 784), dtype=torch.bfloat16, device="cuda")
    assert lib.mapf_encoder_forward(o.data_ptr(), 7, 8, w.data_ptr(), b.data_ptr(), out.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_forward(o.data_ptr() + 1, 0, 1, w.data_ptr(), b.data_ptr(), out.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_forward(o.data_ptr(), 0, 0, w.data_ptr(), b.data_ptr(), out.data_ptr(), None) == 0
    assert lib.mapf_encoder_forward(o.data_ptr(), 0, -1, w.data_ptr(), b.data_ptr(), out.data_ptr(), None) == ERR_INVALID_ARG


@pytest.mark.parametrize("M,scale_mag", [(1, 1.0), (5, 3e-6), (162, 1.0), (1001, 2e-3), (9, 700.0)])
def test_backward_with_head_equals_backward_data(M, scale_mag):
    """mapf_encoder_backward (ReLU mask of the 1x1 layer, NCHW -> position-major transposition and the layer's bias partials
    done while the kernel stages its input) against the separate steps feeding mapf_encoder_backward_data: the masked
    gradient gz7 and all seven gz layers bit for bit, bias partial sums to fp32 summation order.  The chain runs on f16
    gradients times the loss scale S = the power of two that puts max |g_latent| into (8, 16]: the reference steps here scale by
    the same S, `scale_mag` moves the whole gradient across binary orders."""
    from mapf_rl_amd._lib import check, lib
    from mapf_rl_amd.fused import ENC_PACKED_BWD_ELEMS, PackedEncoder

    net = _net(M + 7)
    g = torch.Generator(device="cuda").manual_seed(M)
    obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) < 0.35).to(torch.uint8)
    wp, bp = PackedEncoder().get(net.obs_encoder)
    lat = torch.empty((M, 784), dtype=torch.bfloat16, device="cuda")
    acts = torch.empty((7, M, 49, 128), dtype=torch.float16, device="cuda")
    bits = torch.empty((7, M, 49, 4), dtype=torch.int32, device="cuda")
    check(lib.mapf_encoder_forward_save(obs.data_ptr(), 0, M, wp.data_ptr(), bp.data_ptr(), lat.data_ptr(), acts.data_ptr(),
                                        bits.data_ptr(), None), "mapf_encoder_forward_save")
    glat = (torch.randn((M, 784), device="cuda", generator=g) * scale_mag).to(torch.bfloat16)
    import math
    S = 2.0 ** (3 - math.floor(math.log2(float(glat.float().abs().max()))))  # max * S in [8, 16)
    w32 = [c.weight.detach().float().contiguous() for c in _convs(net)]
    import ctypes
    wpt = torch.empty(ENC_PACKED_BWD_ELEMS, dtype=torch.float16, device="cuda")
    check(lib.mapf_encoder_pack_bwd((ctypes.c_void_p * 8)(*[w.data_ptr() for w in w32]), 0, wpt.data_ptr(), None), "mapf_encoder_pack_bwd")
    nblk = -(-M // 4)
    # reference chain: mask + scale + transpose in PyTorch, then the kernel without the head
    gz7_ref = torch.where(lat.view(M, 16, 49) > 0, (glat.float() * S).to(torch.float16).view(M, 16, 49), torch.zeros((), dtype=torch.float16, device="cuda"))
    gz7_ref = gz7_ref.transpose(1, 2).contiguous()                                   # [M][49][16]
    gz_a = torch.full((7, M, 49, 128), float("nan"), dtype=torch.float16, device="cuda")
    gb_a = torch.empty((7, nblk, 128), dtype=torch.float32, device="cuda")
    check(lib.mapf_encoder_backward_data(gz7_ref.data_ptr(), M, bits.data_ptr(), wpt.data_ptr(), gz_a.data_ptr(), gb_a.data_ptr(), None),
          "mapf_encoder_backward_data")
    gz_b = torch.full((7, M, 49, 128), float("nan"), dtype=torch.float16, device="cuda")
    gb_b = torch.empty((7, nblk, 128), dtype=torch.float32, device="cuda")
    gz7 = torch.full((M, 49, 16), float("nan"), dtype=torch.float16, device="cuda")
    gb7 = torch.full((4 * nblk, 16), float("nan"), dtype=torch.float32, device="cuda")
    sw = torch.full((2,), -1, dtype=torch.int32, device="cuda")
    check(lib.mapf_encoder_backward(glat.data_ptr(), lat.data_ptr(), M, bits.data_ptr(), wpt.data_ptr(), gz_b.data_ptr(), gb_b.data_ptr(),
                                    gz7.data_ptr(), gb7.data_ptr(), sw.data_ptr(), None), "mapf_encoder_backward")
    assert int(sw[0]) == int(glat.abs().max().view(torch.int16)) and float(sw.view(torch.float32)[1]) == 1.0 / S
    assert torch.equal(gz7.view(torch.int16), gz7_ref.view(torch.int16))
    assert torch.equal(gz_a.view(torch.int16), gz_b.view(torch.int16))
    assert torch.equal(gb_a * (1.0 / S), gb_b)  # (a power of two: exact)
    assert torch.isfinite(gz_b.float()).all()
    ref7 = gz7_ref.float().sum(dim=(0, 1)) / S
    assert torch.isfinite(gb7).all()
    assert float((gb7.sum(0) - ref7).abs().max()) <= 1e-4 * max(1.0, float(ref7.abs().max()))


def test_backward_argument_checks():
    from mapf_rl_amd._lib import ERR_INVALID_ARG, lib

    t = torch.zeros(4096, dtype=torch.bfloat16, device="cuda")
    p = t.data_ptr()
    assert lib.mapf_encoder_backward(None, None, 4, None, None, None, None, None, None, None, None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_backward(p, p, 4, p, p, p, p, None, p, p, None) == ERR_INVALID_ARG      # gz7 output missing
    assert lib.mapf_encoder_backward(p + 2, p, 4, p, p, p, p, p, p, p, None) == ERR_INVALID_ARG     # misaligned gradient
    assert lib.mapf_encoder_backward(p, p, 4, p, p, p, p, p, p, None, None) == ERR_INVALID_ARG      # no loss-scale word
    assert lib.mapf_encoder_backward(p, p, -1, p, p, p, p, p, p, p, None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_backward(p, p, 0, p, p, p, p, p, p, p, None) == 0
    import ctypes
    sv = (ctypes.c_void_p * 8)(*[p] * 8)
    out = (ctypes.c_void_p * 7)(*[p] * 7)
    assert lib.mapf_recurrent_backward(sv, p, p, p, 0, 1, 4, out, None, 0, None) == ERR_INVALID_ARG        # T < 1
    assert lib.mapf_recurrent_backward(sv, p, p, p, 2, 1, 129, out, None, 0, None) == ERR_INVALID_ARG      # more than 128 agents
    assert lib.mapf_recurrent_backward(sv, p, p, p, 2, 0, 4, out, None, 0, None) == 0                      # no environments: nothing to do
    bad = (ctypes.c_void_p * 7)(*([p] * 6 + [None]))
    assert lib.mapf_recurrent_backward(sv, p, p, p, 2, 1, 4, bad, None, 0, None) == ERR_INVALID_ARG
    assert lib.mapf_recurrent_forward_save(p, None, p, p, p, 2, 1, 129, p, p, sv, None, 0, None) == ERR_INVALID_ARG


@pytest.mark.parametrize("M,dtype", [(1, "u8"), (2, "u8"), (3, "bf16"), (255, "u8"), (257, "bf16"), (1500, "u8")])
def test_wgrad0_kernel_against_fp32(M, dtype):
    """mapf_encoder_wgrad0 (conv0's weight gradient straight from the raw observations) against the fp32 weight gradient
    of a valid 3x3 convolution; the products are exact in fp32, only the summation order differs."""
    from mapf_rl_amd._lib import ERR_INVALID_ARG, check, lib

    g = torch.Generator(device="cuda").manual_seed(M)
    if dtype == "u8":
        obs = (torch.rand((M, 6, 9, 9), device="cuda", generator=g) * 4).to(torch.uint8)   # not only 0/1: any byte is legal
        kind = 0
    else:
        obs = torch.randn((M, 6, 9, 9), device="cuda", generator=g).to(torch.bfloat16)
        kind = 1
    gz = (torch.randn((M, 7, 7, 128), device="cuda", generator=g) * (torch.rand((M, 7, 7, 128), device="cuda", generator=g) < 0.5)).to(torch.float16)
    ws = torch.full((512, 128, 64), float("nan"), dtype=torch.float32, device="cuda")
    check(lib.mapf_encoder_wgrad0(gz.data_ptr(), obs.data_ptr(), kind, M, None, ws.data_ptr(), None), "mapf_encoder_wgrad0")
    scale = torch.tensor([0.0, 2.0 ** -7], dtype=torch.float32, device="cuda")
    ws2 = torch.full_like(ws, float("nan"))
    check(lib.mapf_encoder_wgrad0(gz.data_ptr(), obs.data_ptr(), kind, M, scale.data_ptr(), ws2.data_ptr(), None), "mapf_encoder_wgrad0")
    assert torch.equal(ws2, ws * 2.0 ** -7)
    tot = ws.sum(0)
    assert torch.isfinite(tot).all() and float(tot[:, 54:].abs().max()) == 0.0
    got = tot[:, :54].reshape(128, 6, 3, 3)
    ref = torch.nn.grad.conv2d_weight(obs.float(), (128, 6, 3, 3), gz.float().permute(0, 3, 1, 2))
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())
    assert lib.mapf_encoder_wgrad0(gz.data_ptr(), obs.data_ptr(), 5, M, None, ws.data_ptr(), None) == ERR_INVALID_ARG
    assert lib.mapf_encoder_wgrad0(None, obs.data_ptr(), kind, M, None, ws.data_ptr(), None) == ERR_INVALID_ARG
