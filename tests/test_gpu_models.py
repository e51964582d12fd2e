"""
Model- and workflow-level parity on the GPU: UNet, FAN and the whole ManipulationClassification training step
(forward, loss, every parameter gradient, two Keras-Adam steps) against the CPU float64 oracle on the same seeded
inputs and the same initial weights (copied from the product model into the oracle).
"""
import numpy as np
import pytest
import torch

from oracle import nets as onets
from oracle import tfops as T
from oracle import workflow as owf

from util import assert_close, bayer_from_rgb, collect_from_workers, natural_images, to64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from neural_imaging_amd import _lib
    _lib.load()
    return torch.device('cuda', 0)


def oracle_params(model):
    return {k: to64(v) for k, v in model.state_dict().items()}


def grads_of(model):
    return {k: v.detach().cpu().numpy() for k, v in model._model.g.items()}


def check_grads(got, ref, names, tol=2e-4):
    worst = []
    for k in names:
        a, b = got[k], ref[k].numpy()
        scale = max(np.abs(b).max(), 1e-12)
        e = np.abs(a - b).max() / scale
        worst.append((e, k))
    worst.sort(reverse=True)
    assert worst[0][0] < tol, 'parameter gradients off (rel-to-max): {}'.format(worst[:5])
    return worst[0]


def test_unet_forward_backward(dev):
    from neural_imaging_amd.models import pipelines
    from neural_imaging_amd import ops
    net = pipelines.UNet(patch_size=32, device=dev)
    assert net.count_parameters() == 7763820
    rgb = natural_images(2, 64, 64, seed=21)
    raw = bayer_from_rgb(rgb)
    p = oracle_params(net)
    for v in p.values():
        v.requires_grad_(True)
    y_ref, t_ref = onets.unet_forward(p, to64(raw), return_tensors=True)
    loss_ref = T.mse255(y_ref, to64(rgb))
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))

    x = torch.from_numpy(raw).to(dev)
    y, ctx = net.forward(x, training=True)
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 1e-4, what='UNet output')
    for name in ('ec11', 'ec32', 'ec52', 'dct1', 'dc11', 'dc42'):
        assert_close(ctx[name].cpu().numpy(), t_ref[name].detach().numpy(), 1e-4, 1e-4, what='UNet ' + name)
    loss, dy = ops.mse255(y, torch.from_numpy(rgb).to(dev), grad_scale=1.0)
    assert abs(float(loss.item()) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-4
    net.backward(ctx, dy)
    check_grads(grads_of(net), g_ref, list(p.keys()))
    # reference surface: process() accepts numpy (also a single 3-D image) and answers .numpy()
    out = net.process(raw[0])
    assert out.shape == (1, 64, 64, 3) and np.abs(out.numpy()[0] - y_ref.detach().numpy()[0]).max() < 1e-4


@pytest.mark.parametrize('n', [2, 1])
def test_unet_at_the_full_patch_size(dev, n):
    """BASELINE.json configs[1] at its real patch size (train_nip.py: RAW 128 x 128 x 4 -> 256 x 256 x 3; models/pipelines.py:190-218)
    at a batch the float64 oracle finishes in seconds: the launch shapes of the bench lines - 128 x 128 level 1 (the row-streaming
    kernels in throughput mode), 8 x 8 level 5 - against the oracle.  Parity mode to the 1e-4 contract, forward and every
    parameter gradient; throughput mode by PSNR and gradient direction."""
    from neural_imaging_amd.models import pipelines
    from neural_imaging_amd import ops
    rgb = natural_images(n, 256, 256, seed=23)
    raw = bayer_from_rgb(rgb)
    net = pipelines.UNet(patch_size=128, device=dev)
    p = oracle_params(net)
    for v in p.values():
        v.requires_grad_(True)
    y_ref = onets.unet_forward(p, to64(raw))
    loss_ref = T.mse255(y_ref, to64(rgb))
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))
    x, tgt = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)
    y, ctx = net.forward(x, training=True)
    assert y.shape == (n, 256, 256, 3)
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 1e-4, what='UNet output at 256 x 256')
    loss, dy = ops.mse255(y, tgt, grad_scale=1.0)
    assert abs(float(loss.item()) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-4
    net.backward(ctx, dy)
    worst = check_grads(grads_of(net), g_ref, list(p.keys()))
    ops.set_compute('bf16')
    try:
        nb = pipelines.UNet(patch_size=128, device=dev)
        nb.load_state_dict(net.state_dict())
        yb, cb = nb.forward(x, training=True)
        lb, dyb = ops.mse255(yb, tgt, grad_scale=1.0)
        nb.backward(cb, dyb)
        gb = grads_of(nb)
    finally:
        ops.set_compute('f32')
    psnr = 10 * np.log10(1.0 / np.mean((yb.float().cpu().numpy().astype(np.float64) - y_ref.detach().numpy()) ** 2))
    assert psnr > 45, psnr
    assert abs(float(lb.item()) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-2
    cosines = []
    for k in p:
        if k.endswith('/kernel'):
            a, g = gb[k].ravel().astype(np.float64), g_ref[k].numpy().ravel()
            cosines.append((float(a @ g / (np.linalg.norm(a) * np.linalg.norm(g) + 1e-300)), k))
    cosines.sort()
    lo = cosines[0][0]
    assert lo > 0.97, cosines[:8]
    print('UNet at 128 x 128 RAW: worst parity-mode gradient {:.2e} ({}), throughput mode PSNR {:.1f} dB, worst cosine {:.4f}'.format(
        worst[0], worst[1], psnr, lo))


def test_unet_training_steps_follow_oracle(dev):
    from neural_imaging_amd.models import pipelines
    net = pipelines.UNet(patch_size=16, device=dev)
    rgb = natural_images(4, 32, 32, seed=5)
    raw = bayer_from_rgb(rgb)
    p = oracle_params(net)
    names = list(p.keys())
    m = [torch.zeros_like(v) for v in p.values()]
    v2 = [torch.zeros_like(v) for v in p.values()]
    for step in range(1, 4):
        for t in p.values():
            t.requires_grad_(True)
        loss_ref = T.mse255(onets.unet_forward(p, to64(raw)), to64(rgb))
        gr = torch.autograd.grad(loss_ref, list(p.values()))
        for t in p.values():
            t.requires_grad_(False)
        with torch.no_grad():
            T.adam_step(list(p.values()), list(gr), m, v2, step, 1e-4)
        loss = net.training_step(raw, rgb, learning_rate=1e-4)
        assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 2e-4, (step, float(loss), float(loss_ref.detach()))
    sd = net.state_dict()
    worst = max(np.abs(sd[k] - p[k].numpy()).max() for k in names)
    assert worst < 5e-5, worst           # 3 steps x lr 1e-4: parameters move by <= 3e-4


def test_fan_forward_backward(dev):
    from neural_imaging_amd.models import forensics
    fan = forensics.FAN(n_classes=5, patch_size=64, device=dev)
    assert fan.count_parameters() == 1145382
    x = natural_images(5, 64, 64, seed=33)
    labels = np.array([0, 1, 2, 3, 4], np.int32)
    p = oracle_params(fan)
    for v in p.values():
        v.requires_grad_(True)
    xt = to64(x).requires_grad_(True)
    probs_ref, t_ref = onets.fan_forward(p, xt, return_tensors=True)
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    gr = torch.autograd.grad(loss_ref, list(p.values()) + [xt])
    g_ref = dict(zip(p.keys(), gr[:-1]))

    probs, ctx = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    assert_close(ctx['constrained'].cpu().numpy(), t_ref['constrained'].detach().numpy(), 2e-3, 1e-5, what='residual')
    assert_close(ctx['pool2'].cpu().numpy(), T.max_pool2(t_ref['conv2']).detach().numpy(), 1e-3, 1e-4, what='FAN pool2')
    assert_close(probs.cpu().numpy(), probs_ref.detach().numpy(), 1e-4, what='FAN probabilities')
    loss, dx = fan.backward(ctx, need_input_grad=True)
    assert abs(float(loss.item()) - float(loss_ref.detach())) < 1e-4
    check_grads(grads_of(fan), g_ref, list(p.keys()), tol=3e-4)
    assert_close(dx.cpu().numpy(), gr[-1].numpy(), 1e-7, 3e-4, what='FAN input gradient')
    # reference surface
    dec = fan.process_and_decide(x)
    assert dec.shape == (5,) and (dec == probs_ref.detach().numpy().argmax(axis=1)).all()


@pytest.mark.parametrize('kernel', [7, 9, 11, 4, 6, 8, 10])
def test_fan_large_kernels(dev, kernel):
    """FAN(kernel = 4, 6 ... 11) (forensics.py:51 allows any integer 3 .. 11): the generic float32 matrix-core kernels
    (conv_mfma.hip with a 4-channel K chunk, conv_wgrad.hip in tap passes) behind the same layer code; probabilities, every
    parameter gradient and the input gradient against the float64 oracle, then the same model in throughput mode (these kernel
    sizes stay float32 there) and one training step.  Even sizes: Keras' SAME padding puts the extra row / column after the image
    (oracle/tfops.py same_pads), the input gradient pads the other way round."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import forensics
    fan = forensics.FAN(n_classes=4, patch_size=64, kernel=kernel, n_filters=8, device=dev)
    assert fan.count_parameters() == sum(int(np.prod(v.shape)) for v in fan.state_dict().values())
    assert fan.summary().startswith('{0}x{0} CNN'.format(kernel))
    x = natural_images(4, 64, 64, seed=35)
    labels = np.array([0, 1, 2, 3], np.int32)
    p = oracle_params(fan)
    assert p['conv2/kernel'].shape == (kernel, kernel, 8, 16)
    for v in p.values():
        v.requires_grad_(True)
    xt = to64(x).requires_grad_(True)
    probs_ref = onets.fan_forward(p, xt)
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    gr = torch.autograd.grad(loss_ref, list(p.values()) + [xt])
    g_ref = dict(zip(p.keys(), gr[:-1]))
    probs, ctx = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    assert_close(probs.cpu().numpy(), probs_ref.detach().numpy(), 1e-4, what='FAN probabilities')
    loss, dx = fan.backward(ctx, need_input_grad=True)
    assert abs(float(loss.item()) - float(loss_ref.detach())) < 1e-4
    check_grads(grads_of(fan), g_ref, list(p.keys()), tol=3e-4)
    assert_close(dx.cpu().numpy(), gr[-1].numpy(), 1e-7, 3e-4, what='FAN input gradient')
    try:
        ops.set_compute('bf16')
        probs_b, ctx_b = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
        fan.backward(ctx_b, need_input_grad=True)
        # only the constrained front filter and the 1x1 layer see bf16 operands here
        assert_close(probs_b.cpu().numpy(), probs_ref.detach().numpy(), 3e-2, what='FAN probabilities, throughput mode')
        l0 = float(fan.training_step(x, labels, learning_rate=1e-3))
        for _ in range(10):
            l1 = float(fan.training_step(x, labels, learning_rate=1e-3))
        assert l1 < l0
    finally:
        ops.set_compute('f32')


def test_even_kernel_sizes_are_refused_loudly(dev):
    """INet / ClassicISP pad REFLECT by (kernel - 1) // 2 and convolve VALID (models/pipelines.py:277-280 of the reference): an even
    kernel returns images one pixel short of the targets the reference's own loss compares them with - nothing to match, refused."""
    from neural_imaging_amd.models import pipelines
    with pytest.raises(NotImplementedError):
        pipelines.INet(patch_size=24, kernel=8, device=dev)
    with pytest.raises(NotImplementedError):
        pipelines.ClassicISP(patch_size=24, kernel=4, device=dev)            # residual=True: the bilinear branch
    assert pipelines.ClassicISP(patch_size=24, kernel=4, c_filters=(8,), residual=False, device=dev).model_code.endswith('4x4_8-3')


@pytest.mark.parametrize('n_classes', [24, 200])
def test_fan_with_many_classes(dev, n_classes):
    """FAN(n_classes) beyond the 16 classes one lane of the fused head holds (the reference allows up to 256, forensics.py:37):
    probabilities, loss, every parameter gradient and the input gradient against the float64 oracle."""
    from neural_imaging_amd.models import forensics
    fan = forensics.FAN(n_classes=n_classes, patch_size=32, device=dev)
    x = natural_images(6, 32, 32, seed=35)
    labels = np.array([0, n_classes - 1, 7, 16, 17, n_classes // 2], np.int32)
    p = oracle_params(fan)
    for v in p.values():
        v.requires_grad_(True)
    xt = to64(x).requires_grad_(True)
    probs_ref = onets.fan_forward(p, xt)
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    gr = torch.autograd.grad(loss_ref, list(p.values()) + [xt])
    probs, ctx = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    assert probs.shape == (6, n_classes)
    assert_close(probs.cpu().numpy(), probs_ref.detach().numpy(), 1e-4, what='FAN probabilities')
    loss, dx = fan.backward(ctx, need_input_grad=True)
    assert abs(float(loss.item()) - float(loss_ref.detach())) < 1e-4
    check_grads(grads_of(fan), dict(zip(p.keys(), gr[:-1])), list(p.keys()), tol=3e-4)
    assert_close(dx.cpu().numpy(), gr[-1].numpy(), 1e-7, 3e-4, what='FAN input gradient')


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('patch', [32, 48])
def test_fan_small_patch_fused_vs_separate_pool(dev, fused, patch, monkeypatch):
    """FAN on small inputs (the down-sampled channel): the fused conv+pool path and the separate conv / pool kernels
    must both match the float64 oracle (48 -> 24 -> 12 -> 6 -> 3 exercises the non-fusable odd level too)."""
    from neural_imaging_amd.models import forensics, layers
    if not fused:
        monkeypatch.setattr(layers.Conv2D, 'can_pool', lambda self, x: False)
    fan = forensics.FAN(n_classes=3, patch_size=patch, device=dev)
    x = natural_images(6, patch, patch, seed=41)
    labels = np.array([0, 1, 2, 0, 1, 2], np.int32)
    p = oracle_params(fan)
    for v in p.values():
        v.requires_grad_(True)
    probs_ref = onets.fan_forward(p, to64(x))
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))
    probs, ctx = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    assert_close(probs.cpu().numpy(), probs_ref.detach().numpy(), 1e-4, what='FAN probabilities')
    fan.backward(ctx)
    check_grads(grads_of(fan), g_ref, list(p.keys()), tol=3e-4)


@pytest.mark.parametrize('use_gap,n_dense', [(False, 0), (True, 2), (False, 1)])
def test_fan_head_variants(dev, use_gap, n_dense):
    """FAN heads beyond the workflow default (models/forensics.py:79-87): Flatten instead of GAP, hidden Dense + LeakyReLU
    layers; Keras layer names dense, dense_1, ... with the classifier last."""
    from neural_imaging_amd.models import forensics
    fan = forensics.FAN(n_classes=4, patch_size=32, use_gap=use_gap, n_dense=n_dense, device=dev)
    sd = fan.state_dict()
    cls = 'dense' if n_dense == 0 else 'dense_{}'.format(n_dense)
    assert sd[cls + '/kernel'].shape[1] == 4
    if n_dense == 2:
        assert sd['dense/kernel'].shape == (256, 128) and sd['dense_1/kernel'].shape == (128, 64)
    if not use_gap and n_dense == 0:
        assert sd['dense/kernel'].shape == (2 * 2 * 256, 4)
    x = natural_images(4, 32, 32, seed=77)
    labels = np.array([0, 1, 2, 3], np.int32)
    p = oracle_params(fan)
    for v in p.values():
        v.requires_grad_(True)
    xt = to64(x).requires_grad_(True)
    probs_ref = onets.fan_forward(p, xt, use_gap=use_gap)
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    gr = torch.autograd.grad(loss_ref, list(p.values()) + [xt])
    g_ref = dict(zip(p.keys(), gr[:-1]))
    probs, ctx = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    assert_close(probs.cpu().numpy(), probs_ref.detach().numpy(), 1e-4, what='FAN probabilities')
    loss, dx = fan.backward(ctx, need_input_grad=True)
    assert abs(float(loss.item()) - float(loss_ref.detach())) < 1e-4
    check_grads(grads_of(fan), g_ref, list(p.keys()), tol=3e-4)
    assert_close(dx.cpu().numpy(), gr[-1].numpy(), 1e-7, 3e-4, what='FAN input gradient')
    with pytest.raises(ValueError):
        forensics.FAN(n_classes=4, patch_size=32, dropout=1.0, device=dev)


@pytest.mark.parametrize('use_gap', [True, False])
def test_fan_dropout(dev, use_gap):
    """Keras Dropout after the hidden Dense layers (models/forensics.py:88): with the masks injected, probabilities, loss
    and every gradient equal the oracle's; inference ignores it; the built-in generator keeps ~(1 - rate) of the units and
    gives a different mask every step."""
    from neural_imaging_amd.models import forensics
    rate = 0.4
    fan = forensics.FAN(n_classes=4, patch_size=32, use_gap=use_gap, n_dense=2, dropout=rate, device=dev)
    x = natural_images(6, 32, 32, seed=78)
    labels = np.array([0, 1, 2, 3, 1, 2], np.int32)
    rng = np.random.RandomState(3)
    masks = [(rng.uniform(size=(6, 128)) >= rate).astype(np.uint8), (rng.uniform(size=(6, 64)) >= rate).astype(np.uint8)]
    p = oracle_params(fan)
    for v in p.values():
        v.requires_grad_(True)
    xt = to64(x).requires_grad_(True)
    probs_ref = onets.fan_forward(p, xt, use_gap=use_gap, dropout=rate, dropout_masks=[torch.from_numpy(m) for m in masks])
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    gr = torch.autograd.grad(loss_ref, list(p.values()) + [xt])
    g_ref = dict(zip(p.keys(), gr[:-1]))
    fan.dropout_masks = [torch.from_numpy(m) for m in masks]
    probs, ctx = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    assert fan.dropout_masks is None
    assert_close(probs.cpu().numpy(), probs_ref.detach().numpy(), 1e-4, what='FAN probabilities under dropout')
    loss, dx = fan.backward(ctx, need_input_grad=True)
    assert abs(float(loss.item()) - float(loss_ref.detach())) < 1e-4
    check_grads(grads_of(fan), g_ref, list(p.keys()), tol=3e-4)
    assert_close(dx.cpu().numpy(), gr[-1].numpy(), 1e-7, 3e-4, what='FAN input gradient under dropout')
    # inference: no dropout
    probs_inf = fan.process(x).numpy()
    assert_close(probs_inf, onets.fan_forward(p, to64(x), use_gap=use_gap).detach().numpy(), 1e-4, what='inference')
    # the built-in generator
    _, c1 = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    _, c2 = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    k1, k2 = c1['dense/keep'].float(), c2['dense/keep'].float()
    assert 0.4 < float(k1.mean()) < 0.8 and not torch.equal(k1, k2)
    assert float(fan.training_step(x, labels, learning_rate=1e-3)) > 0


@pytest.mark.parametrize('kernel,cfa', [(5, 'gbrg'), (3, 'rggb'), (7, 'gbrg'), (9, 'bggr'), (11, 'gbrg')])
def test_inet_forward_backward_and_training(dev, kernel, cfa):
    """INet (models/pipelines.py:233-295), the NIP the reference's own framework tests train (config/tests/framework.json):
    output and every trainable gradient against the float64 oracle, frozen up-sampling, then Keras-Adam steps."""
    from neural_imaging_amd.models import pipelines
    net = pipelines.INet(patch_size=24, kernel=kernel, cfa_pattern=cfa, device=dev)
    assert net.model_code == 'INet_{}_{}x{}'.format(cfa, kernel, kernel) and net.count_parameters() == 48 + 9 * kernel ** 2 + 9 + 48 + 39
    rgb = natural_images(3, 48, 48, seed=21)
    raw = bayer_from_rgb(rgb)
    p = oracle_params(net)
    train = [k for k in p if not k.startswith('up/')]
    for k in train:
        p[k].requires_grad_(True)
    y_ref = onets.inet_forward(p, to64(raw))
    loss_ref = T.mse255(y_ref, to64(rgb))
    g_ref = dict(zip(train, torch.autograd.grad(loss_ref, [p[k] for k in train])))
    y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
    assert y.shape == (3, 48, 48, 3)
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 1e-5, what='INet output')
    from neural_imaging_amd import ops
    loss, dy = ops.mse255(y, torch.from_numpy(rgb).to(dev), grad_scale=1.0)
    net.backward(ctx, dy)
    assert abs(float(loss.item()) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-5
    got = grads_of(net)
    check_grads(got, g_ref, train, tol=2e-4)
    assert np.abs(got['up/kernel']).max() == 0
    # training: the loss falls and the frozen kernel does not move
    up0 = net.state_dict()['up/kernel'].copy()
    losses = [float(net.training_step(raw, rgb, learning_rate=1e-3)) for _ in range(25)]
    assert losses[-1] < 0.8 * losses[0], losses
    assert np.array_equal(net.state_dict()['up/kernel'], up0)
    assert net.process(raw[0]).shape == (1, 48, 48, 3)


def test_inet_trainable_upsampling(dev):
    """INet(trainable_upsampling=True) (models/pipelines.py:262-266): the 1x1 CFA up-sampling filter receives its gradient
    through depth_to_space and the REFLECT-padded demosaicing convolution."""
    from neural_imaging_amd.models import pipelines
    from neural_imaging_amd import ops
    net = pipelines.INet(patch_size=24, kernel=5, trainable_upsampling=True, random_init=True, device=dev)
    assert net.model_code == 'INet_gbrgTR_5x5'
    rgb = natural_images(2, 48, 48, seed=23)
    raw = bayer_from_rgb(rgb)
    p = oracle_params(net)
    names = list(p.keys())
    for k in names:
        p[k].requires_grad_(True)
    loss_ref = T.mse255(onets.inet_forward(p, to64(raw)), to64(rgb))
    g_ref = dict(zip(names, torch.autograd.grad(loss_ref, [p[k] for k in names])))
    y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
    loss, dy = ops.mse255(y, torch.from_numpy(rgb).to(dev), grad_scale=1.0)
    net.backward(ctx, dy)
    check_grads(grads_of(net), g_ref, names, tol=3e-4)
    up0 = net.state_dict()['up/kernel'].copy()
    net.training_step(raw, rgb, learning_rate=1e-3)
    assert np.abs(net.state_dict()['up/kernel'] - up0).max() > 0


@pytest.mark.parametrize('kernel,c_filters,residual', [(5, (8, 8), True), (3, (16,), True), (5, (), True), (3, (8,), False),
                                                       (5, (), False), (7, (8,), True), (11, (), True), (9, (8, 8), False),
                                                       (4, (8,), False), (6, (8, 8), False)])
def test_classic_isp_forward_backward_and_training(dev, kernel, c_filters, residual):
    """ClassicISP (models/pipelines.py:416-514) with its DemosaicingLayer (models/layers.py:206-258): output, the gradient
    of alpha and of every CNN parameter against the float64 oracle, camera setters, constants stay frozen."""
    from neural_imaging_amd.models import pipelines
    from neural_imaging_amd import ops
    srgb = np.array([[1.6, -0.4, -0.2], [-0.3, 1.5, -0.2], [0.0, -0.5, 1.5]], np.float32)
    net = pipelines.ClassicISP(patch_size=24, kernel=kernel, c_filters=c_filters, residual=residual, srgb_mat=srgb,
                               cfa_pattern='rggb', device=dev)
    fs = '-'.join(str(c) for c in c_filters)
    assert net.model_code == 'ClassicISP_rggb_{k}x{k}_{fs}-3{r}'.format(k=kernel, fs=fs, r='R' if residual else '')
    chans = (3,) + tuple(c_filters)
    n_cnn = sum(kernel * kernel * a * b + b for a, b in zip(chans[:-1], chans[1:])) + chans[-1] * 3 + 3
    has_cnn = bool(c_filters) or not residual
    assert net.count_parameters() == (1 if residual else 0) + (n_cnn if has_cnn else 0)
    rgb = natural_images(3, 48, 48, seed=31)
    raw = bayer_from_rgb(rgb)
    if has_cnn:                                                # a larger head so that tanh / sigmoid leave the linear range
        w = net._model.p['demosaicing/out/kernel']
        w.mul_(6.0)
        net._model.p['demosaicing/out/bias'].copy_(torch.tensor([0.3, -0.2, 0.1]))
    p = oracle_params(net)
    train = [k for k in net.trainable_names if has_cnn or k != 'demosaicing/alpha']
    for k in train:
        p[k].requires_grad_(True)
    y_ref = onets.classic_isp_forward(p, to64(raw), residual=residual)
    loss_ref = T.mse255(y_ref, to64(rgb))
    y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
    assert y.shape == (3, 48, 48, 3)
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 2e-5, what='ClassicISP output')
    loss, dy = ops.mse255(y, torch.from_numpy(rgb).to(dev), grad_scale=1.0)
    assert abs(float(loss.item()) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-5
    net.backward(ctx, dy)
    got = grads_of(net)
    if train:
        g_ref = dict(zip(train, torch.autograd.grad(loss_ref, [p[k] for k in train])))
        check_grads(got, g_ref, train, tol=3e-4)
    for k in ('up/kernel', 'srgb/kernel') + (('bilinear/kernel',) if residual else ()):
        assert np.abs(got[k]).max() == 0
    if not has_cnn:
        assert np.abs(got['demosaicing/alpha']).max() == 0
    # training moves the CNN (if any) and leaves the constants alone; the setters replace the constants
    net = pipelines.ClassicISP(patch_size=24, kernel=kernel, c_filters=c_filters, residual=residual, cfa_pattern='rggb',
                               device=dev)
    const0 = {k: net.state_dict()[k].copy() for k in ('up/kernel', 'srgb/kernel')}
    target = np.power(rgb, 1 / 2.2).astype(np.float32)
    losses = [float(net.training_step(raw, target, learning_rate=1e-3)) for _ in range(20)]
    assert np.isfinite(losses).all()
    if has_cnn:
        assert losses[-1] < losses[0], losses
    for k, v in const0.items():
        assert np.array_equal(net.state_dict()[k], v)
    out = net.process(raw[0], cfa_pattern='GBRG', srgb_mat=np.eye(3))
    assert out.shape == (1, 48, 48, 3) and net._h.cfa_pattern == 'gbrg'
    assert np.array_equal(net.state_dict()['srgb/kernel'].reshape(3, 3), np.eye(3, dtype=np.float32))
    from neural_imaging_amd.helpers import kernels as hk
    assert np.array_equal(net.state_dict()['up/kernel'].reshape(4, 12), hk.upsampling_kernel('gbrg').reshape(4, 12))


@pytest.mark.parametrize('metric', ['L1', 'SSIM', 'MS-SSIM'])
def test_nip_loss_metrics(dev, metric):
    """NIPModel(loss_metric=...) (models/pipelines.py:53-63): loss value and every UNet gradient of one training step on
    the L1 / SSIM / MS-SSIM loss against autograd through the float64 oracle."""
    from neural_imaging_amd.models import pipelines
    side = 96 if metric == 'MS-SSIM' else 16                 # five scales need >= 176 output pixels
    net = pipelines.UNet(loss_metric=metric, patch_size=side, device=dev)
    rgb = natural_images(2, 2 * side, 2 * side, seed=11)
    raw = bayer_from_rgb(rgb)
    p = oracle_params(net)
    names = list(p.keys())
    for k in names:
        p[k].requires_grad_(True)
    y_ref = onets.unet_forward(p, to64(raw))
    y_ref = y_ref[0] if isinstance(y_ref, tuple) else y_ref
    loss_ref = T.IMAGE_LOSSES[metric](y_ref, to64(rgb))
    g_ref = dict(zip(names, torch.autograd.grad(loss_ref, [p[k] for k in names])))
    y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
    loss, dy = net.loss_and_grad(y, torch.from_numpy(rgb).to(dev))
    assert abs(float(loss.item()) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-5
    assert abs(float(net.loss(y, rgb)) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-5
    from neural_imaging_amd.helpers import tf_helpers as th        # the same values under the reference's function names (:31-44)
    named = {'L1': th.mae, 'SSIM': th.ssim_loss, 'MS-SSIM': th.msssim_loss}[metric]
    assert abs(float(named(y, rgb)) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-5
    net.backward(ctx, dy)
    # 192 x 192: the float32 sums of the UNet's deepest layers alone are off by 4e-4 (L2) .. 1.1e-3 (SSIM) of the layer's
    # largest gradient at this size (measured); the loss gradient itself matches to 1e-5 (test_image_losses_with_gradient)
    check_grads(grads_of(net), g_ref, names, tol=3e-4 if side == 16 else 3e-3)
    l0 = float(net.training_step(raw, rgb, learning_rate=1e-3))
    for _ in range(10):
        l1 = float(net.training_step(raw, rgb, learning_rate=1e-3))
    assert l1 < l0
    with pytest.raises(ValueError):
        pipelines.UNet(loss_metric='L3', patch_size=16, device=dev)


def test_classic_isp_develops_a_reference_ordered_raw(dev):
    """The constants of ClassicISP / INet mean something only for the reference's RAW layout (helpers/raw.py:204-225: an RGGB-
    ordered stack, the CFA pattern says where the planes sit): with that layout the parameter-free ClassicISP - bilinear
    demosaicing, identity colour matrix, gamma - reproduces a smooth image, and INet's hand-set constants come close."""
    from util import stack_bayer
    from neural_imaging_amd.models import pipelines
    yy, xx = np.mgrid[0:64, 0:64] / 64.0
    rgb = np.stack([0.2 + 0.6 * xx, 0.5 + 0.3 * np.sin(3 * yy), 0.7 - 0.5 * xx * yy], axis=-1)[None].astype(np.float32)
    raw = stack_bayer(rgb)
    isp = pipelines.ClassicISP(patch_size=32, device=dev)                  # c_filters=(): no CNN, nothing to train
    y = isp.process(raw).numpy()
    want = np.power(np.clip(rgb, 1 / 255, 1), 1 / 2.2)
    psnr = lambda a, b: 10 * np.log10(1.0 / np.mean((a - b) ** 2))
    assert psnr(y[:, 2:-2, 2:-2], want[:, 2:-2, 2:-2]) > 45
    swapped = isp.process(bayer_from_rgb(rgb)).numpy()                      # the position-ordered stack is NOT that layout
    assert psnr(swapped[:, 2:-2, 2:-2], want[:, 2:-2, 2:-2]) < 25
    right = psnr(y[:, 2:-2, 2:-2], want[:, 2:-2, 2:-2])
    isp.set_cfa_pattern('rggb')                    # a wrong CFA pattern keeps the colours but shifts the samples by a pixel
    assert psnr(isp.process(raw).numpy()[:, 2:-2, 2:-2], want[:, 2:-2, 2:-2]) < right - 3


def test_workflow_with_inet(dev):
    """train_manipulation.py --nip INet --train nip (config/tests/framework.json 'train-manipulation')."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('INet', manipulations=['sharpen:1', 'gaussian:1'], distribution=dist,
                                    trainable={'nip'}, raw_patch_size=32, device=dev)
    rgb = natural_images(4, 64, 64, seed=3)
    raw = bayer_from_rgb(rgb)
    before = wf.nip.state_dict()['demosaic/kernel'].copy()
    l0 = None
    for _ in range(6):
        loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-3)
        l0 = float(loss) if l0 is None else l0
    assert np.isfinite(float(loss)) and float(loss) < l0
    assert np.abs(wf.nip.state_dict()['demosaic/kernel'] - before).max() > 0
    assert wf.run_workflow_to_decisions(raw).shape == (12,)


@pytest.mark.parametrize('metric', ['L1', 'SSIM'])
def test_workflow_nip_loss_metric(dev, metric):
    """ManipulationClassification(loss_metric=...) (workflows/manipulation_classification.py:14-15,268): the lambda_nip term
    of the joint step uses the NIP's configured loss; its gradient reaches the NIP on top of the classification gradient
    (checked against the same step with lambda_nip = 0 plus the stand-alone loss gradient - both passes are linear in it)."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(3, 64, 64, seed=9)
    raw = bayer_from_rgb(rgb)
    grads = {}
    for lam in (0.0, 0.3):
        wf = ManipulationClassification('INet', manipulations=['sharpen:1', 'gaussian:1'], distribution=dist,
                                        trainable={'nip'}, raw_patch_size=32, loss_metric=metric, device=dev)
        loss, parts = wf.training_step(raw, rgb, lambda_nip=lam, learning_rate=0.0)
        grads[lam] = (wf.nip._model.flat_grad.clone(), float(loss), parts)
    net = wf.nip
    y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
    lval, dy = net.loss_and_grad(y, torch.from_numpy(rgb).to(dev), grad_scale=0.3)
    net._model.flat_grad.zero_()
    net.backward(ctx, dy)
    want = grads[0.0][0] + net._model.flat_grad
    got = grads[0.3][0]
    assert float((got - want).abs().max()) <= 2e-4 * float(want.abs().max())
    assert abs(grads[0.3][1] - (grads[0.0][1] + 0.3 * float(lval.item()))) < 1e-3 * abs(grads[0.3][1])
    with pytest.raises(ValueError):
        ManipulationClassification('INet', distribution=dist, loss_metric='L3', device=dev)


def test_workflow_augmentation_strengths(dev):
    """augment=True (workflows/manipulation_classification.py:199-208): one strength per operation and step from the numpy
    global stream, inside the reference's ranges; the same seed reproduces the step, another one does not."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('INet', manipulations=['sharpen', 'resample', 'gaussian', 'jpeg', 'awgn', 'gamma', 'median'],
                                    distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev)
    rgb = natural_images(2, 64, 64, seed=19)
    raw = bayer_from_rgb(rgb)
    outs = []
    for seed in (5, 5, 6):
        np.random.seed(seed)
        m = wf.run_manipulations(wf.nip.process(raw), randomize=True).numpy()
        assert m.shape == (8 * 2, 64, 64, 3) and np.isfinite(m).all()
        probe = np.random.uniform()                      # position of the global stream after the draws
        outs.append((m, probe))
    keep = [i for i, n in enumerate(['native'] + list(wf._operations.keys())) if n != 'awgn']    # awgn adds device noise
    rows = np.concatenate([np.arange(2 * i, 2 * i + 2) for i in keep])
    assert np.array_equal(outs[0][0][rows], outs[1][0][rows]) and outs[0][1] == outs[1][1]
    assert not np.array_equal(outs[0][0][rows], outs[2][0][rows])
    np.random.seed(5)
    loss, _ = wf.training_step(raw, rgb, lambda_nip=0.1, augment=True, learning_rate=1e-4)
    assert np.isfinite(float(loss))
    times = wf.manipulations_timing(wf.nip.process(raw))                  # workflows/...:210-221
    assert list(times.keys()) == list(wf._operations.keys()) and all(0 < t < 5 for t in times.values())


def test_workflow_augmentation_draws_match_the_reference_order(dev):
    """augment=True against the oracle: the reference draws ONE np.random.uniform(*range[name]) per operation
    and step, in the order of its operation table, from numpy's global stream (workflows/manipulation_classification.py:80-90,
    199-208).  The same seed replayed on the host gives the strengths the oracle applies; every manipulated slice must match
    (awgn aside: its noise comes from the device generator), and so must the stream's position afterwards."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    from oracle.workflow import OP_ORDER, apply_manipulation
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    manips = ['median', 'gamma', 'jpeg', 'gaussian', 'resample', 'sharpen', 'awgn']          # given out of order on purpose
    wf = ManipulationClassification('INet', manipulations=manips, distribution=dist, trainable={'nip'}, raw_patch_size=32,
                                    device=dev)
    assert list(wf._operations.keys()) == OP_ORDER
    ranges = {'sharpen': (0.25, 1.5), 'resample': (40, 90), 'gaussian': (0.5, 7), 'jpeg': (50, 90), 'awgn': (1, 5),
              'gamma': (1, 5), 'median': (3, 9)}                                           # workflows/...:80-88
    rgb = natural_images(2, 64, 64, seed=23)
    for seed in (3, 11):
        np.random.seed(seed)
        m = wf.run_manipulations(rgb, randomize=True).numpy()
        after = np.random.uniform()
        np.random.seed(seed)
        drawn = [np.random.uniform(*ranges[n]) for n in OP_ORDER]
        assert np.random.uniform() == after
        assert np.array_equal(m[:2], rgb)
        for k, (name, s) in enumerate(zip(OP_ORDER, drawn)):
            if name == 'awgn':
                continue
            ref = apply_manipulation(name, to64(rgb), s).numpy()
            # hard roundings inside (jpeg index path, soft_quantization of gamma) can flip on float32 ties: almost everywhere
            d = np.abs(m[2 * (k + 1):2 * (k + 2)] - ref)
            assert np.mean(d > 2e-4) < 2e-3 and d.max() < 0.1, (seed, name, s, float(d.max()), float(np.mean(d > 2e-4)))


def test_jpeg_process_with_another_quality(dev):
    """JPEG.process(x, quality != the constructor's) on the device against the oracle: a number, a
    (lo, hi) range resolved by np.random.randint and a list resolved by np.random.choice (models/jpeg.py:210-225) - the draw
    is replayed on the host with the same seed."""
    from neural_imaging_amd.models import jpeg as mj
    from oracle import djpeg as odj
    codec = mj.JPEG(quality=50, codec='soft', device=dev)
    x = natural_images(2, 48, 64, seed=29)
    for q in (90, (30, 70), [20, 55, 85, 95]):
        np.random.seed(4)
        y = codec.process(x, quality=q).numpy()
        np.random.seed(4)
        resolved = odj.resolve_quality(q)
        ref = odj.djpeg_torch(to64(x), resolved, 'soft')[0].numpy()
        d = np.abs(y - ref)
        assert np.mean(d > 2e-4) < 2e-3 and d.max() < 0.1, (q, resolved, float(d.max()))
    assert np.abs(codec.process(x).numpy() - odj.djpeg_torch(to64(x), 50, 'soft')[0].numpy()).max() < 0.1


@pytest.mark.parametrize('n_layers,nf,kernel', [(4, 16, 3), (3, 24, 3), (3, 16, 5), (3, 16, 7), (2, 8, 9), (2, 8, 11)])
def test_dnet_forward_backward(dev, n_layers, nf, kernel):
    """DNet (models/pipelines.py:298-349): VALID conv + ReLU + REFLECT re-pad chains, two-tensor projection, frozen
    up-sampling; output and every trainable gradient against the float64 oracle - for every odd kernel size the reference's
    ParamSpec admits (pipelines.py:308: 3 .. 11; 7 / 9 / 11 through the generic float32 matrix-core kernels)."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import pipelines
    ps = 16 if kernel < 9 else 24          # (the re-pad's gradient folds at most two sources per pixel: size > 2 x pad)
    net = pipelines.DNet(patch_size=ps, n_layers=n_layers, n_features=nf, kernel=kernel, device=dev)
    assert net.model_code == 'DNet_{0}x{0}_{1}x{2}f'.format(kernel, n_layers, nf)
    rgb = natural_images(3, 2 * ps, 2 * ps, seed=23)
    raw = bayer_from_rgb(rgb)
    p = oracle_params(net)
    train = [k for k in p if not k.startswith('up/')]
    for k in train:
        p[k].requires_grad_(True)
    y_ref = onets.dnet_forward(p, to64(raw))
    loss_ref = T.mse255(y_ref, to64(rgb))
    g_ref = dict(zip(train, torch.autograd.grad(loss_ref, [p[k] for k in train])))
    y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
    assert y.shape == (3, 2 * ps, 2 * ps, 3)
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 2e-5, what='DNet output')
    loss, dy = ops.mse255(y, torch.from_numpy(rgb).to(dev), grad_scale=1.0)
    net.backward(ctx, dy)
    got = grads_of(net)
    check_grads(got, g_ref, train, tol=3e-4)
    assert np.abs(got['up/kernel']).max() == 0
    l0 = float(net.training_step(raw, rgb, learning_rate=1e-3))
    for _ in range(10):
        l1 = float(net.training_step(raw, rgb, learning_rate=1e-3))
    assert l1 < l0


def test_fan_bf16_storage_is_bit_neutral(dev):
    """Throughput mode: FAN-internal tensors stored as bf16 vs float32 - identical probabilities, loss, input gradient and
    parameter gradients, bit for bit (the consumers of those tensors round to bf16 either way)."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import forensics
    x = torch.from_numpy(natural_images(6, 64, 64, seed=5)).to(dev)
    labels = torch.from_numpy(np.array([0, 1, 2, 3, 4, 0], np.int32)).to(dev)
    res = {}
    ops.set_compute('bf16')
    try:
        for store in (True, False):
            ops.STORE_BF16 = store
            fan = forensics.FAN(n_classes=5, patch_size=64, device=dev)
            probs, ctx = fan.forward(x, labels, training=True)
            assert (ctx['pool2'].dtype == torch.bfloat16) == store
            loss, dx = fan.backward(ctx, need_input_grad=True)
            res[store] = (probs.cpu().numpy(), float(loss.item()), dx.cpu().numpy(), grads_of(fan))
    finally:
        ops.STORE_BF16 = True
        ops.set_compute('f32')
    a, b = res[True], res[False]
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
    for k in a[3]:
        if k.endswith('/bias') and k.startswith('conv'):      # float32 column sums of the (rounded vs exact) gradient tile
            assert np.allclose(a[3][k], b[3][k], rtol=0, atol=2e-3 * np.abs(b[3][k]).max()), k
        elif k in ('conv2/kernel', 'conv3/kernel', 'conv4/kernel'):
            # the bf16-stored path takes the pooled-gradient weight-gradient kernel (csrc/wgrad5.hip), the float32-stored one the
            # 8-wave kernel on the un-pooled gradient: the same exact bf16 x bf16 products, summed in float32 in another order
            assert np.allclose(a[3][k], b[3][k], rtol=0, atol=2e-5 * np.abs(b[3][k]).max()), k
        else:
            assert np.array_equal(a[3][k], b[3][k]), k


@pytest.mark.parametrize('size', [(2, 64, 64), (3, 32, 48)])
def test_unet_bf16_storage(dev, size):
    """Throughput mode: UNet-internal activations / gradients stored as bf16 vs float32.  The forward pass is bit-neutral (every
    consumer rounds to bf16 operands, takes a sign or a maximum); the backward pass differs only by where a gradient is rounded
    (once at the store instead of at the next operand load, after the skip-gradient add) and by the float32 bias sums."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import pipelines
    n, h, w = size
    rgb = natural_images(n, 2 * h, 2 * w, seed=31)
    raw = torch.from_numpy(bayer_from_rgb(rgb)).to(dev)
    tgt = torch.from_numpy(rgb).to(dev)
    res = {}
    ops.set_compute('bf16')
    try:
        for store in (True, False):
            ops.STORE_BF16 = store
            net = pipelines.UNet(patch_size=h, device=dev)
            y, ctx = net.forward(raw, training=True)
            for name in ('ec11', 'ec12', 'ep1', 'ec52', 'dct1', 'dc11', 'dc42'):
                assert (ctx[name].dtype == torch.bfloat16) == store, name
            assert y.dtype == torch.float32 and (ctx['dc5'] is None or ctx['dc5'].dtype == torch.float32)   # (None: fused into y)
            loss, dy = ops.mse255(y, tgt, grad_scale=1.0)
            net.backward(ctx, dy)
            res[store] = (y.cpu().numpy(), float(loss.item()), grads_of(net), ctx['ec32'].float().cpu().numpy())
    finally:
        ops.STORE_BF16 = True
        ops.set_compute('f32')
    a, b = res[True], res[False]
    assert np.array_equal(a[0], b[0]) and a[1] == b[1], 'forward pass must be bit-neutral'
    worst = 1.0
    for k in a[2]:
        ga, gb = a[2][k].ravel().astype(np.float64), b[2][k].ravel().astype(np.float64)
        cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb) + 1e-300))
        worst = min(worst, cos)
        assert cos > 0.995, (k, cos)
        assert abs(np.linalg.norm(ga) / (np.linalg.norm(gb) + 1e-300) - 1.0) < 2e-2, k
    # the decoder's last layers see identical inputs: their weight gradients agree to the bf16 rounding of one tensor
    assert np.allclose(a[2]['dc5/kernel'], b[2]['dc5/kernel'], rtol=0, atol=1e-6 + 1e-5 * np.abs(b[2]['dc5/kernel']).max())


@pytest.mark.parametrize('patch', [48, 40, 24])
def test_fan_odd_pyramids_in_throughput_mode(dev, patch):
    """Throughput mode on feature pyramids that stop being even (48 -> 24 -> 12 -> 6 -> 3, 40 -> .. -> 5 -> 2): fused layers
    with bf16-stored tensors hand over to the separate conv / pool kernels and back; gradients stay aligned with float32."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import forensics
    x = torch.from_numpy(natural_images(6, patch, patch, seed=43)).to(dev)
    labels = torch.from_numpy(np.array([0, 1, 2, 0, 1, 2], np.int32)).to(dev)
    res = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        try:
            fan = forensics.FAN(n_classes=3, patch_size=patch, device=dev)
            probs, ctx = fan.forward(x, labels, training=True)
            loss, dx = fan.backward(ctx, need_input_grad=True)
            res[mode] = (probs.cpu().numpy(), float(loss.item()), grads_of(fan), dx.cpu().numpy())
        finally:
            ops.set_compute('f32')
    assert np.abs(res['f32'][0] - res['bf16'][0]).max() < 2e-2 and abs(res['f32'][1] - res['bf16'][1]) < 2e-2
    for k in ('conv1/kernel', 'conv3/kernel', 'conv4/kernel', 'conv1x1/kernel', 'dense/kernel'):
        a, b = res['f32'][2][k].ravel(), res['bf16'][2][k].ravel()
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.97, k
    a, b = res['f32'][3].ravel(), res['bf16'][3].ravel()
    assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.97


def _sync_oracle(wf, ref):
    ref.nip = onets.OrderedDict((k, to64(v)) for k, v in wf.nip.state_dict().items())
    ref.fan = onets.OrderedDict((k, to64(v)) for k, v in wf.fan.state_dict().items())


@pytest.mark.parametrize('downsampling', ['bilinear:2', 'pool:2'])
def test_workflow_channel_downsampling(dev, downsampling):
    """The [downsample] stage of the channel (workflows/manipulation_classification.py:231-243): average pooling and
    tf.image.resize(bilinear) to shape[1] // factor, forward and backward through the whole channel."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    manips = ['sharpen:1', 'gaussian:0.83']
    mode = downsampling.split(':')[0] if downsampling.startswith('bilinear') else downsampling
    dist = {'downsampling': downsampling if mode != 'bilinear' else 'bilinear', 'compression': 'jpeg',
            'compression_params': {'quality': 80, 'codec': 'sin'}}
    wf = ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip'},
                                    raw_patch_size=32, device=dev)
    ref = owf.Workflow(manipulations=manips, trainable=('nip',), jpeg_quality=80, jpeg_codec='sin',
                       downsampling=dist['downsampling'])
    _sync_oracle(wf, ref)
    rgb = natural_images(2, 64, 64, seed=18)
    raw = bayer_from_rgb(rgb)
    Y, c, C, ent, probs = wf.run_workflow(raw)
    Yr, cr, Cr, _, pr = ref.run_workflow(to64(raw))
    assert c.shape == (6, 32, 32, 3)
    assert_close(c.numpy(), cr.numpy(), 2e-4, what='down-sampled batch')
    assert_close(C.numpy(), Cr.numpy(), 3e-4, what='codec output')
    assert_close(probs.numpy(), pr.numpy(), 1e-3, what='probabilities')
    loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), 0.1)
    loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
    assert abs(float(parts['ce']) - parts_ref['ce']) < 1e-3
    names = list(ref.fan.keys()) + list(ref.nip.keys())
    got = grads_of(wf.fan)
    got.update(grads_of(wf.nip))
    want = dict(zip(names, grads))
    # The FAN's first layers sit behind its x100 residual high-pass filter: float32-level differences of the 32x32 codec
    # output (<= 3e-4, asserted above) move their gradients by a few per cent on so few pixels.  The FAN itself is pinned
    # at this size by test_fan_small_patch_*; here the UNet gradients - which cross the down-sampling backward - carry
    # the tight bound and the FAN front end a direction check.
    front = ('conv1/kernel', 'conv1/bias', 'conv2/bias', 'conv2/kernel', 'constrained/kernel')
    check_grads(got, want, [k for k in names if k not in front], tol=5e-3)
    for k in front:
        a, b = np.asarray(got[k], np.float64).ravel(), want[k].detach().numpy().ravel()
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.999, k


@pytest.mark.parametrize('trainable', [('nip',), ()])
def test_workflow_training_step_smooth_channel(dev, trainable):
    """Whole-channel parity with a codec whose rounding approximation is continuous ('sin', models/layers.py:125) and
    without the hard-rounding 'jpeg' manipulation: every gradient must match the float64 oracle tightly, for two
    consecutive Keras-Adam steps."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    manips = ['sharpen:1', 'resample:50', 'gaussian:0.83']
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'sin'}}
    wf = ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable=set(trainable),
                                    raw_patch_size=32, device=dev)
    ref = owf.Workflow(manipulations=manips, trainable=trainable, jpeg_quality=80, jpeg_codec='sin')
    _sync_oracle(wf, ref)
    rgb = natural_images(2, 64, 64, seed=8)
    raw = bayer_from_rgb(rgb)
    lam = 0.1
    Y, c, C, ent, probs = wf.run_workflow(raw)
    Yr, cr, Cr, _, pr = ref.run_workflow(to64(raw))
    assert_close(Y.numpy(), Yr.numpy(), 1e-4, what='workflow Y')
    assert_close(c.numpy(), cr.numpy(), 2e-4, what='workflow manipulated batch')
    assert_close(C.numpy(), Cr.numpy(), 3e-4, what='workflow codec output')
    assert_close(probs.numpy(), pr.numpy(), 1e-3, what='workflow probabilities')
    assert c.shape == (8, 64, 64, 3) and probs.shape == (8, 4)
    for step in range(2):
        loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), lam)
        loss, parts = wf.training_step(raw, rgb, lambda_nip=lam, learning_rate=1e-4)
        assert abs(float(parts['ce']) - parts_ref['ce']) < 1e-3, (float(parts['ce']), parts_ref['ce'])
        assert abs(float(parts['nip']) - parts_ref['nip']) / parts_ref['nip'] < 1e-4
        if step == 0:
            names = list(ref.fan.keys()) + (list(ref.nip.keys()) if 'nip' in trainable else [])
            got = grads_of(wf.fan)
            if 'nip' in trainable:
                got.update(grads_of(wf.nip))
            # bias gradients are sums with heavy cancellation and the hard clips of sharpen/gaussian are kinks: 5e-3.
            # (only the first step starts from identical weights: Adam's first update is +-lr per weight whatever the
            # gradient magnitude, so float32-level gradient noise already moves weights apart by up to 2 lr)
            check_grads(got, dict(zip(names, grads)), names, tol=5e-3)
        if ref._m is None:
            ref._m = [torch.zeros_like(p) for p in params]
            ref._v = [torch.zeros_like(p) for p in params]
        ref._t += 1
        with torch.no_grad():
            T.adam_step(params, grads, ref._m, ref._v, ref._t, 1e-4)
    sd = wf.fan.state_dict()
    assert max(np.abs(sd[k] - ref.fan[k].numpy()).max() for k in sd) < 4.5e-4          # <= 2 steps x 2 lr
    expect = float(parts['ce']) + (lam * float(parts['nip']) if 'nip' in trainable else 0)
    assert float(loss) == pytest.approx(expect, rel=1e-5)


def test_workflow_with_trainable_jpeg_tables(dev):
    """The channel with JPEG(trainable=True) as its codec and trainable = {nip, dcn}: the quantisation tables are the codec's
    weights (models/jpeg.py:57-62), the codec's loss term is lambda_dcn * MeanSquaredError(c, C) (models/jpeg.py:197,
    workflows/...:275-277) and one shared Adam step moves FAN, NIP and tables.  'sin' codec (continuous), against the oracle."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    manips = ['resample:50']                  # no hard clip between the NIP and the codec: no kinks for float32 to land on
    dist = {'downsampling': 'none', 'compression': 'jpeg',
            'compression_params': {'quality': 70, 'codec': 'sin', 'trainable': True}}
    wf = ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip', 'dcn'}, raw_patch_size=32,
                                    device=dev)
    ref = owf.Workflow(manipulations=manips, trainable=('nip', 'dcn'), jpeg_quality=70, jpeg_codec='sin', jpeg_trainable=True)
    _sync_oracle(wf, ref)
    rgb = natural_images(2, 64, 64, seed=18)
    raw = bayer_from_rgb(rgb)
    lam, lam_dcn = 0.1, 50.0
    loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), lam, lam_dcn)
    q_before = wf.codec._model.flat.clone()
    loss, parts = wf.training_step(raw, rgb, lambda_nip=lam, lambda_dcn=lam_dcn, learning_rate=1e-3)
    assert abs(float(parts['ce']) - parts_ref['ce']) < 1e-3
    assert abs(float(parts['dcn']) - parts_ref['dcn']) / parts_ref['dcn'] < 1e-3
    names = list(ref.fan.keys()) + list(ref.nip.keys()) + list(ref.jpeg_tables.keys())
    got = grads_of(wf.fan)
    got.update(grads_of(wf.nip))
    got.update(grads_of(wf.codec))
    refg = dict(zip(names, grads))
    # four FAN images only: the float32 noise of the bias / 3-channel filter sums (5e-3 on the 8-image channel tests) is 1 - 2 %
    check_grads(got, refg, list(ref.fan.keys()) + list(ref.nip.keys()), tol=2.5e-2)
    # a table entry's gradient is a sum over every block of terms z cos(2 pi z) - sin(2 pi z) / (2 pi) with z = X / Q up to ~100:
    # signs alternate and the sum is small against its terms, so it is held to 5 % of the largest entry plus the direction
    # (tests/test_gpu_ops.py::test_djpeg_trainable_tables pins the kernel itself at 3e-4 on given gradients)
    for k in ref.jpeg_tables:
        a, b = got[k].ravel().astype(np.float64), refg[k].numpy().ravel()
        assert np.abs(a - b).max() <= 5e-2 * np.abs(b).max(), (k, np.abs(a - b).max() / np.abs(b).max())
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.999, k
    moved = (wf.codec._model.flat - q_before).abs()
    # Adam's first step: lr |g| / (|g| + 1e-7) per entry - up to lr, less where the gradient is of the size of Keras' epsilon
    assert float(moved.max()) <= 1.001e-3 and float(moved.mean()) > 1e-4
    assert float(loss) == pytest.approx(float(parts['ce']) + lam * float(parts['nip']) + lam_dcn * float(parts['dcn']), rel=1e-5)


def test_workflow_training_step_default_channel(dev):
    """The BASELINE configuration (hard rounding in both JPEG stages).  A rounding tie that flips between float32 and
    float64 moves a block by a quantisation step, and the x100 high-pass residual filter of the FAN amplifies it, so
    per-tensor parity is judged by direction (cosine) and the losses; the index path itself is pinned bit-exactly in
    test_gpu_ops.py."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev)
    ref = owf.Workflow(trainable=('nip',), jpeg_quality=80)
    _sync_oracle(wf, ref)
    rgb = natural_images(2, 64, 64, seed=8)
    raw = bayer_from_rgb(rgb)
    Y, c, C, ent, probs = wf.run_workflow(raw)
    Yr, cr, Cr, _, pr = ref.run_workflow(to64(raw))
    assert_close(Y.numpy(), Yr.numpy(), 1e-4, what='workflow Y')
    dC = np.abs(C.numpy() - Cr.numpy())
    assert np.mean(dC > 1e-3) < 5e-3, np.mean(dC > 1e-3)
    assert c.shape == (10, 64, 64, 3) and probs.shape == (10, 5)
    loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), 0.1)
    loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
    assert abs(float(parts['ce']) - parts_ref['ce']) < 5e-3
    assert abs(float(parts['nip']) - parts_ref['nip']) / parts_ref['nip'] < 1e-3
    names = list(ref.fan.keys()) + list(ref.nip.keys())
    got = grads_of(wf.fan)
    got.update(grads_of(wf.nip))
    for k, gr in zip(names, grads):
        a, b = got[k].ravel().astype(np.float64), gr.numpy().ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
        assert cos > 0.98, (k, cos)
    # the reference's NaN guard (workflows/...:281-282): poison one weight, the step must raise and not update
    wf.fan._model.p['dense/bias'][0] = float('nan')
    before = wf.nip.state_dict()['ec11/kernel'].copy()
    with pytest.raises(RuntimeError):
        wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
    assert np.array_equal(before, wf.nip.state_dict()['ec11/kernel'])


def test_default_channel_parity_outside_flipped_blocks(dev):
    """Channel-level 1e-4 parity on the BASELINE configuration (hard rounding in the 'jpeg' manipulation AND in the channel
    codec; workflows/manipulation_classification.py:260-285, models/jpeg.py:129-131).  The only legitimate float32 / float64
    divergence is a rounding tie of X / Q that falls the other way: both sides return their two integer index tensors, the
    8x8 blocks whose indices differ are identified and masked, and everything else is held to the contract:
      * per batch: flip rate < 1e-4 of the coefficients, |C - C_oracle| <= 1e-4 on every block without a flip;
      * on a batch WITHOUT any flip (the seeds are walked until one is found): losses 1e-4, every parameter gradient of the FAN
        and the UNet within GRAD_TOL relative to its largest entry."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    from oracle import djpeg as odj
    GRAD_TOL = 1e-3
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev)
    ref = owf.Workflow(trainable=('nip',), jpeg_quality=80)
    _sync_oracle(wf, ref)
    qt = ops.qtables_device(80, dev)
    jpeg_class = 1 + ref.operations.index('jpeg')
    clean = None
    rates = []
    for seed in range(40, 56):
        rgb = natural_images(1, 64, 64, seed=seed)
        raw = bayer_from_rgb(rgb)
        b = raw.shape[0]
        Y, c, C, ent, probs = wf.run_workflow(raw)
        Yr, cr, Cr, _, pr = ref.run_workflow(to64(raw))
        assert_close(Y.numpy(), Yr.numpy(), 1e-4, what='workflow Y')
        # the two index tensors of each side: stage 1 = the 'jpeg' manipulation of Y, stage 2 = the channel codec on c
        i1 = ops.djpeg_fwd(torch.as_tensor(Y.numpy()).to(dev), qt, 'soft', want_idx=True)[2].cpu().numpy()
        i2 = ops.djpeg_fwd(torch.as_tensor(c.numpy()).to(dev), qt, 'soft', want_idx=True)[2].cpu().numpy()
        r1 = odj.djpeg_torch(Yr, 80, 'soft')[2].numpy()
        r2 = odj.djpeg_torch(cr, 80, 'soft')[2].numpy()
        f1, f2 = i1 != np.rint(r1), i2 != np.rint(r2)                # (n, 3, hb, wb, 8, 8)
        rate = (f1.sum() + f2.sum()) / float(f1.size + f2.size)
        rates.append(rate)
        assert rate < 1e-4, (seed, rate)
        bad = f2.any(axis=(1, 4, 5))                                  # (5 b, hb, wb) blocks of the codec's input / output
        bad[jpeg_class * b:(jpeg_class + 1) * b] |= f1.any(axis=(1, 4, 5))
        keep = ~np.kron(bad, np.ones((1, 8, 8), bool))[..., None]     # per pixel
        dC = np.abs(C.numpy().astype(np.float64) - Cr.numpy()) * keep
        assert dC.max() <= 1e-4, (seed, float(dC.max()), int(bad.sum()))
        if not bad.any():
            assert_close(probs.numpy(), pr.numpy(), 1e-4, what='probabilities of a batch without a flipped tie')    # 5 classes, float32 FAN
            clean = (seed, raw, rgb)
            break
    assert clean is not None, 'no batch without a rounding flip in 16 seeds: rates {}'.format(rates)
    seed, raw, rgb = clean
    loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), 0.1)
    # the same restatement evaluated in float32 - the arithmetic the reference itself runs in (TF2 CPU, float32): how far ITS
    # gradients sit from the exact ones bounds what any float32 implementation can be asked for.  On this channel that is not
    # small everywhere: the FAN's x100 prediction-error filter is ~0 on smooth content, the first LeakyReLU / MaxPool2D behind it
    # then decide on rounding noise, and conv1/bias, constrained/kernel (sums over those pixels) move by 1 - 30 % between float32
    # and float64 while every other tensor agrees to < 1e-3 (profiles/r05_default_channel_parity.txt).
    ref32 = owf.Workflow(trainable=('nip',), jpeg_quality=80, dtype=torch.float32)
    ref32.nip = onets.OrderedDict((k, v.to(torch.float32)) for k, v in ref.nip.items())
    ref32.fan = onets.OrderedDict((k, v.to(torch.float32)) for k, v in ref.fan.items())
    _, _, _, grads32, _ = ref32.loss_and_grads(torch.tensor(raw), torch.tensor(rgb), 0.1)
    loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
    assert abs(float(parts['ce']) - parts_ref['ce']) < 1e-4, (float(parts['ce']), parts_ref['ce'])
    assert abs(float(parts['nip']) - parts_ref['nip']) / parts_ref['nip'] < 1e-4
    names = list(ref.fan.keys()) + list(ref.nip.keys())
    got = grads_of(wf.fan)
    got.update(grads_of(wf.nip))
    rows, loose = [], []
    for k, g64, g32 in zip(names, grads, grads32):
        b = g64.numpy()
        scale = max(np.abs(b).max(), 1e-12)
        e_prod = np.abs(got[k].astype(np.float64) - b).max() / scale
        e_ref32 = np.abs(g32.numpy().astype(np.float64) - b).max() / scale
        rows.append((e_prod, e_ref32, k))
        if e_ref32 > GRAD_TOL / 1.5:
            loose.append(k)
        assert e_prod <= max(GRAD_TOL, 1.5 * e_ref32), (k, e_prod, e_ref32)
    rows.sort(reverse=True)
    # the ill-conditioned ones are the FAN's front end (and, by seed, a bias or two behind it) - never the bulk of the tensors
    assert len(loose) <= 6 and all(k.split('/')[0] in ('constrained', 'conv1', 'conv2', 'conv3') or k.endswith('/bias')
                                   for k in loose), loose
    print('default channel, seed {} (no flipped tie), flip rates walked {}'.format(seed, ['%.1e' % r for r in rates]))
    for e_prod, e_ref32, k in rows[:8]:
        print('   {:22s} product vs float64 {:.2e}   float32 restatement vs float64 {:.2e}'.format(k, e_prod, e_ref32))


def test_full_patch_size_channel_against_the_oracle(dev):
    """BASELINE.json configs[3] at its FULL patch size (RAW 128 x 128 -> 256 x 256 x 3 images; the reference's
    workflows/manipulation_classification.py:260-285 with hard rounding in both JPEG stages) against the float64 oracle, at a
    batch the oracle finishes in seconds (B = 2 -> ten FAN images): the 128-pixel-wide UNet level, the 256-pixel image chain and
    the 32 x 32 blocks-per-plane dJPEG are the launch shapes of the bench line, which the reduced-size tests do not reach.
    Forward to the 1e-4 contract outside 8 x 8 blocks with a flipped rounding tie; losses; gradient directions (at ~4 M coefficients
    a batch without any flip does not exist - the per-tensor 1e-3 check is test_default_channel_parity_outside_flipped_blocks)."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    from oracle import djpeg as odj
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=128, device=dev)
    ref = owf.Workflow(trainable=('nip',), jpeg_quality=80)
    _sync_oracle(wf, ref)
    qt = ops.qtables_device(80, dev)
    jpeg_class = 1 + ref.operations.index('jpeg')
    rgb = natural_images(2, 256, 256, seed=8)
    raw = bayer_from_rgb(rgb)
    b = raw.shape[0]
    Y, c, C, ent, probs = wf.run_workflow(raw)
    Yr, cr, Cr, _, pr = ref.run_workflow(to64(raw))
    assert Y.shape == (b, 256, 256, 3) and C.shape == (5 * b, 256, 256, 3)
    assert_close(Y.numpy(), Yr.numpy(), 1e-4, what='UNet output at 256 x 256')
    i1 = ops.djpeg_fwd(torch.as_tensor(Y.numpy()).to(dev), qt, 'soft', want_idx=True)[2].cpu().numpy()
    i2 = ops.djpeg_fwd(torch.as_tensor(c.numpy()).to(dev), qt, 'soft', want_idx=True)[2].cpu().numpy()
    f1 = i1 != np.rint(odj.djpeg_torch(Yr, 80, 'soft')[2].numpy())
    f2 = i2 != np.rint(odj.djpeg_torch(cr, 80, 'soft')[2].numpy())
    rate = (f1.sum() + f2.sum()) / float(f1.size + f2.size)
    assert rate < 1e-4, rate
    # the five manipulated classes: only the 'jpeg' class has a rounding stage behind it
    bad1 = np.zeros((5 * b, 32, 32), bool)
    bad1[jpeg_class * b:(jpeg_class + 1) * b] = f1.any(axis=(1, 4, 5))
    dc = np.abs(c.numpy().astype(np.float64) - cr.numpy()) * ~np.kron(bad1, np.ones((1, 8, 8), bool))[..., None]
    assert dc.max() <= 1e-4, float(dc.max())
    bad = f2.any(axis=(1, 4, 5)) | bad1
    keep = ~np.kron(bad, np.ones((1, 8, 8), bool))[..., None]
    dC = np.abs(C.numpy().astype(np.float64) - Cr.numpy()) * keep
    assert dC.max() <= 1e-4, (float(dC.max()), int(bad.sum()))
    assert keep.mean() > 0.99                                         # the mask hides a handful of blocks, not the image
    loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), 0.1)
    loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
    assert abs(float(parts['nip']) - parts_ref['nip']) / parts_ref['nip'] < 1e-4
    assert abs(float(parts['ce']) - parts_ref['ce']) < 5e-3, (float(parts['ce']), parts_ref['ce'])
    names = list(ref.fan.keys()) + list(ref.nip.keys())
    got = grads_of(wf.fan)
    got.update(grads_of(wf.nip))
    for k, gr in zip(names, grads):
        a, g = got[k].ravel().astype(np.float64), gr.numpy().ravel()
        cos = float(a @ g / (np.linalg.norm(a) * np.linalg.norm(g) + 1e-300))
        assert cos > 0.98, (k, cos)
    print('full patch size: flip rate {:.1e}, {} of {} blocks masked, max |dC| outside {:.2e}'.format(
        rate, int(bad.sum()), bad.size, float(dC.max())))
    # the throughput mode of the bench line (bf16 MFMA operands, bf16-stored activations, the row-streaming level-1 kernels at
    # their 128-pixel width) against the SAME float64 oracle, judged as BASELINE.json judges it: PSNR, losses, directions
    ops.set_compute('bf16')
    try:
        wb = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=128, device=dev)
        wb.nip.load_state_dict({k: v.numpy() for k, v in ref.nip.items()})      # the weights the oracle was given
        wb.fan.load_state_dict({k: v.numpy() for k, v in ref.fan.items()})
        Yb = wb.run_workflow(raw)[0].numpy()
        lb, pb = wb.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
        gb = grads_of(wb.fan)
        gb.update(grads_of(wb.nip))
    finally:
        ops.set_compute('f32')
    psnr = 10 * np.log10(1.0 / np.mean((Yb.astype(np.float64) - Yr.numpy()) ** 2))
    assert psnr > 45, psnr
    assert abs(float(pb['nip']) - parts_ref['nip']) / parts_ref['nip'] < 1e-2
    assert abs(float(pb['ce']) - parts_ref['ce']) < 3e-2, (float(pb['ce']), parts_ref['ce'])
    worst = 1.0
    for k, gr in zip(names, grads):
        if k.endswith('/kernel') and k.split('/')[0] not in ('constrained', 'conv1'):      # the ill-conditioned front end: see above
            a, g = gb[k].ravel().astype(np.float64), gr.numpy().ravel()
            cos = float(a @ g / (np.linalg.norm(a) * np.linalg.norm(g) + 1e-300))
            worst = min(worst, cos)
            assert cos > 0.9, (k, cos)
    print('   throughput mode vs the oracle: PSNR {:.1f} dB, ce {:.4f} vs {:.4f}, worst kernel-gradient cosine {:.4f}'.format(
        psnr, float(pb['ce']), parts_ref['ce'], worst))


def test_deferred_slab_reductions_give_the_same_gradients(dev, monkeypatch):
    """ops.DEFER_REDUCE (opt-in): the split-K reductions of every side-stream weight gradient of a step in one batched launch per
    side stream (nimg_conv2d_wgrad_bf16_deferred + nimg_reduce_slabs_batch) - the same sums to the bit as one reduction per layer."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(4, 64, 64, seed=77)
    raw = bayer_from_rgb(rgb)
    res = {}
    try:
        ops.set_compute('bf16')
        for defer in (False, True):
            monkeypatch.setattr(ops, 'DEFER_REDUCE', defer)
            wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev)
            for _ in range(2):
                loss, _ = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
            torch.cuda.synchronize()
            res[defer] = (float(loss), {k: v.copy() for k, v in grads_of(wf.nip).items()}, {k: v.copy() for k, v in grads_of(wf.fan).items()},
                          wf.nip.state_dict(), wf.fan.state_dict())
    finally:
        ops.set_compute('f32')
    assert res[False][0] == res[True][0]
    for part in (1, 2, 3, 4):
        for k in res[False][part]:
            assert np.array_equal(res[False][part][k], res[True][part][k]), (part, k)


@pytest.mark.parametrize('raw_patch', [32, 64])
def test_chained_slab_reductions_give_the_same_gradients(dev, monkeypatch, raw_patch):
    """ops.CHAIN_REDUCE: the split-K reduction of a side-stream weight gradient runs in the prologue of the NEXT weight-gradient kernel
    of its stream (nimg_conv2d_wgrad_bf16_chained; common.h reduce_seq restates reduce_slabs' additions per column, in its order),
    the last one of a chain at the join - the same sums to the bit as one reduction launch per layer, UNet, FAN and a codec step."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(4, 2 * raw_patch, 2 * raw_patch, seed=78)
    raw = bayer_from_rgb(rgb)
    res = {}
    try:
        ops.set_compute('bf16')
        for chain in (False, True):
            monkeypatch.setattr(ops, 'CHAIN_REDUCE', chain)
            wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=raw_patch, device=dev)
            for _ in range(2):
                loss, _ = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
            torch.cuda.synchronize()
            dcn = compression.TwitterDCN(patch_size=2 * raw_patch, device=dev)
            dcn.training_step(rgb, 1e-4)
            torch.cuda.synchronize()
            res[chain] = (float(loss), {k: v.copy() for k, v in grads_of(wf.nip).items()}, {k: v.copy() for k, v in grads_of(wf.fan).items()},
                          wf.nip.state_dict(), wf.fan.state_dict(), {k: v.copy() for k, v in grads_of(dcn).items()})
            assert not ops._CHAIN                                   # every chain ended at a join
    finally:
        ops.set_compute('f32')
    assert res[False][0] == res[True][0]
    for part in (1, 2, 3, 4, 5):
        for k in res[False][part]:
            assert np.array_equal(res[False][part][k], res[True][part][k]), (part, k)


def test_twitter_dcn_forward_backward(dev):
    """TwitterDCN-32C (models/compression.py:197-279): reconstruction, hard latent indices (exact), entropy, loss and
    every parameter gradient against the float64 oracle; then the reference's training_step contract."""
    from neural_imaging_amd.models import compression
    dcn = compression.TwitterDCN(patch_size=32, device=dev)
    assert dcn.count_parameters() == 2533293
    x = natural_images(2, 32, 32, seed=13)
    p = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
    for v in p.values():
        v.requires_grad_(True)
    y_ref, ent_ref, lat_ref = onets.dcn_forward(p, to64(x))
    loss_ref = onets.dcn_loss(to64(x), y_ref, ent_ref, 250.0)
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))

    xt = torch.from_numpy(x).to(dev)
    y, ent, ctx = dcn.forward(xt, training=True)
    lat = ctx[0]['latent'].cpu().numpy()
    assert np.array_equal(np.round(lat), np.round(lat_ref.detach().numpy())), 'latent indices differ'
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 1e-4, what='DCN reconstruction')
    assert abs(float(ent.item()) - float(ent_ref)) < 1e-5
    from neural_imaging_amd import ops
    l2, dy = ops.l2_loss(xt, y, grad_scale=1.0)
    dcn.backward(ctx, dy, entropy_coef=250.0)
    total = float(l2.item()) + 250.0 * float(ent.item())
    assert abs(total - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-4
    check_grads(grads_of(dcn), g_ref, list(p.keys()), tol=1e-3)
    # reference surface
    z = dcn.compress(x[0])
    assert z.shape == (1, 4, 4, 32) and dcn.decompress(z).shape == (1, 32, 32, 3)
    out = dcn.training_step(x, learning_rate=1e-4)
    assert set(out.keys()) == {'loss', 'ssim', 'entropy'} and abs(out['loss'] - np.sqrt(2 * float(loss_ref.detach()))) < 1e-2
    yy, ee = dcn.process(x, return_entropy=True)
    assert yy.shape == (2, 32, 32, 3) and np.isfinite(float(ee))


@pytest.mark.parametrize('family', ['INet', 'DNet', 'ClassicISP', 'TwitterDCN'])
def test_model_families_in_throughput_mode(dev, family):
    """Every model family besides the UNet / FAN pair of the bench also runs with bf16 MFMA operands (ops.set_compute('bf16')):
    same weights, same input - the output stays within bf16 rounding of the float32 result and the parameter gradients
    point the same way."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression, pipelines
    rgb = natural_images(4, 64, 64, seed=41)
    raw = bayer_from_rgb(rgb)
    make = {'INet': lambda: pipelines.INet(patch_size=32, random_init=True, device=dev),
            'DNet': lambda: pipelines.DNet(patch_size=32, n_layers=4, n_features=16, device=dev),
            'ClassicISP': lambda: pipelines.ClassicISP(patch_size=32, c_filters=(16, 16), kernel=3, device=dev),
            'TwitterDCN': lambda: compression.TwitterDCN(patch_size=64, n_features=8, device=dev)}[family]
    res = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        try:
            net = make()
            if family == 'TwitterDCN':
                x = torch.from_numpy(rgb).to(dev)
                y, ent, ctx = net.forward(x, training=True)
                l2, dy = ops.l2_loss(x, y, grad_scale=1.0)
                net.backward(ctx, dy, entropy_coef=250.0)
            else:
                y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
                _, dy = net.loss_and_grad(y, torch.from_numpy(rgb).to(dev))
                net.backward(ctx, dy)
            res[mode] = (y.float().cpu().numpy(), net._model.flat_grad.clone())
        finally:
            ops.set_compute('f32')
    assert np.isfinite(res['bf16'][0]).all()
    assert np.abs(res['f32'][0] - res['bf16'][0]).max() < 0.06, np.abs(res['f32'][0] - res['bf16'][0]).max()
    mse = float(np.mean((res['f32'][0] - res['bf16'][0]) ** 2))
    assert 10 * np.log10(1.0 / max(mse, 1e-12)) > 38                              # PSNR between the two modes' outputs
    a, b_ = res['f32'][1].double(), res['bf16'][1].double()
    assert float((a * b_).sum() / (a.norm() * b_.norm())) > 0.97


def test_workflow_bf16_throughput_mode(dev):
    """bf16 MFMA mode of the convolutions: judged like BASELINE.json asks - PSNR of the ISP output against the f32
    path, loss agreement, gradient direction - not on the 1e-4 contract."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(2, 64, 64, seed=8)
    raw = bayer_from_rgb(rgb)
    res = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        try:
            wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev)
            Y = wf.run_workflow(raw)[0].numpy()
            loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=1e-4)
            res[mode] = (Y, float(parts['ce']), float(parts['nip']), grads_of(wf.nip), grads_of(wf.fan))
        finally:
            ops.set_compute('f32')
    psnr = 10 * np.log10(1.0 / np.mean((res['f32'][0] - res['bf16'][0]) ** 2))
    assert psnr > 45, psnr
    assert abs(res['f32'][1] - res['bf16'][1]) < 2e-2 and abs(res['f32'][2] - res['bf16'][2]) / res['f32'][2] < 1e-2
    # dc11 sits behind the 2x2-pixel bottleneck of this tiny patch (and behind the bf16 transposed convolution): its
    # gradient is a sum of few, strongly cancelling terms, hence the looser bound
    for k, lo in (('ec11/kernel', 0.98), ('ec32/kernel', 0.98), ('dc11/kernel', 0.9), ('dc42/kernel', 0.98)):
        a, b = res['f32'][3][k].ravel(), res['bf16'][3][k].ravel()
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > lo, k
    for k in ('conv2/kernel', 'conv4/kernel', 'dense/kernel'):
        a, b = res['f32'][4][k].ravel(), res['bf16'][4][k].ravel()
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.95, k


def test_workflow_with_learned_codec_joint_training(dev):
    """Config 5 shape of the channel: UNet -> manipulations -> TwitterDCN -> FAN with trainable = {nip, dcn}
    (workflows/manipulation_classification.py:267-277: loss = CE + ln * nip + lc * (l2 + 250 H))."""
    from neural_imaging_amd.models import compression
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    manips = ['sharpen:1', 'resample:50', 'gaussian:0.83']
    dcn = compression.TwitterDCN(patch_size=64, device=dev)
    dist = {'downsampling': 'none', 'compression': 'dcn', 'compression_params': {'model': dcn}}
    wf = ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip', 'dcn'},
                                    raw_patch_size=32, device=dev)
    ref = owf.Workflow(manipulations=manips, codec='dcn', trainable=('nip', 'dcn'))
    _sync_oracle(wf, ref)
    ref.dcn = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
    rgb = natural_images(2, 64, 64, seed=8)
    raw = bayer_from_rgb(rgb)
    loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), 0.1, 0.01)
    loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, lambda_dcn=0.01, learning_rate=1e-4)
    assert abs(float(parts['ce']) - parts_ref['ce']) < 2e-3
    assert abs(float(parts['nip']) - parts_ref['nip']) / parts_ref['nip'] < 1e-3
    assert abs(parts['dcn'] - parts_ref['dcn']) / parts_ref['dcn'] < 1e-3
    assert abs(float(loss) - float(loss_ref.detach())) / float(loss_ref.detach()) < 1e-3
    names = list(ref.fan.keys()) + list(ref.nip.keys()) + list(ref.dcn.keys())
    got = grads_of(wf.fan)
    got.update(grads_of(wf.nip))
    got.update(grads_of(dcn))
    worst = 1.0
    for k, gr in zip(names, grads):
        a, b = got[k].ravel().astype(np.float64), gr.numpy().ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
        worst = min(worst, cos)
        assert cos > 0.99, (k, cos)
    assert wf.is_trainable('dcn') and 'TwitterDCN' in wf.summary()


def test_learned_codec_channel_at_the_full_patch_size(dev):
    """BASELINE.json configs[4] at its real patch size - UNet (RAW 128 x 128) -> manipulations -> TwitterDCN-32C on 256 x 256 -> FAN,
    trainable = {nip, dcn} (workflows/manipulation_classification.py:267-277) - at B = 1 raw patch (five codec / FAN images), which
    the float64 oracle finishes in seconds: losses and gradient directions in parity mode, and the throughput mode of the bench line
    against the same oracle."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    rgb = natural_images(1, 256, 256, seed=9)
    raw = bayer_from_rgb(rgb)
    ref = owf.Workflow(codec='dcn', trainable=('nip', 'dcn'))
    res = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        try:
            dcn = compression.TwitterDCN(patch_size=256, device=dev)
            dist = {'downsampling': 'none', 'compression': 'dcn', 'compression_params': {'model': dcn}}
            wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip', 'dcn'}, raw_patch_size=128, device=dev)
            if mode == 'f32':
                _sync_oracle(wf, ref)
                ref.dcn = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
                loss_ref, parts_ref, params, grads, _ = ref.loss_and_grads(to64(raw), to64(rgb), 0.1, 0.01)
            else:
                wf.nip.load_state_dict({k: v.numpy() for k, v in ref.nip.items()})
                wf.fan.load_state_dict({k: v.numpy() for k, v in ref.fan.items()})
                dcn.load_state_dict({k: v.numpy() for k, v in ref.dcn.items()})
            loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, lambda_dcn=0.01, learning_rate=1e-4)
            got = grads_of(wf.fan)
            got.update(grads_of(wf.nip))
            got.update(grads_of(dcn))
            res[mode] = (float(loss), {k: float(v) for k, v in parts.items()}, got)
        finally:
            ops.set_compute('f32')
    names = list(ref.fan.keys()) + list(ref.nip.keys()) + list(ref.dcn.keys())
    lref = float(loss_ref.detach())
    deep = ('ec41', 'ec42', 'ec51', 'ec52', 'dct1', 'dct2', 'dc11', 'dc12')       # UNet levels 4 - 5 and the way back up
    for mode, tol_ce, tol_rel, lo in (('f32', 2e-3, 1e-3, 0.98), ('bf16', 3e-2, 3e-2, 0.9)):
        loss, parts, got = res[mode]
        assert abs(parts['ce'] - parts_ref['ce']) < tol_ce, (mode, parts['ce'], parts_ref['ce'])
        assert abs(parts['nip'] - parts_ref['nip']) / parts_ref['nip'] < tol_rel, mode
        assert abs(parts['dcn'] - parts_ref['dcn']) / parts_ref['dcn'] < tol_rel, mode
        assert abs(loss - lref) / lref < tol_rel, (mode, loss, lref)
        cosines = {}
        for k, gr in zip(names, grads):
            a, b = got[k].ravel().astype(np.float64), gr.numpy().ravel()
            cosines[k] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
        if mode == 'f32':
            assert min(cosines.values()) > lo, sorted((c, k) for k, c in cosines.items())[:8]
        else:
            # throughput mode: the FAN's and the codec's kernels point the oracle's way to 3 digits.  The UNet's gradient arrives
            # through the (randomly initialised) codec, whose l2 term dominates this loss: bf16 WEIGHTS are a spatially coherent
            # ~0.4 % perturbation of that linear map, i.e. a low-frequency error on a gradient image whose low-frequency content
            # nearly cancels - the deeper the level the less of the oracle's direction survives (measured: level 1 0.997, level 2
            # 0.96, level 3 0.92 - 0.94, level 4 0.66 - 0.69, level 5 0.36 - 0.45; the same UNet under its own loss: 0.9995 at every
            # level, test_unet_at_the_full_patch_size).  The yardstick is tests/test_oracle.py::
            # test_codec_channel_gradient_conditioning_under_bf16_weights: the float64 oracle ITSELF with its weights rounded to bf16
            # keeps 0.09 - 0.30 at levels 4 - 5.  Levels 1 - 3 are held to the floor; 4 - 5 only to the right half-space.
            kern = {k: c for k, c in cosines.items() if k.endswith('/kernel') and k.split('/')[0] not in ('constrained', 'conv1')}
            for k, c in kern.items():
                own = k in ref.nip
                floor = 0.25 if (own and k.split('/')[0] in deep) else (lo if own else 0.99)
                assert c > floor, (k, c, sorted((c2, k2) for k2, c2 in kern.items())[:12])
        worst = min((c, k) for k, c in cosines.items() if mode == 'f32' or k.endswith('/kernel'))
        print('learned-codec channel at full patch size, {}: loss {:.5f} vs {:.5f}, worst gradient cosine {:.4f} ({})'.format(
            mode, loss, lref, *worst))


def test_channel_learns_and_compute_modes_agree(dev):
    """The reference's own acceptance test (config/tests/framework.json 'train-manipulation': INet, sharpen + gaussian,
    --train nip, validation accuracy > 0.50) on synthetic patches, in both compute modes: the classifier reaches the
    floor within 100 steps, and the bf16 throughput mode follows the float32 loss trajectory (accuracy / PSNR parity)."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(64, 64, 64, seed=100)
    raw = bayer_from_rgb(rgb)
    vrgb = natural_images(24, 64, 64, seed=200)
    vraw = bayer_from_rgb(vrgb)
    labels = np.repeat(np.arange(3), 24)
    traj = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        try:
            wf = ManipulationClassification('INet', manipulations=['sharpen:1', 'gaussian:1'], distribution=dist,
                                            trainable={'nip'}, raw_patch_size=32, device=dev)
            rng = np.random.RandomState(0)
            losses = []
            for step in range(100):
                idx = rng.choice(64, 8, replace=False)
                loss, parts = wf.training_step(raw[idx], rgb[idx], lambda_nip=0.1, learning_rate=1e-4)
                losses.append(float(loss))
            acc = float(np.mean(np.asarray(wf.run_workflow_to_decisions(vraw)) == labels))
            y = wf.nip.process(vraw).numpy()
            psnr = float(np.mean(10 * np.log10(1.0 / np.mean((y - vrgb) ** 2, axis=(1, 2, 3)))))
            traj[mode] = (np.array(losses), acc, psnr)
        finally:
            ops.set_compute('f32')
    for mode in traj:
        assert traj[mode][1] > 0.5, (mode, traj[mode][1])                   # the reference's floor
        assert traj[mode][0][-1] < 0.8 * traj[mode][0][0]
    assert abs(traj['f32'][1] - traj['bf16'][1]) <= 0.05                     # accuracy parity
    assert abs(traj['f32'][2] - traj['bf16'][2]) < 0.2                       # PSNR parity (dB)
    assert np.max(np.abs(traj['f32'][0] - traj['bf16'][0]) / traj['f32'][0]) < 2e-2


def test_full_size_step_properties(dev):
    """BASELINE.json configs[3] at its full size (64 RAW patches of 128 x 128 -> 320 FAN images of 256 x 256, UNet -> five
    manipulation classes -> dJPEG QF 80 -> FAN), where the float64 oracle would take minutes: size-independent properties.
      * the analytic FAN gradient of the float32 step equals the directional derivative of the loss measured by central
        differences along a random direction (the FAN sees a fixed input when only its weights move, so the rounding stages
        upstream do not enter);
      * the bf16 throughput mode reproduces the float32 loss and gradients of the same step;
      * the class order / label layout of run_workflow at this size: B contiguous rows per class."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    b = 64
    rgb = natural_images(b, 256, 256, seed=77)
    raw = bayer_from_rgb(rgb)
    res = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        try:
            wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip', 'fan'}, raw_patch_size=128, device=dev,
                                            manipulations=['sharpen', 'resample', 'gaussian', 'jpeg'])
            loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=0.0)     # lr 0: gradients, no update
            res[mode] = (float(loss), wf.fan._model.flat_grad.clone(), wf.nip._model.flat_grad.clone())
            if mode == 'f32':
                assert wf.n_classes == 5
                Y, c, C, _, probs = wf.run_workflow(raw)
                assert Y.shape == (b, 256, 256, 3) and C.shape == (5 * b, 256, 256, 3) and probs.shape == (5 * b, 5)
                assert np.array_equal(c.numpy()[:b], Y.numpy())                         # class 0 = the native images
                # directional derivative of the loss w.r.t. the FAN parameters
                gfan = res[mode][1]
                w0 = wf.fan._model.flat.clone()
                gen = torch.Generator(device='cpu').manual_seed(5)
                d = torch.randn(w0.shape, generator=gen).to(dev) * (w0 != 0)              # keep the alignment gaps at zero
                d = d / d.norm() * 0.02 * w0.norm()
                fd = []
                for sgn in (1.0, -1.0):
                    wf.fan._model.flat.copy_(w0 + sgn * d)
                    l, _ = wf.training_step(raw, rgb, lambda_nip=0.1, learning_rate=0.0)
                    fd.append(float(l))
                wf.fan._model.flat.copy_(w0)
                num = (fd[0] - fd[1]) / 2.0
                ana = float((gfan.double() * d.double()).sum())
                assert abs(num) > 1e-4 and abs(num - ana) <= 0.03 * abs(num), (num, ana, fd, res[mode][0])
        finally:
            ops.set_compute('f32')
    lf, lb = res['f32'][0], res['bf16'][0]
    assert np.isfinite(lf) and abs(lf - lb) / lf < 1e-2, (lf, lb)
    cos = lambda a, c: float((a.double() * c.double()).sum() / (a.double().norm() * c.double().norm()))
    assert cos(res['f32'][1], res['bf16'][1]) > 0.995
    assert cos(res['f32'][2], res['bf16'][2]) > 0.97


def test_training_harness_outputs(dev, tmp_path):
    """H1 (training/manipulation.py:36-335): epoch loop, lr decay, validation cadence, training.json keys, checkpoints
    and the 'directory exists => skip' idempotence, on a synthetic dataset."""
    import json
    import os
    from neural_imaging_amd.training import manipulation as tm
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'pool:2', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('UNet', manipulations=['sharpen:1', 'gaussian:1'], distribution=dist,
                                    trainable={'nip'}, raw_patch_size=16, device=dev)
    data = tm.SyntheticDataset(8, 4, patch_size=16)
    spec = {'camera_name': 'synthetic', 'use_pretrained_nip': False, 'patch_size': 16, 'batch_size': 4,
            'n_epochs': 3, 'validation_schedule': 2, 'lambda_nip': 0.1, 'lambda_dcn': 0, 'run_number': 0,
            'learning_rate': 1e-4, 'augment': False}
    mdir = tm.train_manipulation_nip(wf, spec, data, {'root': str(tmp_path)})
    run_dir = os.path.dirname(mdir)
    assert run_dir.endswith(os.path.join('synthetic', 'UNet', 'ln-0.1000', 'fixed-codec', '000'))
    prog = json.load(open(os.path.join(run_dir, 'training.json')))
    assert set(prog.keys()) >= {'summary', 'distribution', 'manipulations', 'nip', 'forensics', 'codec'}
    assert prog['manipulations'] == ['native', 'sharpen:1.0', 'gaussian:1.0']
    sm = prog['summary']                                  # the reference's keys and number formats (training/manipulation.py:159-190)
    assert sm['NIP Regularization'] == '0.100' and sm['Learning rate'] == '0.000100' and sm['# Epochs'] == '3'
    assert sm['Learning rate decay rate'] == '0.900' and sm['Batch shape'] == '(1, 16, 16, 4)' and sm['NIP loss'] == 'L2'
    assert list(sm.keys())[:3] == ['Problem', 'Dataset', 'Camera name'] and 'FAN input patch' in sm and len(sm) == 28
    assert len(prog['forensics']['performance']['accuracy']['validation']) == 3        # epochs 0 and 2 + the final pass
    conf = np.asarray(prog['forensics']['performance']['confusion'])
    assert conf.shape == (3, 3) and np.allclose(conf.sum(axis=1), 1.0)                 # rows = true class, normalised
    assert len(prog['nip']['performance']['loss']['training']) == 3
    assert len(prog['nip']['performance']['psnr']['validation']) >= 1
    assert os.path.isfile(os.path.join(mdir, 'fan', 'fan.h5')) and os.path.isfile(os.path.join(mdir, 'unet', 'unet.h5'))
    before = os.path.getmtime(os.path.join(run_dir, 'training.json'))
    assert tm.train_manipulation_nip(wf, spec, data, {'root': str(tmp_path)}) == mdir   # exists => skipped
    assert os.path.getmtime(os.path.join(run_dir, 'training.json')) == before
    with pytest.raises(RuntimeError):
        tm.train_manipulation_nip(wf, {'n_epochs': 1}, data, {'root': str(tmp_path)})   # missing camera_name
    # a trainable differentiable-JPEG codec (quantisation tables as weights): validate_jpeg inside the loop (:241-242), 'lc-' directory
    from neural_imaging_amd.training import validation as tv
    dist_t = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 70, 'codec': 'sin', 'trainable': True}}
    wt = ManipulationClassification('UNet', manipulations=['gaussian:1'], distribution=dist_t, trainable={'nip', 'dcn'},
                                    raw_patch_size=16, device=dev)
    spec_t = dict(spec, lambda_dcn=10.0, run_number=1)
    mdir_t = tm.train_manipulation_nip(wt, spec_t, data, {'root': str(tmp_path)})
    assert os.path.dirname(mdir_t).endswith(os.path.join('synthetic', 'UNet', 'ln-0.1000', 'lc-10.0000', '001'))
    codec_perf = json.load(open(os.path.join(os.path.dirname(mdir_t), 'training.json')))['codec']['performance']
    assert len(codec_perf['psnr']['validation']) == 2 and len(codec_perf['ssim']['validation']) == 2      # epochs 0 and 2
    assert all(30 < v < 60 for v in codec_perf['psnr']['validation']) and all(np.isnan(v) for v in codec_perf['entropy']['validation'])
    vj = tv.validate_jpeg(wt.codec, data, batch_size=2)
    assert set(vj) == {'psnr', 'ssim', 'entropy'} and 0.8 < vj['ssim'] <= 1.0 and np.isnan(vj['entropy'])
    with pytest.raises(ValueError):
        tv.validate_jpeg(wt.fan, data)


def _tiny_dataset(n_train, n_val, h, w, val_patch, seed):
    """helpers/dataset.Dataset over synthetic full-resolution images: RGB uint8 + the uint16 RAW stacks of its Bayer mosaic."""
    from neural_imaging_amd.helpers import dataset
    rgb = (natural_images(n_train + n_val, h, w, seed=seed) * 255).round().astype(np.uint8)
    raw = (bayer_from_rgb(rgb.astype(np.float32) / 255) * 65535).round().astype(np.uint16)
    return dataset.Dataset.from_arrays({'x': raw[:n_train], 'y': rgb[:n_train]},
                                       {'x': raw[n_train:, :val_patch // 2, :val_patch // 2],
                                        'y': rgb[n_train:, :val_patch, :val_patch]})


@pytest.mark.parametrize('feed', ['host', 'device'])
def test_nip_pretraining_harness(dev, tmp_path, feed):
    """H2 (training/pipeline.py:105-256): epoch loop over Dataset batches, validation cadence, progress.json, checkpoint,
    resume and the 'directory exists => skip' rule - fed from the host Dataset and from the HBM-resident DeviceDataset."""
    import json
    import os
    from neural_imaging_amd.helpers import dataset
    from neural_imaging_amd.models import pipelines
    from neural_imaging_amd.training import pipeline as tp
    data = _tiny_dataset(8, 4, 96, 128, 64, seed=5)
    if feed == 'device':
        data = dataset.DeviceDataset(data, device=dev, seed=3)
    np.random.seed(1)
    net = pipelines.UNet(patch_size=16, device=dev)
    out = tp.train_nip_model(net, 'synthetic', n_epochs=6, lr_schedule={0: 1e-3, 3: 5e-4}, validation_schedule=2,
                             patch_size=32, batch_size=4, data=data, out_directory_root=str(tmp_path), discard=None)
    assert out.endswith(os.path.join('synthetic', net.model_code, 'unet'))
    prog = json.load(open(os.path.join(out, 'progress.json')))
    assert set(prog.keys()) == {'performance', 'summary', 'args'} and prog['summary']['Epoch'] == 5
    perf = prog['performance']
    assert len(perf['loss']['training']) == 6 and len(perf['loss']['validation']) == 3      # epochs 0, 2, 4
    assert len(perf['psnr']['validation']) == 3 and len(perf['ssim']['validation']) == 3
    assert perf['loss']['training'][-1] < 0.9 * perf['loss']['training'][0], perf['loss']['training']
    assert perf['psnr']['validation'][-1] > perf['psnr']['validation'][0]
    assert os.path.isfile(os.path.join(out, 'unet.h5'))
    assert tp.train_nip_model(net, 'synthetic', n_epochs=6, patch_size=32, batch_size=4, data=data,
                              out_directory_root=str(tmp_path)) == out                       # exists => skipped
    net2 = pipelines.UNet(patch_size=16, device=dev, seed=99)
    tp.train_nip_model(net2, 'synthetic', n_epochs=8, validation_schedule=2, patch_size=32, batch_size=4, data=data,
                       out_directory_root=str(tmp_path), resume=True, discard=None)
    prog2 = json.load(open(os.path.join(out, 'progress.json')))
    assert prog2['summary']['Start epoch'] == 6 and len(prog2['performance']['loss']['training']) == 8
    assert prog2['performance']['loss']['training'][6] < perf['loss']['training'][0]            # resumed, not restarted
    # resuming a FINISHED run trains nothing and leaves progress file and checkpoint alone (no epoch label that grows per call)
    stamp = os.path.getmtime(os.path.join(out, 'unet.h5'))
    tp.train_nip_model(net2, 'synthetic', n_epochs=8, validation_schedule=2, patch_size=32, batch_size=4, data=data,
                       out_directory_root=str(tmp_path), resume=True, discard=None)
    prog3 = json.load(open(os.path.join(out, 'progress.json')))
    assert prog3['summary']['Epoch'] == prog2['summary']['Epoch'] == 7 and os.path.getmtime(os.path.join(out, 'unet.h5')) == stamp
    assert len(prog3['performance']['loss']['training']) == 8
    with pytest.raises(ValueError):
        tp.train_nip_model(net, 'synthetic', patch_size=32, batch_size=16, data=data, out_directory_root=str(tmp_path))
    with pytest.raises(FileNotFoundError):              # resume without a progress.json to resume from (training/pipeline.py:141-142)
        tp.train_nip_model(net, 'nowhere', n_epochs=1, patch_size=32, batch_size=4, data=data, resume=True,
                           out_directory_root=str(tmp_path))
    # the bare loop (:259-302): steps only, from the Dataset and from an iterable of batches
    net3 = pipelines.UNet(patch_size=16, device=dev)
    bx, by = data.next_training_batch(0, 4, 32, discard=None)
    l0 = float(net3.training_step(bx, by, 0.0))
    assert tp.train_nip_bare(net3, 'synthetic', n_epochs=3, patch_size=32, batch_size=4, data=data, discard=None,
                             out_directory_root=str(tmp_path)).endswith(os.path.join('synthetic', net3.model_code, 'unet'))
    tp.train_nip_bare(net3, 'synthetic', n_epochs=2, data=[(bx, by)] * 2)
    assert float(net3.training_step(bx, by, 0.0)) < l0


@pytest.mark.parametrize('feed', ['host', 'device'])
def test_dcn_pretraining_harness(dev, tmp_path, feed):
    """H2 (training/compression.py:123-309): flips, training_step per batch, lr reduction schedule, validation metrics,
    progress.json and checkpoint, from both dataset kinds."""
    import json
    import os
    from neural_imaging_amd.helpers import dataset
    from neural_imaging_amd.models import compression
    from neural_imaging_amd.training import compression as tc
    host = _tiny_dataset(8, 4, 96, 128, 64, seed=6)
    data = dataset.Dataset.from_arrays({'y': host.data['training']['y']}, {'y': host.data['validation']['y']})
    if feed == 'device':
        data = dataset.DeviceDataset(data, device=dev, seed=4)
    np.random.seed(2)
    dcn = compression.TwitterDCN(patch_size=64, n_features=8, device=dev)
    spec = {'n_epochs': 13, 'batch_size': 4, 'patch_size': 64, 'learning_rate': 5e-4, 'validation_schedule': 4,
            'learning_rate_reduction_schedule': 8}
    out = tc.train_dcn(dcn, spec, data, directory=str(tmp_path))
    prog = json.load(open(os.path.join(out, 'progress.json')))
    perf = prog['performance']
    assert len(perf['loss']['training']) == 13 and len(perf['entropy']['training']) == 13    # written at epochs 0, 4, 8, 12
    assert all(len(perf[k]['validation']) == 4 for k in ('ssim', 'psnr', 'entropy', 'loss'))
    tl = perf['loss']['training']
    # 26 Adam steps on 8 tiny images: the rate-distortion sum (l2 + 250 H, ~48) wanders by +-1.5 from batch to batch, so the
    # harness is held to a bounded loss and to the validation PSNR it logs, not to a monotone trend
    assert np.isfinite(tl).all() and max(tl) < 1.5 * tl[0], tl
    assert perf['psnr']['validation'][-1] > perf['psnr']['validation'][0]
    assert os.path.isfile(os.path.join(out, 'twitterdcn.h5'))
    assert tc.train_dcn(dcn, spec, data, directory=str(tmp_path)) == out                    # exists => skipped
    restored = compression.TwitterDCN.restore(out, patch_size=64, device=dev)
    for a, b_ in zip(dcn.parameters, restored.parameters):
        assert torch.equal(a, b_)



@pytest.mark.parametrize('codec', ['jpeg', 'dcn'])
def test_data_parallel_step_equals_global_batch(dev, codec):
    """two ranks (gloo, both on cuda:0) each run training_step on one half of a global batch; the all-reduced
    gradients / world, the losses and the parameters after the shared Adam step equal the single-rank step on the whole
    batch (float32 mode).  Covers the mean-loss 1/world scaling, the un-divided l2_loss of the learned codec, its batch-global
    entropy histogram (all-reduced in the forward pass), the decoder / encoder gradient buckets and the NaN flag."""
    import socket
    import torch.multiprocessing as mp
    from dp_worker import channel_state, dp_step_worker, make_channel
    from neural_imaging_amd import ops
    ops.set_compute('f32')
    rgb = natural_images(4, 64, 64, seed=21)
    raw = bayer_from_rgb(rgb)
    wf, kw = make_channel(codec, dev)
    loss, parts = wf.training_step(raw, rgb, learning_rate=1e-4, **kw)
    wf.check_nan()
    g_ref, p_ref = channel_state(wf)
    ref = (float(parts['ce']), float(parts['nip']), float(parts['dcn']))
    del wf
    res, procs = collect_from_workers(
        lambda ctx, port, q: [ctx.Process(target=dp_step_worker, args=(r, 2, port, q, codec, raw, rgb)) for r in range(2)], 2, 600)
    res = sorted(res, key=lambda r: r[0])
    assert all(p.exitcode == 0 for p in procs)
    # CE and the NIP loss are means over the rank's shard: their rank average is the global-batch value
    assert abs(0.5 * (res[0][1] + res[1][1]) - ref[0]) < 1e-5 * max(1.0, abs(ref[0]))
    assert abs(0.5 * (res[0][2] + res[1][2]) - ref[1]) < 1e-5 * max(1.0, abs(ref[1]))
    for k in range(len(g_ref)):
        scale = np.abs(g_ref[k]).max()
        for r in res:
            assert np.abs(r[4][k] / 2.0 - g_ref[k]).max() <= 1e-5 * scale + 1e-12, ('gradients', k, r[0])
            assert np.abs(r[5][k] - p_ref[k]).max() <= 2e-6, ('parameters after Adam', k, r[0])
        assert np.array_equal(res[0][4][k], res[1][4][k])          # both ranks hold the same reduced buffer


@pytest.mark.gpu
@pytest.mark.parametrize('codec,mode', [('jpeg', 'f32'), ('jpeg', 'bf16'), ('dcn', 'f32')])
def test_data_parallel_step_nccl_world1(dev, codec, mode):
    """the data-parallel step through RCCL on the one GPU a test box has.  A one-rank `nccl` process group
    with the collectives forced (parallel.force_collectives) sends the three gradient buckets (FAN, UNet decoder, UNet encoder
    [, codec]), the NaN flag and the codec's entropy histogram through ProcessGroupNCCL - its stream / event hand-off with the
    library's raw-stream launches and the side-stream joins is what this proves.  A sum over one rank is the identity, so three
    training steps must leave losses, gradients and post-Adam parameters EQUAL to the plain step's (bit for bit with dJPEG; the
    learned codec's float64 histogram uses LDS atomics whose order is not fixed: 1e-6)."""
    import torch.multiprocessing as mp
    from dp_worker import dp_nccl_world1_worker
    rgb = natural_images(4, 64, 64, seed=22)
    raw = bayer_from_rgb(rgb)
    (out,), procs = collect_from_workers(
        lambda ctx, port, q: [ctx.Process(target=dp_nccl_world1_worker, args=(q, codec, raw, rgb, mode))], 1, 600)
    assert procs[0].exitcode == 0
    (l0, g0, p0), (l1, g1, p1) = out['plain'], out['nccl']
    exact = codec == 'jpeg'
    for a, b in zip(l0, l1):
        for u, v in zip(a, b):
            assert (u == v or (np.isnan(u) and np.isnan(v))) if exact else abs(u - v) <= 1e-6 * max(1.0, abs(u))
    for k in range(len(g0)):
        if exact:
            assert np.array_equal(g0[k], g1[k]), ('gradients', k)
            assert np.array_equal(p0[k], p1[k]), ('parameters after 3 Adam steps', k)
        else:
            assert np.abs(g0[k] - g1[k]).max() <= 1e-6 * np.abs(g0[k]).max() + 1e-12, ('gradients', k)
            assert np.abs(p0[k] - p1[k]).max() <= 1e-6, ('parameters after 3 Adam steps', k)


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_late_fan_weight_gradients_are_the_same_gradients(dev, mode, monkeypatch):
    """The workflow issues the FAN's weight gradients behind its input-gradient chain (forensics.LATE_PARAMS, from
    LATE_MIN_IMAGES FAN images): same kernels on the same operands - gradients, NaN flag and the Adam step are bit-identical to
    the order that launches them beside the input gradients."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import forensics
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    ops.set_compute(mode)
    try:
        dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
        rgb = natural_images(2, 64, 64, seed=33)
        raw = bayer_from_rgb(rgb)
        bx, by = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)
        monkeypatch.setattr(forensics, 'LATE_MIN_IMAGES', 0)
        state = []
        for late in (True, False):
            monkeypatch.setattr(forensics, 'LATE_PARAMS', late)
            wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev,
                                            nan_check='deferred')
            for _ in range(2):
                loss, _ = wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-3)
            wf.check_nan()
            state.append((float(loss), wf.fan._model.flat_grad.clone(), wf.nip._model.flat_grad.clone(),
                          wf.fan._model.flat.clone(), wf.nip._model.flat.clone()))
        assert state[0][0] == state[1][0]
        for a, b in zip(state[0][1:], state[1][1:]):
            assert torch.equal(a, b)
    finally:
        ops.set_compute('f32')


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_pipelined_fan_update_gives_the_same_weights(dev, mode):
    """ManipulationClassification(pipeline_fan_update=True): the FAN's weight gradients and Adam update of step n are issued at the
    start of step n + 1, beside its UNet forward.  Same kernels, same operands, same update order: after finish_pending() the
    losses of every step, both gradient buffers and both models' weights are bit-identical to the unpipelined run; between steps
    the FAN is one update behind until finish_pending() (which run_workflow calls)."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    ops.set_compute(mode)
    try:
        dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
        rgb = natural_images(4, 64, 64, seed=41)
        raw = bayer_from_rgb(rgb)
        batches = [(torch.from_numpy(raw[i:i + 2]).to(dev), torch.from_numpy(rgb[i:i + 2]).to(dev)) for i in (0, 2)]
        state = []
        for piped in (True, False):
            wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev,
                                            nan_check='deferred', pipeline_fan_update=piped)
            losses = []
            for k in range(5):
                loss, _ = wf.training_step(*batches[k % 2], lambda_nip=0.1, learning_rate=1e-3)
                losses.append(loss)
            if piped:
                assert wf._pending_fan is not None
                behind = wf.fan._model.flat.clone()
            probs = wf.run_workflow(batches[0][0])[-1].numpy()            # finishes the pending update first
            assert wf._pending_fan is None
            if piped:
                assert not torch.equal(behind, wf.fan._model.flat)
            wf.check_nan()
            state.append(([float(v) for v in losses], probs, wf.fan._model.flat_grad.clone(), wf.nip._model.flat_grad.clone(),
                          wf.fan._model.flat.clone(), wf.nip._model.flat.clone(), wf.fan._model.m.clone(), wf.fan._model.v.clone()))
        assert state[0][0] == state[1][0]
        assert np.array_equal(state[0][1], state[1][1])
        for a, b in zip(state[0][2:], state[1][2:]):
            assert torch.equal(a, b)
    finally:
        ops.set_compute('f32')


def test_fused_head_gradient_gives_the_same_step(dev, monkeypatch):
    """The workflow's one-pass hand-over to the UNet backward (UNet.head_gradient) against the three separate passes: same
    loss, bit-identical gradients and parameters after two steps."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(2, 64, 64, seed=37)
    raw = bayer_from_rgb(rgb)
    bx, by = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)
    state = []
    for fused in (True, False):
        monkeypatch.setattr(ops, 'FUSED_HEAD_GRAD', fused)
        wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev,
                                        nan_check='deferred')
        for _ in range(2):
            loss, comp = wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-3)
        wf.check_nan()
        state.append((float(comp['nip']), wf.nip._model.flat_grad.clone(), wf.fan._model.flat_grad.clone(),
                      wf.nip._model.flat.clone()))
    assert abs(state[0][0] - state[1][0]) <= 1e-6 * abs(state[1][0])
    for a, b in zip(state[0][1:], state[1][1:]):
        assert torch.equal(a, b)


def test_captured_step_replays_the_eager_step(dev):
    """graphs.CapturedStep: the hipGraph replay of the training step walks the same weights trajectory as eager launches
    (including Keras Adam's per-step bias correction, which the replay reads from device memory)."""
    from neural_imaging_amd import graphs, ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    ops.set_compute('f32')
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(2, 64, 64, seed=31)
    raw = bayer_from_rgb(rgb)
    bx, by = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)
    # Adam's first steps move every weight by ~lr whatever the gradient's size, so a 1-ulp difference anywhere grows
    # chaotically over steps: compare ONE replay against ONE eager step from identical states (2 eager steps each)
    flows = []
    for _ in range(2):
        wf = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev,
                                        nan_check='deferred')
        for _ in range(2):
            wf.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-3)
        flows.append(wf)
    eager, captured = flows
    assert torch.equal(eager.nip._model.flat, captured.nip._model.flat)              # the eager step is deterministic
    runner = graphs.CapturedStep(captured, bx, by, learning_rate=1e-3, lambda_nip=0.1, warmup=1)
    eager.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-3)                  # both: 3 eager steps so far
    assert torch.equal(eager.fan._model.flat, captured.fan._model.flat) and captured._step == 3
    loss_e, _ = eager.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-3)
    loss_c, _ = runner.step()
    assert captured._step == 4
    eager.check_nan(), captured.check_nan()
    assert abs(float(loss_e) - float(loss_c)) <= 1e-6 * abs(float(loss_e))
    for a, b in ((eager.fan, captured.fan), (eager.nip, captured.nip)):
        assert torch.equal(a._model.flat_grad, b._model.flat_grad)                   # same kernels, same order, same bits
        assert (a._model.flat - b._model.flat).abs().max().item() <= 1e-7            # Keras Adam with the device-side rate
    runner.step()
    assert captured._step == 5 and abs(float(runner._rate_dev.item()) - ops.adam_lr_t(1e-3, 5)) < 1e-10


def test_run_ahead_bounds_the_steps_in_flight(dev):
    """graphs.RunAhead(depth): after every call at most `depth` recorded steps are still unfinished on the host side."""
    from neural_imaging_amd import graphs
    pace = graphs.RunAhead(2)
    x = torch.zeros((1 << 22,), device=dev)
    for k in range(6):
        x.add_(1.0)
        pace()
        assert len(pace.events) <= 2
        if k >= 2:                                  # the step issued two calls ago has been waited for
            assert float(x[0].item()) >= k - 1
    torch.cuda.synchronize()
    assert float(x[0].item()) == 6.0


@pytest.mark.parametrize('family', ['nip', 'dcn'])
def test_captured_model_step_replays_the_eager_step(dev, family):
    """graphs.CapturedModelStep: NIPModel.training_step / DCN.training_step (configs 2 / 3) replayed from a HIP graph walk the
    same weights as eager launches - same kernels in the same order (gradients bit-equal), Keras Adam's bias-corrected rate read
    from device memory."""
    from neural_imaging_amd import graphs, ops
    from neural_imaging_amd.models import compression, pipelines
    ops.set_compute('f32')
    rgb = natural_images(2, 64, 64, seed=33)
    if family == 'nip':
        make = lambda: pipelines.UNet(patch_size=32, device=dev)
        batch = (torch.from_numpy(bayer_from_rgb(rgb)).to(dev), torch.from_numpy(rgb).to(dev))
    else:
        make = lambda: compression.TwitterDCN(patch_size=64, device=dev)
        batch = (torch.from_numpy(rgb).to(dev),)
    eager, captured = make(), make()
    kw = dict(sync=False) if family == 'dcn' else {}
    for m in (eager, captured):
        for _ in range(2):
            m.training_step(*batch, learning_rate=1e-3, **kw)
    assert torch.equal(eager._model.flat, captured._model.flat)
    runner = graphs.CapturedModelStep(captured, *batch, learning_rate=1e-3, warmup=1)
    eager.training_step(*batch, learning_rate=1e-3, **kw)                    # both: 3 eager steps so far
    assert captured._model.step == 3 and (eager._model.flat - captured._model.flat).abs().max().item() <= 1e-7
    eager.training_step(*batch, learning_rate=1e-3, **kw)
    runner.step()
    assert captured._model.step == 4
    if family == 'nip':
        assert torch.equal(eager._model.flat_grad, captured._model.flat_grad)
    else:       # the codec's float64 histogram uses LDS atomics whose order is not fixed
        assert (eager._model.flat_grad - captured._model.flat_grad).abs().max().item() <= 1e-5 * eager._model.flat_grad.abs().max().item()
    assert (eager._model.flat - captured._model.flat).abs().max().item() <= 2e-6
    runner.step()
    assert captured._model.step == 5 and abs(float(runner._rate_dev.item()) - ops.adam_lr_t(1e-3, 5)) < 1e-10


def test_captured_step_survives_workspace_regrowth(dev):
    """the graph holds raw addresses of buffers that live outside its pool (ops.Workspace scratch, the
    manipulations' filter-tap cache).  A LARGER eager step after the capture re-grows the workspaces and nine other strengths
    evict the cached tables; CapturedStep pins what it captured on, so the replay still equals the eager step."""
    from neural_imaging_amd import graphs, ops
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    ops.set_compute('f32')
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    rgb = natural_images(2, 64, 64, seed=33)
    raw = bayer_from_rgb(rgb)
    bx, by = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)
    flows = [ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=32, device=dev,
                                        nan_check='deferred') for _ in range(2)]
    eager, captured = flows
    runner = graphs.CapturedStep(captured, bx, by, learning_rate=1e-3, lambda_nip=0.1, warmup=1)
    eager.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-3)          # = the capture's one warm-up step
    assert torch.equal(eager.fan._model.flat, captured.fan._model.flat)
    ws_before = ops._ws.buf.data_ptr()
    assert any(t.data_ptr() == ws_before for t in runner._pins if isinstance(t, torch.Tensor))
    # a bigger eager job on a third flow: more split-K scratch than the captured step ever asked for ...
    big_rgb = natural_images(6, 128, 128, seed=34)
    big = ManipulationClassification('UNet', distribution=dist, trainable={'nip'}, raw_patch_size=64, device=dev,
                                     nan_check='deferred')
    big.training_step(bayer_from_rgb(big_rgb), big_rgb, lambda_nip=0.1, learning_rate=1e-3)
    for ws in (ops._ws, ops._ws_side):                                        # ... and an explicit re-growth of both
        if ws.buf is not None:
            ws.get(2 * ws.buf.numel(), dev)
    assert ops._ws.buf.data_ptr() != ws_before
    # ... and enough other strengths through the captured flow's own manipulations to evict its cached filter tables
    Y = captured.nip.process(raw)
    for k in range(10):
        captured.run_manipulations(Y, override={'sharpen': 0.3 + 0.1 * k, 'resample': 50, 'gaussian': 0.6 + 0.1 * k, 'jpeg': 80})
    junk = torch.full((ops._ws.buf.numel(),), 255, dtype=torch.uint8, device=dev)     # recycle freed blocks with garbage
    del junk
    torch.cuda.synchronize()
    loss_e, _ = eager.training_step(bx, by, lambda_nip=0.1, learning_rate=1e-3)
    loss_c, _ = runner.step()
    eager.check_nan(), captured.check_nan()
    assert abs(float(loss_e) - float(loss_c)) <= 1e-6 * abs(float(loss_e))
    for a, b in ((eager.fan, captured.fan), (eager.nip, captured.nip)):
        assert torch.equal(a._model.flat_grad, b._model.flat_grad)


def test_validate_fan_on_device(dev):
    """validation.validate_fan (device-side decisions + confusion counts, one read-back) against the reference's host loop
    (training/validation.py:163-202) restated with numpy on the same decisions."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.training import manipulation as tm, validation
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    ops.set_compute('f32')
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    wf = ManipulationClassification('UNet', manipulations=['sharpen:1', 'gaussian:1'], distribution=dist, trainable={'nip'},
                                    raw_patch_size=16, device=dev)
    data = tm.SyntheticDataset(4, 23, patch_size=16)
    acc, conf, labels = validation.validate_fan(wf, data, get_labels=True)
    size, k = 10, wf.n_classes
    ref_conf, ref_acc, ref_labels = np.zeros((k, k)), [], []
    for b in range(23 // size):
        bx = data.next_validation_batch(b, size)[0]
        truth = wf._batch_labels(size)
        pred = wf.run_workflow_to_decisions(bx)
        ref_labels += list(pred)
        for c in range(k):
            for c_ in range(k):
                ref_conf[c, c_] += np.sum((truth == c) * (pred == c_))
        ref_acc.append(np.mean(pred == truth))
    assert labels == [int(v) for v in ref_labels]
    assert np.array_equal(conf, ref_conf / 20) and abs(acc - np.mean(ref_acc)) < 1e-12
    probs = torch.tensor([[0.2, 0.5, 0.5], [0.9, 0.05, 0.05], [0.1, 0.1, 0.8]], device=dev)
    lab = torch.tensor([1, 0, 0], dtype=torch.int32, device=dev)
    cm = torch.zeros((3, 3), dtype=torch.int64, device=dev)
    pred = ops.confusion_accumulate(probs, lab, cm)
    assert pred.tolist() == [1, 0, 2] and cm.tolist() == [[1, 0, 1], [0, 1, 0], [0, 0, 0]]      # first maximum wins


def test_twitter_dcn_at_256(dev):
    """configs[2] at its real size: TwitterDCN-32C on a 256x256 patch (models/compression.py:197-279) - hard latent indices
    exact, reconstruction 1e-4, entropy 1e-5 and the training loss against the float64 oracle."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    dcn = compression.TwitterDCN(patch_size=256, device=dev)
    x = natural_images(1, 256, 256, seed=17)
    p = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
    with torch.no_grad():
        y_ref, ent_ref, lat_ref = onets.dcn_forward(p, to64(x))
        loss_ref = float(onets.dcn_loss(to64(x), y_ref, ent_ref, 250.0))
    xt = torch.from_numpy(x).to(dev)
    y, ent, ctx = dcn.forward(xt, training=True)
    lat = ctx[0]['latent'].cpu().numpy()
    assert lat.shape == (1, 32, 32, 32)
    assert np.array_equal(np.round(lat), np.round(lat_ref.numpy())), 'latent indices differ'
    assert_close(y.cpu().numpy(), y_ref.numpy(), 1e-4, what='DCN reconstruction at 256x256')
    assert abs(float(ent.item()) - float(ent_ref)) < 1e-5
    l2, _ = ops.l2_loss(xt, y)
    assert abs(float(l2.item()) + 250.0 * float(ent.item()) - loss_ref) / loss_ref < 1e-4


def test_twitter_dcn_gradients_at_256(dev):
    """configs[2] at its real patch size, backward: every parameter gradient of the l2 + 250 H loss of two 256 x 256 patches
    (train_dcn.py; models/compression.py:197-279) against the float64 oracle in parity mode; in throughput mode (the bench line's)
    the reconstruction by PSNR, the loss, and the kernel gradients by direction - the 64 x 64 x 128 residual blocks and the
    128 x 128 stride-2 layers are launch shapes the 32-pixel tests do not reach."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    dcn = compression.TwitterDCN(patch_size=256, device=dev)
    x = natural_images(2, 256, 256, seed=19)
    p = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
    for v in p.values():
        v.requires_grad_(True)
    y_ref, ent_ref, lat_ref = onets.dcn_forward(p, to64(x))
    loss_ref = onets.dcn_loss(to64(x), y_ref, ent_ref, 250.0)
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))
    xt = torch.from_numpy(x).to(dev)
    y, ent, ctx = dcn.forward(xt, training=True)
    assert np.array_equal(np.round(ctx[0]['latent'].cpu().numpy()), np.round(lat_ref.detach().numpy())), 'latent indices differ'
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 1e-4, what='DCN reconstruction at 256x256')
    l2, dy = ops.l2_loss(xt, y, grad_scale=1.0)
    dcn.backward(ctx, dy, entropy_coef=250.0)
    # the same restatement in float32 (the arithmetic the reference runs in) bounds what a float32 implementation can be asked for:
    # at this size the LeakyReLU in front of the first residual block sees pre-activations that round to either side of zero, and
    # er1a's gradients move by 2e-3 between float32 and float64 (every other tensor < 7e-4)
    p32 = onets.OrderedDict((k, v.detach().to(torch.float32).requires_grad_(True)) for k, v in p.items())
    y32, e32, _ = onets.dcn_forward(p32, torch.tensor(x))
    g32 = dict(zip(p32.keys(), torch.autograd.grad(onets.dcn_loss(torch.tensor(x), y32, e32, 250.0), list(p32.values()))))
    got, rows, loose = grads_of(dcn), [], []
    for k in p:
        b = g_ref[k].numpy()
        scale = max(np.abs(b).max(), 1e-12)
        e_prod = np.abs(got[k].astype(np.float64) - b).max() / scale
        e_ref32 = np.abs(g32[k].numpy().astype(np.float64) - b).max() / scale
        rows.append((e_prod, k))
        if e_ref32 > 1e-3 / 1.5:
            loose.append(k)
        assert e_prod <= max(1e-3, 2.0 * e_ref32), (k, e_prod, e_ref32)
    assert len(loose) <= 3 and all(k.split('/')[0] in ('er1a', 'e2') for k in loose), loose
    worst = max(rows)
    ops.set_compute('bf16')
    try:
        db = compression.TwitterDCN(patch_size=256, device=dev)
        db.load_state_dict(dcn.state_dict())
        yb, eb, cb = db.forward(xt, training=True)
        lb, dyb = ops.l2_loss(xt, yb.float(), grad_scale=1.0)
        db.backward(cb, dyb, entropy_coef=250.0)
        gb = grads_of(db)
    finally:
        ops.set_compute('f32')
    psnr = 10 * np.log10(1.0 / np.mean((yb.float().cpu().numpy().astype(np.float64) - y_ref.detach().numpy()) ** 2))
    total = float(lb.item()) + 250.0 * float(eb.item())
    assert psnr > 40, psnr
    assert abs(total - float(loss_ref.detach())) / float(loss_ref.detach()) < 3e-2, (total, float(loss_ref.detach()))
    lo = 1.0
    for k in p:
        if k.endswith('/kernel'):
            a, g = gb[k].ravel().astype(np.float64), g_ref[k].numpy().ravel()
            cos = float(a @ g / (np.linalg.norm(a) * np.linalg.norm(g) + 1e-300))
            lo = min(lo, cos)
            assert cos > 0.9, (k, cos)
    print('DCN at 256 x 256: worst parity-mode gradient {:.2e} ({}), throughput mode PSNR {:.1f} dB, worst cosine {:.4f}'.format(
        worst[0], worst[1], psnr, lo))


def test_full_channel_smooth_variant_bf16_gradient_directions(dev):
    """configs[4] with every ill-conditioned or hard decision taken out of the channel: the throughput mode's gradient of EVERY
    parameter tensor of UNet, codec and FAN must point where the float32 mode's does (cosine > 0.98; a wrong sign or a dropped term
    in one layer fails this, which the 0.7 floor the real channel needs for the UNet would not catch).  Taken out
    (tools/c5_wf_diag.py measures each): the jpeg manipulation and the codec's hard latent (rounding='identity'), the entropy term
    (the soft histogram of a CONTINUOUS latent is steep between the centres: its gradient decorrelates under bf16 perturbations;
    at the hard-quantised latent of the real codec it is stable), and the sharpen manipulation - its backward runs through
    rgb->hsv, whose Jacobian grows like 1 / (max - min) on near-grey pixels; the codec's summed l2 term sends a large gradient
    through it and the two modes' UNet outputs differ enough there for the few dominating pixels to change (|g| 4.3e4 vs 1.9e5,
    median cosine 0.42) - a property of the reference's channel, not of either mode.  With resample + gaussian the smallest
    cosine over all tensors is 0.995."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    manips = ['resample:50', 'gaussian:0.83']
    rgb = natural_images(2, 128, 128, seed=9)
    raw = bayer_from_rgb(rgb)
    res = {}
    try:
        for mode in ('f32', 'bf16'):
            ops.set_compute(mode)
            dcn = compression.TwitterDCN(patch_size=128, rounding='identity', entropy_weight=0, device=dev)
            dist = {'downsampling': 'none', 'compression': 'dcn', 'compression_params': {'model': dcn}}
            wf = ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip', 'dcn'},
                                            raw_patch_size=64, device=dev)
            loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, lambda_dcn=0.1, learning_rate=1e-4)
            res[mode] = (float(parts['ce']), float(parts['nip']), float(parts['dcn']),
                         {'nip/' + k: v for k, v in grads_of(wf.nip).items()}, {'dcn/' + k: v for k, v in grads_of(dcn).items()},
                         {'fan/' + k: v for k, v in grads_of(wf.fan).items()})
    finally:
        ops.set_compute('f32')
    assert abs(res['bf16'][0] - res['f32'][0]) < 5e-2
    assert abs(res['bf16'][1] - res['f32'][1]) / res['f32'][1] < 2e-2 and abs(res['bf16'][2] - res['f32'][2]) / res['f32'][2] < 2e-2
    cos = lambda a, b: float(a.ravel() @ b.ravel() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    low = {}
    for gi in (3, 4, 5):
        for k, a in res['f32'][gi].items():
            if a.size >= 16 and np.linalg.norm(a) > 0:          # scalars (latent scaling) and untouched tensors carry no direction
                c = cos(a, res['bf16'][gi][k])
                if c <= 0.98:
                    low[k] = round(c, 4)
    assert not low, sorted(low.items(), key=lambda kv: kv[1])[:40]


def test_dcn_bf16_storage_inside_residual_blocks_is_bit_neutral(dev, monkeypatch):
    """Throughput mode stores the tensors inside the codec's residual blocks (the activation between the two convolutions and
    its gradient) as bf16: their consumers round to bf16 or test the sign, so the reconstruction, the latent and the entropy are
    bit-identical to float32 storage, the weight gradients agree to summation order (different kernel variants take bf16 inputs)
    and only the bias gradients of the blocks' first layers see the rounding (they sum the stored gradient).
    (With bf16 storage the second layer otherwise runs as a 3x3 layer on the space-to-depth image of the first - ANOTHER kernel,
    i.e. another float32 summation order, 3e-7 on its output, which the bf16 roundings and the gamma = 25 soft codebook behind
    it turn into 2 % on the encoder's gradients: switched off here to compare storage with storage; that form has its own
    oracle test, test_strided_layer_on_a_space_to_depth_stored_input, and runs in every other codec test.)"""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    monkeypatch.setattr(compression, '_NO_S2D_CHAIN', True)         # the A/B switch NIMG_NO_S2D_CHAIN, read once at import
    x = torch.from_numpy(natural_images(2, 64, 64, seed=23)).to(dev)
    ops.set_compute('bf16')
    out = {}
    try:
        for store in (True, False):
            ops.STORE_BF16 = store
            dcn = compression.TwitterDCN(patch_size=64, device=dev)
            y, ent, ctx = dcn.forward(x, training=True)
            assert (ctx[0]['er1a'].dtype == torch.bfloat16) == store and ctx[0]['n1'].dtype == torch.float32
            _, dy = ops.l2_loss(x, y, grad_scale=1.0)
            dcn.backward(ctx, dy, entropy_coef=250.0)
            ops.join_side_stream()
            out[store] = (y.clone(), ctx[0]['latent'].clone(), float(ent.item()),
                          {k: v.clone() for k, v in dcn._model.g.items()})
    finally:
        ops.STORE_BF16 = True
        ops.set_compute('f32')
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1]) and out[True][2] == out[False][2]
    worst = {'kernel': 0.0, 'bias': 0.0}
    for k, gb in out[False][3].items():
        rel = float((out[True][3][k] - gb).abs().max()) / (float(gb.abs().max()) + 1e-30)
        kind = 'bias' if k.endswith('/bias') else 'kernel'
        worst[kind] = max(worst[kind], rel)
    # kernels: the same bf16 products, another summation order.  Biases of the blocks' first layers: the fused bias gradient sums
    # the gradient tensor as stored - bf16-rounded values instead of float32 ones (unbiased, 2^-9 relative per element)
    assert worst['kernel'] < 5e-5 and worst['bias'] < 3e-3, worst


def test_full_channel_with_learned_codec_in_throughput_mode(dev):
    """configs[4]: UNet -> manipulations -> TwitterDCN -> FAN, trainable {nip, dcn}, in the bf16 throughput mode: losses against
    the float64 oracle of the same step and against the float32 mode of the same weights; gradient directions of the two modes."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    manips = ['sharpen:1', 'resample:50', 'gaussian:0.83', 'jpeg:80']
    rgb = natural_images(2, 128, 128, seed=9)
    raw = bayer_from_rgb(rgb)
    res = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        dcn = compression.TwitterDCN(patch_size=128, device=dev)
        dist = {'downsampling': 'none', 'compression': 'dcn', 'compression_params': {'model': dcn}}
        wf = ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip', 'dcn'},
                                        raw_patch_size=64, device=dev)
        if mode == 'f32':
            ref = owf.Workflow(manipulations=manips, codec='dcn', trainable=('nip', 'dcn'))
            _sync_oracle(wf, ref)
            ref.dcn = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
            with torch.no_grad():       # forward only: the three loss terms of workflows/...:267-277
                Yr, cr, Cr, er, pr = ref.run_workflow(to64(raw))
                parts_ref = {'ce': float(T.sparse_ce_from_probs(pr, ref.batch_labels(2))), 'nip': float(T.mse255(to64(rgb), Yr)),
                             'dcn': float(onets.dcn_loss(cr, Cr, er))}
        loss, parts = wf.training_step(raw, rgb, lambda_nip=0.1, lambda_dcn=0.1, learning_rate=1e-4)
        res[mode] = (float(parts['ce']), float(parts['nip']), float(parts['dcn']), grads_of(wf.nip), grads_of(dcn), grads_of(wf.fan))
    for mode, (tce, trel) in (('f32', (5e-3, 1e-3)), ('bf16', (5e-2, 2e-2))):
        ce, nip, dl = res[mode][:3]
        assert abs(ce - parts_ref['ce']) < tce, (mode, ce, parts_ref['ce'])
        assert abs(nip - parts_ref['nip']) / parts_ref['nip'] < trel, (mode, nip, parts_ref['nip'])
        assert abs(dl - parts_ref['dcn']) / parts_ref['dcn'] < trel, (mode, dl, parts_ref['dcn'])
    cos = lambda a, b: float(a.ravel() @ b.ravel() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    # the UNet's gradient arrives through the codec's hard latent quantisation (straight-through) and the rounding inside the
    # jpeg manipulation: bf16 operands flip a few of those decisions, hence the looser bound on its direction
    found = {k: cos(res['f32'][gi][k], res['bf16'][gi][k])
             for gi, keys in ((3, ('ec12/kernel', 'dc42/kernel')), (4, ('e2/kernel', 'er2a/kernel', 'd256/kernel')),
                              (5, ('conv3/kernel', 'dense/kernel'))) for k in keys}
    floors = {'ec12/kernel': 0.7, 'dc42/kernel': 0.7}
    assert all(v > floors.get(k, 0.9) for k, v in found.items()), found


# ----------------------------------------------------------------------------------------------------------------------
# the hyper-parameters the reference accepts beyond its defaults
ACTS = ['relu', 'tanh', 'sigmoid', 'softsign']


def test_activation_ops(dev):
    """nimg_activation_fwd / _bwd: helpers/tf_helpers.py:22-28 `activation_mapping` element-wise, the derivative from the output."""
    from neural_imaging_amd import ops
    x = torch.from_numpy(np.random.RandomState(3).uniform(-3, 3, (2, 5, 7, 9)).astype(np.float32))       # 630 values: a scalar tail
    gy = torch.from_numpy(np.random.RandomState(4).uniform(-1, 1, x.shape).astype(np.float32))
    for kind in ['leaky_relu'] + ACTS:
        xr = to64(x.numpy()).requires_grad_(True)
        yr = T.activation(xr, kind)
        (yr * to64(gy.numpy())).sum().backward()
        y = ops.activation(x.to(dev), kind)
        assert_close(y.cpu().numpy(), yr.detach().numpy(), 1e-6, 1e-6, what=kind)
        dx = ops.activation_bwd(gy.to(dev), y, kind)
        assert_close(dx.cpu().numpy(), xr.grad.numpy(), 2e-6, 1e-5, what=kind + ' derivative')
        y2 = x.to(dev).clone()
        assert ops.activation(y2, kind, out=y2) is y2 and torch.equal(y2, y)                 # in place
    with pytest.raises(KeyError):
        ops.activation(x.to(dev), 'gelu')


@pytest.mark.parametrize('act', ACTS)
def test_unet_activations(dev, act):
    """UNet(activation=...) (models/pipelines.py:179-183): output, intermediate tensors and every parameter gradient against
    the oracle for the four members of activation_mapping besides LeakyReLU."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import pipelines
    net = pipelines.UNet(patch_size=16, device=dev, n_steps=3, activation=act)
    rgb = natural_images(2, 32, 32, seed=31)
    raw = bayer_from_rgb(rgb)
    p = oracle_params(net)
    for v in p.values():
        v.requires_grad_(True)
    y_ref, t_ref = onets.unet_forward(p, to64(raw), n_steps=3, return_tensors=True, activation=act)
    loss_ref = T.mse255(y_ref, to64(rgb))
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))
    for mode, tol_y, tol_g in (('f32', 1e-4, 3e-4), ('bf16', 3e-2, None)):
        ops.set_compute(mode)
        y, ctx = net.forward(torch.from_numpy(raw).to(dev), training=True)
        assert_close(y.cpu().numpy(), y_ref.detach().numpy(), tol_y, what='UNet({}) output, {}'.format(act, mode))
        loss, dy = ops.mse255(y, torch.from_numpy(rgb).to(dev), grad_scale=1.0)
        net.backward(ctx, dy)
        if tol_g is not None:
            for name in ('ec11', 'ec22', 'dc11', 'dc22'):
                assert_close(ctx[name].cpu().numpy(), t_ref[name].detach().numpy(), 1e-4, 1e-4, what=name)
            check_grads(grads_of(net), g_ref, list(p.keys()), tol=tol_g)
        else:           # throughput mode: float32-stored tensors, bf16 matrix operands - the gradients point the same way
            got = grads_of(net)
            for k in p:
                a, b = got[k].ravel(), g_ref[k].numpy().ravel()
                if np.linalg.norm(b) > 0 and a.size >= 16:
                    assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)) > 0.98, (act, k)
    with pytest.raises(ValueError):
        pipelines.UNet(patch_size=16, device=dev, activation='gelu')


@pytest.mark.parametrize('act', ACTS)
@pytest.mark.parametrize('head', ['gap', 'dense'])
def test_fan_activations(dev, act, head):
    """FAN(activation=...) (models/forensics.py:55-59, 69, 76, 87) with the GAP head and with hidden Dense layers + dropout."""
    from neural_imaging_amd.models import forensics
    kw = dict(use_gap=True, n_dense=0) if head == 'gap' else dict(use_gap=False, n_dense=2, dropout=0.25)
    fan = forensics.FAN(n_classes=5, patch_size=32, device=dev, n_convolutions=3, activation=act, **kw)
    x = natural_images(5, 32, 32, seed=35)
    labels = np.array([0, 1, 2, 3, 4], np.int32)
    p = oracle_params(fan)
    for v in p.values():
        v.requires_grad_(True)
    masks = None
    if head == 'dense':
        rng = np.random.RandomState(8)
        masks = [torch.from_numpy((rng.uniform(size=(5, d.cout)) >= 0.25).astype(np.uint8)) for d in fan._hidden]
        fan.dropout_masks = [m.clone() for m in masks]
    xt = to64(x).requires_grad_(True)
    probs_ref = onets.fan_forward(p, xt, n_convolutions=3, use_gap=kw['use_gap'], dropout=kw.get('dropout', 0.0),
                                  dropout_masks=masks, activation=act)
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    gr = torch.autograd.grad(loss_ref, list(p.values()) + [xt])
    g_ref = dict(zip(p.keys(), gr[:-1]))
    probs, ctx = fan.forward(torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev), training=True)
    assert_close(probs.cpu().numpy(), probs_ref.detach().numpy(), 1e-4, what='FAN({}) probabilities'.format(act))
    loss, dx = fan.backward(ctx, need_input_grad=True)
    assert abs(float(loss.item()) - float(loss_ref.detach())) < 1e-4
    check_grads(grads_of(fan), g_ref, list(p.keys()), tol=1e-3)
    assert_close(dx.cpu().numpy(), gr[-1].numpy(), 1e-7, 2e-3, what='FAN input gradient')


@pytest.mark.parametrize('act', ACTS)
def test_twitter_dcn_activations(dev, act):
    """TwitterDCN(activation=...) (models/compression.py:197-215): every layer activation follows the hyper-parameter EXCEPT
    the tf.nn.leaky_relu in front of the first residual block (:224)."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    dcn = compression.TwitterDCN(patch_size=32, device=dev, activation=act)
    x = natural_images(2, 32, 32, seed=14)
    p = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
    for v in p.values():
        v.requires_grad_(True)
    y_ref, ent_ref, lat_ref = onets.dcn_forward(p, to64(x), activation=act)
    loss_ref = onets.dcn_loss(to64(x), y_ref, ent_ref, 250.0)
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))
    xt = torch.from_numpy(x).to(dev)
    y, ent, ctx = dcn.forward(xt, training=True)
    assert np.array_equal(np.round(ctx[0]['latent'].cpu().numpy()), np.round(lat_ref.detach().numpy()))
    assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 1e-4, what='DCN({}) reconstruction'.format(act))
    assert abs(float(ent.item()) - float(ent_ref)) < 1e-5
    _, dy = ops.l2_loss(xt, y, grad_scale=1.0)
    dcn.backward(ctx, dy, entropy_coef=250.0)
    check_grads(grads_of(dcn), g_ref, list(p.keys()), tol=1e-3)
    ops.set_compute('bf16')                                    # throughput mode: runs (float32 storage), same reconstruction
    yb, _, ctxb = dcn.forward(xt, training=True)
    dcn.backward(ctxb, dy, entropy_coef=250.0)
    assert float((yb - y).abs().max()) < 0.1 and all(torch.isfinite(v).all() for v in dcn._model.g.values())


@pytest.mark.parametrize('rounding,bpf,nf', [('soft', 5, 32), ('sin', 5, 32), ('identity', 4, 16), ('soft-codebook', 7, 32),
                                             ('soft-codebook', 8, 16), ('soft-codebook', 6, 64), ('soft', 8, 64),
                                             ('soft-codebook', 3, 16)])
def test_twitter_dcn_hyperparameters(dev, rounding, bpf, nf):
    """The codec's hyper-parameter space (models/compression.py:53-59, config/twitter.csv: n_features 16 / 32 / 64): latent
    rounding soft | sin | identity (models/layers.py:118-134) next to the soft codebook, codebooks of 3 .. 8 bits per feature
    (8 .. 256 centres), 16 / 64 latent features - reconstruction, latent, entropy and all gradients against the oracle."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    dcn = compression.TwitterDCN(patch_size=32, device=dev, rounding=rounding, latent_bpf=bpf, n_features=nf)
    assert dcn.get_codebook().size == 2 ** bpf and dcn.latent_shape == (4, 4, nf)
    dcn._model.p['latent_scaling'].fill_(3.0)                  # spread the latent over several codebook entries
    x = natural_images(2, 32, 32, seed=15)
    p = onets.OrderedDict((k, to64(v)) for k, v in dcn.state_dict().items())
    for v in p.values():
        v.requires_grad_(True)
    y_ref, ent_ref, lat_ref = onets.dcn_forward(p, to64(x), latent_bpf=bpf, rounding=rounding)
    loss_ref = onets.dcn_loss(to64(x), y_ref, ent_ref, 250.0)
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))
    xt = torch.from_numpy(x).to(dev)
    y, ent, ctx = dcn.forward(xt, training=True)
    lat = ctx[0]['latent'].cpu().numpy()
    if rounding in ('soft', 'soft-codebook'):
        # hard values: equal wherever the float32 / float64 pre-quantisation values are not within 1e-4 of a rounding tie
        zr = lat_ref.detach().numpy()
        assert (np.round(lat) != np.round(zr)).mean() < 1e-3
        keep = np.round(lat) == np.round(zr)
    else:
        assert_close(lat, lat_ref.detach().numpy(), 1e-4, 1e-4, what='latent ({})'.format(rounding))
        keep = None
    if keep is None or keep.all():
        assert_close(y.cpu().numpy(), y_ref.detach().numpy(), 1e-4, what='reconstruction')
        assert abs(float(ent.item()) - float(ent_ref)) < 2e-5
        _, dy = ops.l2_loss(xt, y, grad_scale=1.0)
        dcn.backward(ctx, dy, entropy_coef=250.0)
        check_grads(grads_of(dcn), g_ref, list(p.keys()), tol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('hw,c', [((16, 16), 256), ((8, 8), 256), ((8, 16), 128), ((8, 8), 64)])
def test_fused_head_kernels_against_the_generic_path(dev, hw, c):
    """csrc/head.hip (throughput mode): 1x1 conv + LeakyReLU + global average pooling in one pass, the activation kept as ONE SIGN
    BIT per value; the input gradient of the 1x1 layer built from those bits and the classifier's dlogits.  Against the generic
    kernels on the same bf16 operands (1x1 convolution -> pooling; pooling backward -> 1x1 input gradient): the same products, other
    summation orders."""
    from neural_imaging_amd import ops
    n, k = 7, 5
    g = torch.Generator(device='cpu').manual_seed(3)
    x = (torch.randn((n, hw[0], hw[1], c), generator=g) * 0.7).to(dev).to(torch.bfloat16)
    w = (torch.randn((1, 1, c, c), generator=g) * (1.0 / np.sqrt(c))).to(dev)
    b = (torch.randn((c,), generator=g) * 0.1).to(dev)
    wd = (torch.randn((c, k), generator=g) * 0.1).to(dev)
    dlogits = (torch.randn((n, k), generator=g) * 0.3).to(dev)
    ops.set_compute('bf16')
    try:
        assert ops.head_fused_ok(x, c)
        gap, mask, mask_p = ops.head_fwd(x, w, b)
        a = ops.conv2d(x, w, b, act='leaky_relu').float()                     # generic 1x1 kernel, float32 result
        assert_close(gap.cpu().numpy(), a.mean(dim=(1, 2)).cpu().numpy(), 1e-5, 1e-4, what='pooled feature')
        bits = (mask.view(n, hw[0], hw[1], c // 32, 1) >> torch.arange(32, device=dev, dtype=torch.int32)) & 1
        want = (a > 0).view(n, hw[0], hw[1], c // 32, 32).to(torch.int32)
        # (a value within float32 rounding of zero may fall on either side in another summation order: none in this draw)
        assert int((bits != want).sum().item()) <= 2
        # backward: the generic path = pooling backward (float32 gradient tensor) -> 1x1 input gradient with the mask of the input
        dact_ref = (dlogits @ wd.t() / float(hw[0] * hw[1]))[:, None, None, :] * torch.where(a > 0, 1.0, ops.LRELU_ALPHA)
        dact = ops.head_dact(mask, dlogits, wd, x.shape)
        assert_close(dact.float().cpu().numpy(), dact_ref.to(torch.bfloat16).float().cpu().numpy(), 1e-6, 1e-2, what='dAct')
        dx_ref = ops.conv2d_dgrad(dact_ref.contiguous(), w, hw, act_mask=x, out_bf16=True).float().cpu().numpy()
        dx = ops.head_dgrad(mask, dlogits, wd, w, x, x.shape).float().cpu().numpy()
        scale = np.abs(dx_ref).max()
        assert np.abs(dx - dx_ref).max() <= 1e-2 * scale                      # one bf16 ulp of the largest entry
        assert np.mean(np.abs(dx - dx_ref) > 1e-3 * scale) < 0.02
        # the same bits, pixel-major
        bits_p = (mask_p.view(n, hw[0] * hw[1] // 32, 1, c) >> torch.arange(32, device=dev, dtype=torch.int32).view(1, 1, 32, 1)) & 1
        assert torch.equal(bits_p.reshape(n, hw[0], hw[1], c), bits.reshape(n, hw[0], hw[1], c))
        # weight + bias gradient of the 1x1 layer: the generic kernel on the materialised gradient
        dw_ref, db_ref = torch.empty_like(w), torch.empty_like(b)
        ops.conv2d_wgrad(x, dact, 1, dw=dw_ref, db=db_ref)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        ops.head_wgrad(x, mask_p, dlogits, wd, dw.view(c, c), db)
        assert_close(dw.cpu().numpy(), dw_ref.cpu().numpy(), 1e-7, 1e-4, what='head dW')
        assert_close(db_ref.cpu().numpy(), dact_ref.sum(dim=(0, 1, 2)).cpu().numpy(), 1e-7, 1e-2, what='generic db')   # (bf16-rounded tensor)
        assert_close(db.cpu().numpy(), dact_ref.sum(dim=(0, 1, 2)).cpu().numpy(), 1e-7, 1e-4, what='head db')      # (exact g)
        dx0 = ops.head_dgrad(mask, dlogits, wd, w, None, x.shape).float().cpu().numpy()        # no mask below
        dx0_ref = ops.conv2d_dgrad(dact_ref.contiguous(), w, hw, out_bf16=True).float().cpu().numpy()
        assert np.abs(dx0 - dx0_ref).max() <= 1e-2 * np.abs(dx0_ref).max()
    finally:
        ops.set_compute('f32')


@pytest.mark.gpu
def test_fan_with_the_fused_head(dev, monkeypatch):
    """A FAN whose last feature map has 64 pixels (patch 128): throughput mode takes the fused head (csrc/head.hip).  Same
    probabilities / loss / gradients as with the generic head kernels (NIMG_NO_HEAD_FUSED path) up to bf16 rounding, and the float64
    oracle within the throughput mode's usual tolerances."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import forensics
    x = natural_images(5, 128, 128, seed=41)
    labels = np.array([0, 1, 2, 3, 4], np.int32)
    xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(labels).to(dev)
    res = {}
    ops.set_compute('bf16')
    try:
        for fused in (True, False):
            monkeypatch.setattr(ops, 'HEAD_FUSED', fused)
            fan = forensics.FAN(n_classes=5, patch_size=128, device=dev)
            probs, ctx = fan.forward(xd, ld, training=True)
            assert ('head_mask' in ctx) == fused
            loss, dx = fan.backward(ctx, need_input_grad=True)
            res[fused] = (probs.cpu().numpy(), float(loss.item()), dx.cpu().numpy(), grads_of(fan))
            assert np.array_equal(fan.forward(xd)[0].cpu().numpy(), probs.cpu().numpy())     # inference: no mask, same numbers
    finally:
        ops.set_compute('f32')
    a, b = res[True], res[False]
    assert_close(a[0], b[0], 1e-5, 1e-4, what='probabilities, fused vs generic head')
    assert abs(a[1] - b[1]) < 1e-5
    cos = lambda u, v: float((u * v).sum() / (np.linalg.norm(u) * np.linalg.norm(v) + 1e-30))
    assert cos(a[2], b[2]) > 0.999, cos(a[2], b[2])
    for k in a[3]:
        assert cos(a[3][k].ravel(), b[3][k].ravel()) > 0.999, (k, cos(a[3][k].ravel(), b[3][k].ravel()))
    # the float64 oracle
    fan = forensics.FAN(n_classes=5, patch_size=128, device=dev)
    p = oracle_params(fan)
    for v in p.values():
        v.requires_grad_(True)
    probs_ref = onets.fan_forward(p, to64(x))
    loss_ref = T.sparse_ce_from_probs(probs_ref, labels)
    g_ref = dict(zip(p.keys(), torch.autograd.grad(loss_ref, list(p.values()))))
    assert np.abs(a[0] - probs_ref.detach().numpy()).max() < 2e-2
    for k in ('conv1x1/kernel', 'conv4/kernel', 'conv3/kernel', 'dense/kernel' if 'dense/kernel' in g_ref else list(g_ref)[-2]):
        assert cos(a[3][k].ravel(), g_ref[k].numpy().ravel()) > 0.98, k


def test_learned_codec_channel_learns_in_both_compute_modes(dev):
    """BASELINE.json configs[4] (UNet -> manipulations -> TwitterDCN -> FAN, nip + dcn + fan trainable) LEARNS in the bf16
    throughput mode as it does in float32 (at random initialisation the bf16 gradient that reaches the deep UNet
    levels through the codec is mostly conditioning noise - section 5 of DESIGN.md - so single-step gradient cosines say little;
    what matters is the trajectory).  One initialisation, the same batches, 500 joint steps per mode: the objective falls, and the
    three things the channel is trained for - ISP fidelity, codec reconstruction, classification loss - end up together."""
    from neural_imaging_amd import ops
    from neural_imaging_amd.models import compression
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    rgb = natural_images(48, 64, 64, seed=300)
    raw = bayer_from_rgb(rgb)
    vrgb = natural_images(16, 64, 64, seed=400)
    vraw = bayer_from_rgb(vrgb)
    out = {}
    for mode in ('f32', 'bf16'):
        ops.set_compute(mode)
        try:
            torch.manual_seed(0)
            codec = compression.TwitterDCN(patch_size=64, n_features=32, device=dev)
            dist = {'downsampling': 'none', 'compression': 'dcn', 'compression_params': {'model': codec}}
            wf = ManipulationClassification('UNet', manipulations=['sharpen:1', 'gaussian:0.83'], distribution=dist,
                                            trainable={'nip', 'dcn'}, raw_patch_size=32, device=dev)
            rng = np.random.RandomState(1)
            losses, ces = [], []
            for step in range(500):
                idx = rng.choice(48, 8, replace=False)
                loss, parts = wf.training_step(raw[idx], rgb[idx], lambda_nip=0.1, lambda_dcn=0.1, learning_rate=1e-4)
                losses.append(float(loss))
                ces.append(float(parts['ce']))
            res = wf.run_workflow(vraw)
            Y, c, C = (np.asarray(res[i].numpy() if hasattr(res[i], 'numpy') else res[i]) for i in (0, 1, 2))
            psnr = lambda a, b: float(np.mean(10 * np.log10(1.0 / np.mean((a - b) ** 2, axis=(1, 2, 3)))))
            out[mode] = {'loss': np.array(losses), 'ce': np.array(ces), 'isp_psnr': psnr(Y, vrgb), 'codec_psnr': psnr(C, c)}
        finally:
            ops.set_compute('f32')
    print({m: (round(v['loss'][:10].mean(), 2), round(v['loss'][-10:].mean(), 2), round(v['ce'][-20:].mean(), 3),
               round(v['isp_psnr'], 2), round(v['codec_psnr'], 2)) for m, v in out.items()})
    for m, v in out.items():                    # measured: 3531 -> 253 (float32), 3573 -> 264 (bf16)
        assert v['loss'][-10:].mean() < 0.1 * v['loss'][:10].mean(), (m, v['loss'][:10].mean(), v['loss'][-10:].mean())
    a, b = out['f32'], out['bf16']
    assert abs(a['isp_psnr'] - b['isp_psnr']) < 0.5, (a['isp_psnr'], b['isp_psnr'])             # 14.91 / 15.07 dB
    assert abs(a['codec_psnr'] - b['codec_psnr']) < 1.0, (a['codec_psnr'], b['codec_psnr'])     # 26.57 / 26.86 dB
    assert abs(a['loss'][-10:].mean() - b['loss'][-10:].mean()) < 0.1 * a['loss'][-10:].mean()
    # (the classifier's take-off falls on another step in the two runs - CE 1.03 vs 0.59 at step 500, chance = 1.10: not asserted)
