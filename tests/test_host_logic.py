"""Host-side logic that needs no GPU: hyper-parameter validation, JPEG quality resolution, TF padding rules, resample
operators, workflow construction / error behaviour (mirrors the reference's ValueError / NotImplementedError contract),
checkpoint round trip, and the data-parallel plumbing on 2 gloo ranks."""
import os
import socket
import sys

from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import tfops as T

from neural_imaging_amd import ops, parallel
from util import collect_from_workers
from neural_imaging_amd.helpers import kernels as hk
from neural_imaging_amd.helpers.paramspec import ParamSpec
from neural_imaging_amd.models import forensics, jpeg, pipelines
from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification

DIST = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}


def test_paramspec_semantics():
    h = ParamSpec({'n': (5, int, (2, 6)), 'act': ('leaky_relu', str, {'leaky_relu', 'relu'}), 'x': (1.0, float, None)})
    assert h.n == 5 and h.act == 'leaky_relu'
    h.update(n='3', x=2)
    assert h.n == 3 and isinstance(h.x, float) and h.changed_params() == {'n': 3, 'x': 2.0}
    with pytest.raises(ValueError):
        h.update(n=7)
    with pytest.raises(ValueError):
        h.update(act='gelu')
    with pytest.raises(ValueError):
        h.update(bogus=1)
    with pytest.raises(ValueError):
        h.n = 4
    h.update(n=None)            # None keeps the current value (paramspec.py:146)
    assert h.n == 3 and h.to_json()['act'] == 'leaky_relu'


def test_same_padding_matches_tf_rule():
    for size in (7, 8, 32, 128, 255, 256):
        for k, s in ((1, 1), (3, 1), (5, 1), (5, 2), (2, 2)):
            out, before = ops.same_pads(size, k, s)
            ref_before, ref_after = T.same_pads(size, k, s)
            assert before == ref_before and out == -(-size // s)
            assert (out - 1) * s + k - size <= before + ref_after or ref_before + ref_after == 0
    assert ops.same_pads(256, 5, 2) == (128, 1)          # 1 before / 2 after


def test_jpeg_quality_resolution():
    np.random.seed(0)
    assert jpeg.JPEG.resolve_quality(80) == 80 and jpeg.JPEG.resolve_quality(80.0) == 80
    for _ in range(20):
        assert 50 <= jpeg.JPEG.resolve_quality((50, 90)) < 90          # hi-exclusive like np.random.randint
        assert jpeg.JPEG.resolve_quality([10, 50, 95]) in (10, 50, 95)
    for bad in (None, 0, 101, 'x', (0, 50)):
        with pytest.raises(ValueError):
            jpeg.JPEG.resolve_quality(bad)
    with pytest.raises(ValueError):
        jpeg.JPEG(80, 'bogus')
    with pytest.raises(ValueError):
        jpeg.DifferentiableJPEG(0)
    j = jpeg.JPEG(80, 'soft', device='cpu')
    assert j.summary() == 'JPEG (soft) QF=80' and j.count_parameters() == 0 and j.estimate_qf() == 80
    assert jpeg.JPEG((50, 90), 'soft', device='cpu').summary() == 'JPEG (soft) QF~[50,90]'
    assert repr(j) == 'JPEG(quality=80,codec="soft",trainable=False)'                  # models/jpeg.py:253-257
    # trainable tables (models/jpeg.py:57-62, 265-278): the summaries and the QF estimate read the LEARNED tables
    jt = jpeg.JPEG(70, 'soft', trainable=True, device='cpu')
    assert repr(jt) == 'JPEG(quality=70,codec="soft",trainable=True)' and jt.count_parameters() == 128
    assert jt.summary() == 'JPEG (soft) trainable QF~70/70' and jt.estimate_qf() == 70
    from neural_imaging_amd import ops as _ops
    jt._model.p['Q_mtx_luma'].copy_(_ops.qtables_device(30, 'cpu')[0])
    assert jt.estimate_qf() == 30 and jt.summary() == 'JPEG (soft) trainable QF~30/70'


def test_djpeg_reciprocal_division_is_the_ieee_quotient(tmp_path):
    """csrc/djpeg.hip divides by the IJG table entries with a correctly rounded reciprocal + one residual correction; the claim
    'equal to x / q for every float x and every integer q in 1 .. 255' is checked here on every 7th mantissa (the full sweep,
    tools/probe/div_markstein_check.c without an argument, takes ~15 core-seconds and is what the kernel comment cites)."""
    import subprocess
    exe = str(tmp_path / 'chk')
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'probe', 'div_markstein_check.c')
    subprocess.run(['gcc', '-O2', '-ffp-contract=off', src, '-lm', '-o', exe], check=True)
    out = subprocess.run([exe, '7'], check=True, capture_output=True, text=True).stdout
    assert out.startswith('mismatches 0 '), out
    assert int(out.split('alone:')[1].strip(' )\n')) > 0          # the correction is needed: x * rc alone is often 1 ulp off


def test_side_stream_of_a_gradient_buffer_is_stable():
    """ops._side_index: the parameter gradients are spread round-robin over the side streams, but a gradient buffer (the key) is
    served by the same stream ever after - two launches that accumulate into one buffer must stay ordered."""
    from neural_imaging_amd import ops
    n = ops._SIDE['n']
    keys = [('test-key', i) for i in range(3 * n + 1)]
    first = [ops._side_index(k) for k in keys]
    assert first == [ops._side_index(k) for k in keys]
    assert all(0 <= k < n for k in first) and len(set(first)) == min(n, len(keys))
    assert ops._side_index(None) == 0


def test_nearest_resample_operator_matches_oracle():
    """method='nearest' of manipulation_resample (tf_helpers.py:68-76): the composed axis operator picks the pixels the restated
    ResizeNearestNeighbor picks; down by 2 keeps the odd pixels (floor((o + 0.5) * 2) = 2 o + 1), up by 2 repeats each twice."""
    assert hk.nearest_axis_matrix(8, 4).argmax(axis=1).tolist() == [1, 3, 5, 7]
    assert hk.nearest_axis_matrix(4, 8).argmax(axis=1).tolist() == [0, 0, 1, 1, 2, 2, 3, 3]
    for size, factor in ((32, 50), (64, 73), (20, 30), (48, 100)):
        small = size * factor // 100
        m = hk.nearest_axis_matrix(small, size) @ hk.nearest_axis_matrix(size, small)
        assert ((m == 0) | (m == 1)).all() and (m.sum(axis=1) == 1).all()
        x = torch.rand(1, size, size, 2, dtype=torch.float64)
        ref = T.resize_nearest(T.resize_nearest(x, small, small), size, size)
        got = torch.einsum('ab,nbwc->nawc', torch.tensor(m), x)
        got = torch.einsum('ab,nhbc->nhac', torch.tensor(m), got)
        assert torch.equal(got, ref)


def test_resample_operator_matches_oracle():
    for size, factor in ((32, 50), (64, 73), (48, 40)):
        small = size * factor // 100
        m = hk.bilinear_axis_matrix(small, size) @ hk.bilinear_axis_matrix(size, small)
        x = torch.rand(1, size, size, 2, dtype=torch.float64)
        ref = T.resize_bilinear(T.resize_bilinear(x, small, small), size, size)
        got = torch.einsum('ab,nbwc->nawc', torch.tensor(m), x)
        got = torch.einsum('ab,nhbc->nhac', torch.tensor(m), got)
        assert (got - ref).abs().max() < 1e-12
        op = ops.AxisOperator(m, torch.device('cpu'))
        rowptr, col, val = op.fwd
        assert int(rowptr[-1]) == len(col) and (np.diff(rowptr.numpy()) <= 4).all()
        dense = np.zeros_like(m)
        for r in range(size):
            for e in range(int(rowptr[r]), int(rowptr[r + 1])):
                dense[r, int(col[e])] = float(val[e])
        assert np.abs(dense - m).max() < 1e-7
        rp, cc, vv = op.bwd
        dt = np.zeros_like(m)
        for r in range(size):
            for e in range(int(rp[r]), int(rp[r + 1])):
                dt[r, int(cc[e])] = float(vv[e])
        assert np.abs(dt - m.T).max() < 1e-7


def test_workflow_construction_and_errors():
    wf = ManipulationClassification('UNet', distribution=DIST, trainable={'nip'}, raw_patch_size=32, device='cpu')
    assert wf.n_classes == 5 and wf._forensics_classes[0] == 'native' and wf.downsampling_factor == 1
    assert (wf._batch_labels(2) == np.array([0, 0, 1, 1, 2, 2, 3, 3, 4, 4])).all()
    assert wf.is_trainable('fan') and wf.is_trainable('nip') and not wf.is_trainable('dcn')
    assert wf.nip.count_parameters() == 7763820 and wf.fan.count_parameters() == 1145382
    assert 'UNet' in wf.summary() and 'nsrgj' in wf.summary_compact()
    with pytest.raises(ValueError):
        ManipulationClassification('UNet', distribution=DIST, raw_patch_size=8, device='cpu')
    with pytest.raises(ValueError):
        ManipulationClassification('NoSuchNet', distribution=DIST, device='cpu')
    with pytest.raises(ValueError):
        ManipulationClassification('UNet', manipulations=['sharpen', 'bogus'], distribution=DIST, device='cpu')
    with pytest.raises(ValueError):
        ManipulationClassification('UNet', distribution=DIST, loss_metric='L7', device='cpu')
    with pytest.raises(ValueError):
        ManipulationClassification('UNet', distribution=DIST, trainable={'dcn'}, device='cpu')
    wf2 = ManipulationClassification('ONet', manipulations=['sharpen:0.5', 'gaussian:1'],
                                     distribution={'downsampling': 'pool:2', 'compression': 'none'}, device='cpu',
                                     raw_patch_size=64)
    assert wf2.n_classes == 3 and wf2._strengths['sharpen'] == 0.5 and wf2.downsampling_factor == 2
    assert wf2.fan.patch_size == 64
    # no CPU fallback: running the channel without a GPU must fail loudly, not silently compute elsewhere
    with pytest.raises(RuntimeError):
        wf.run_workflow(np.zeros((1, 32, 32, 4), np.float32))


def test_model_surface_and_checkpoint_roundtrip(tmp_path):
    net = pipelines.UNet(patch_size=32, device='cpu')
    assert net.class_name == 'UNet' and net.scoped_name == 'unet' and net.model_code == 'UNet_5'
    assert net.x.shape == (None, 32, 32, 4) and net.y.shape == (None, 64, 64, 3)
    assert len(net.parameters) == 46 and net.parameters[0].shape == (3, 3, 4, 32)
    net.save_model(str(tmp_path), save_args=True)
    assert os.path.isfile(os.path.join(str(tmp_path), 'unet', 'unet.h5'))
    net2 = pipelines.UNet.restore(str(tmp_path), patch_size=32, device='cpu', seed=99)
    for a, b in zip(net.parameters, net2.parameters):
        assert torch.equal(a, b)
    net.log_metric('loss', 'training', [1.0, 3.0])
    assert net.pop_metric('loss', 'training') == 2.0
    with pytest.raises(ValueError):
        pipelines.UNet(device='cpu', n_steps=9)
    fan = forensics.FAN(5, patch_size=64, device='cpu')
    assert fan._h.use_gap is True and fan._h.n_dense == 0          # ctor values override the ParamSpec defaults
    assert fan._loss([0, 1], np.full((2, 5), 0.2)) == pytest.approx(np.log(5))
    with pytest.raises(ValueError):
        forensics.FAN(1, device='cpu')
    # tfmodel.py:144-148: one row per tensor, shares in per cent
    table = net.count_parameters_breakdown()
    assert list(table.columns) == ['name', 'shape', 'parameters', 'total'] and len(table) == 46
    assert int(table['parameters'].sum()) == net.count_parameters() and table['name'][0] == 'ec11/kernel'
    assert abs(float(table['total'].sum()) - 100.0) < 1.0
    with pytest.raises(NotImplementedError):
        net.deploy_model(str(tmp_path))
    with pytest.raises(NotImplementedError):
        net.migrate_model(str(tmp_path))
    # pipelines.py:114-141: patch descriptions; a bare name is a sub-directory of the NIP snapshot root
    assert net._input_description == '32\u00d732\u00d74' and net._output_description == '64\u00d764\u00d73'
    assert net.summary().endswith('32\u00d732\u00d74 -> 64\u00d764\u00d73')
    assert pipelines.UNet(device='cpu').summary().endswith('(raw) -> (rgb)')            # no fixed patch size (helpers/utils.py:257-263)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        net.save_model('D90')
        assert os.path.isfile(os.path.join('data', 'models', 'nip', 'D90', 'unet', 'unet.h5'))
        net3 = pipelines.UNet(patch_size=32, device='cpu', seed=5)
        net3.load_model('D90')
        assert all(torch.equal(a, b) for a, b in zip(net.parameters, net3.parameters))
    finally:
        os.chdir(cwd)
    isp = pipelines.ClassicISP(patch_size=16, kernel=3, c_filters=(32, 32), cfa_pattern='rggb', device='cpu')
    assert isp.summary() == 'ClassicISP[rggb] + CNN demosaicing [2+1 layers : 3x3x32 -> 1x1x3]'                # pipelines.py:529-539
    assert isp.summary_compact() == 'ClassicISP[rggb, 2+1 conv2D 3x3x32 > 1x1x3]'
    assert pipelines.ClassicISP(patch_size=16, c_filters=(8, 16), device='cpu').summary_compact() == 'ClassicISP[gbrg, 2+1 conv2D 5x5x* > 1x1x3]'


def test_hdf5_reader_against_libhdf5_bytes():
    """helpers/hdf5.py reads a Keras-layout weight file that the real HDF5 library wrote (tests/golden/make_keras_h5.py):
    fixed- and variable-length string attributes, several symbol-table nodes per group, scalar datasets."""
    from neural_imaging_amd.helpers import hdf5, keras_h5
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    exp = np.load(os.path.join(gdir, 'keras_weights_expected.npz'))
    root = hdf5.read_hdf5(os.path.join(gdir, 'keras_weights_h5py.h5'))
    assert root.attrs['backend'] == b'tensorflow' and root.attrs['keras_version'] == b'2.2.4-tf'
    assert len(root.attrs['layer_names']) == 26 and len(root) == 26
    layers = keras_h5.load_weights(os.path.join(gdir, 'keras_weights_h5py.h5'))
    assert [l for l, _ in layers][:4] == ['input_1', 'conv2d', 'max_pooling2d', 'conv2d_transpose']
    assert layers[0][1] == [] and layers[2][1] == []
    got = OrderedDict((l + '|' + w.replace('/', '|'), a) for l, ws in layers for w, a in ws)
    assert list(got.keys()) == [str(k) for k in exp['order']]
    for k, a in got.items():
        assert a.dtype == np.float32 and a.shape == exp[k].shape and np.array_equal(a, exp[k]), k
    assert got['demosaicing_layer|demosaicing_layer|alpha:0'].shape == ()
    with pytest.raises(ValueError):
        hdf5.read_hdf5(os.path.join(gdir, 'keras_weights_expected.npz'))


def test_hdf5_writer_roundtrip_and_libhdf5_acceptance(tmp_path):
    """What helpers/hdf5.py writes it reads back bit-exactly; when this image's HDF5 tools are present (they are not part
    of the product), libhdf5 itself must accept the file too."""
    import shutil
    import subprocess
    from neural_imaging_amd.helpers import hdf5, keras_h5
    rng = np.random.RandomState(3)
    layers = [('input_1', [])]
    for i in range(70):                                      # > 64 entries: more than one symbol node under the B-tree
        n = 'conv2d_{}'.format(i)
        layers.append((n, [(n + '/kernel:0', rng.normal(size=(3, 3, i % 5 + 1, 4)).astype(np.float32)),
                           (n + '/bias:0', rng.normal(size=(4,)).astype(np.float32))]))
    layers.append(('discrete_latent', [('discrete_latent/latent_scaling:0', np.float32(1.25))]))
    path = str(tmp_path / 'w.h5')
    keras_h5.save_weights(path, layers)
    back = keras_h5.load_weights(path)
    assert [l for l, _ in back] == [l for l, _ in layers]
    for (_, ws), (_, wb) in zip(layers, back):
        assert [w for w, _ in ws] == [w for w, _ in wb]
        for (_, a), (_, b) in zip(ws, wb):
            assert b.dtype == np.float32 and np.array_equal(np.asarray(a), b)
    root = hdf5.Group()
    root['ints'] = hdf5.Dataset(np.arange(6, dtype=np.int64).reshape(2, 3), {'note': 7, 'tag': 'abc'})
    root['f64'] = np.linspace(0, 1, 5)
    root['empty'] = np.zeros((0, 3), np.float32)
    hdf5.write_hdf5(str(tmp_path / 'm.h5'), root)
    m = hdf5.read_hdf5(str(tmp_path / 'm.h5'))
    assert m['ints'].value.tolist() == [[0, 1, 2], [3, 4, 5]] and m['ints'].attrs['note'] == 7
    assert m['ints'].attrs['tag'] == b'abc' and m['f64'].value.dtype == np.float64 and m['empty'].shape == (0, 3)
    h5dump = shutil.which('h5dump') or ('/opt/conda/bin/h5dump' if os.path.isfile('/opt/conda/bin/h5dump') else None)
    if h5dump:
        out = subprocess.run([h5dump, '-H', path], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and 'conv2d_69' in out.stdout and 'H5T_IEEE_F32LE' in out.stdout, out.stderr[-400:]
        out = subprocess.run([h5dump, '-d', '/discrete_latent/discrete_latent/latent_scaling:0', path],
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and '1.25' in out.stdout, out.stderr[-400:]


def test_keras_checkpoints_of_every_model(tmp_path):
    """save_model / load_model / restore through <class>.h5 (models/tfmodel.py:150-182) for every model family, with the
    Keras by-order loading rule and its shape check; constants of ClassicISP are not stored."""
    from neural_imaging_amd.helpers import keras_h5
    from neural_imaging_amd.models import compression
    makers = [lambda s: pipelines.UNet(patch_size=16, device='cpu', seed=s),
              lambda s: pipelines.INet(patch_size=16, device='cpu', seed=s, random_init=True),
              lambda s: pipelines.DNet(patch_size=16, device='cpu', seed=s, n_layers=3, n_features=8),
              lambda s: pipelines.ClassicISP(patch_size=16, device='cpu', seed=s, c_filters=(8, 8), kernel=3),
              lambda s: forensics.FAN(5, patch_size=64, device='cpu', seed=s),
              lambda s: forensics.FAN(3, patch_size=32, device='cpu', seed=s, use_gap=False, n_dense=1),
              lambda s: compression.TwitterDCN(patch_size=32, device='cpu', seed=s, n_features=8)]
    for k, make in enumerate(makers):
        a, b = make(1), make(2)
        if a.parameters:
            assert any(not torch.equal(x, y) for x, y in zip(a.parameters, b.parameters))
        d = str(tmp_path / 'm{}'.format(k))
        a.save_model(d, save_args=True)
        path = os.path.join(d, a.scoped_name, a.class_name.lower() + '.h5')
        stored = [w for _, ws in keras_h5.load_weights(path) for w, _ in ws]
        assert len(stored) == len([n for n in a.parameter_names if n not in a._h5_skip])
        b.load_model(d)
        for n, x, y in zip(a.parameter_names, a.parameters, b.parameters):
            assert torch.equal(x, y), (a.class_name, n)
    # the codec's file has the reference's layout: two nested Models, 'encoder' (9 convolutions + the latent scale) and 'decoder'
    layers = keras_h5.load_weights(path)
    assert [l for l, _ in layers] == ['encoder', 'decoder'] and len(layers[0][1]) == 19 and len(layers[1][1]) == 18
    assert layers[0][1][-1][0].startswith('latent_scaling') and all(w.endswith(':0') for _, ws in layers for w, _ in ws)
    isp = makers[3](1)
    a = isp
    a.save_model(str(tmp_path / 'isp'))
    stored = [w for _, ws in keras_h5.load_weights(os.path.join(str(tmp_path / 'isp'), a.scoped_name, a.class_name.lower() + '.h5'))
              for w, _ in ws]
    assert not any('up' in w or 'srgb' in w for w in stored) and isp._h5_skip == ('up/kernel', 'srgb/kernel')
    isp_layers = keras_h5.load_weights(os.path.join(str(tmp_path / 'isp'), a.scoped_name, a.class_name.lower() + '.h5'))
    assert [l for l, _ in isp_layers] == ['demosaicing_layer'] and stored[0].endswith('alpha:0') and 'bilinear' in stored[-1]
    # a file of another architecture is refused with the offending tensor named
    unet = pipelines.UNet(patch_size=16, device='cpu')
    with pytest.raises(ValueError):
        unet.load_keras_weights(path)
    fan_a, fan_b = forensics.FAN(5, patch_size=64, device='cpu'), forensics.FAN(4, patch_size=64, device='cpu')
    fan_a.save_model(str(tmp_path / 'f'))
    with pytest.raises(ValueError, match='shape'):
        fan_b.load_model(str(tmp_path / 'f'))


def _feed_images(n, h, w, seed):
    rng = np.random.RandomState(seed)
    rgb = np.zeros((n, h, w, 3), np.uint8)
    for i in range(n):
        base = np.full((h, w, 3), 40.0 + 25 * i)
        base[h // 4:h // 2, w // 3:] += 80 * rng.uniform(-1, 1, (h // 2 - h // 4, w - w // 3, 3))
        base[:, :w // 4] += 15 * np.sin(np.arange(w // 4) / 2.0)[None, :, None]
        rgb[i] = np.clip(base + rng.normal(0, 2, (h, w, 3)), 0, 255)
    raw = rng.randint(0, 65536, (n, h // 2, w // 2, 4)).astype(np.uint16)
    return raw, rgb


def test_host_dataset_from_directory_and_arrays(tmp_path):
    """helpers/dataset.Dataset over *.npy + *.png pairs (reference helpers/dataset.py, helpers/loading.py): the split, the
    loaded shapes, a training batch equal to the oracle's cut at the coordinates the seeded numpy RNG gives."""
    from PIL import Image
    from oracle import datafeed as odf
    from neural_imaging_amd.helpers import dataset, loading
    raw, rgb = _feed_images(7, 64, 96, 5)
    for i in range(7):
        Image.fromarray(rgb[i]).save(str(tmp_path / 'img_{:02d}.png'.format(i)))
        np.save(str(tmp_path / 'img_{:02d}.npy'.format(i)), raw[i])
    data = dataset.Dataset(str(tmp_path), n_images=4, v_images=3, val_rgb_patch_size=32, val_n_patches=2, randomize=11)
    files = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith('.png'))
    np.random.seed(11)
    np.random.shuffle(files)
    assert data.files['training'] == files[:4] and data.files['validation'] == files[4:7]
    assert data.count_training == 4 and data.count_validation == 6 and data.rgb_patch_size == 32 and data.is_raw_and_rgb()
    assert data['training']['x'].shape == (4, 32, 48, 4) and data['training']['y'].dtype == np.uint8
    assert (data.H, data.W) == (64, 96) and 'raw+rgb' in data.summary()
    order = [int(f[4:6]) for f in files[:4]]
    assert np.array_equal(data['training']['y'], rgb[order]) and np.array_equal(data['training']['x'], raw[order])
    for discard in (None, 'flat', 'flat-aggressive', 'dark-n-textured'):
        np.random.seed(3)
        x, y = data.next_training_batch(1, 2, 32, discard, max_attempts=6)
        np.random.seed(3)
        xy = [odf.sample_patch(rgb[order[2 + b]], 32, discard, 6) for b in range(2)]
        xr, yr = odf.cut_batch(raw[order], rgb[order], [2, 3], xy, 32)
        assert x.dtype == np.float32 and np.array_equal(x, xr) and np.array_equal(y, yr)
    xv, yv = data.next_validation_batch(1, 3)
    assert xv.shape == (3, 16, 16, 4) and yv.shape == (3, 32, 32, 3) and float(yv.max()) <= 1.0
    with pytest.raises(ValueError):
        data.next_training_batch(2, 2, 32)
    with pytest.raises(ValueError):
        dataset.Dataset(str(tmp_path), n_images=6, v_images=3)
    with pytest.raises(ValueError):
        dataset.Dataset(str(tmp_path / 'nowhere'))
    arr = dataset.Dataset.from_arrays({'y': rgb[:4]}, {'y': rgb[4:, :32, :32]})
    assert arr.loaded_data == 'rgb' and arr.next_training_batch(0, 4, 48, 'flat').shape == (4, 48, 48, 3)
    with pytest.raises(ValueError):
        dataset.Dataset.from_arrays({'y': rgb[:4].astype(np.float32)}, {'y': rgb[4:]})
    with pytest.raises(RuntimeError):
        dataset.DeviceDataset(arr, device='cpu')
    # helpers/dataset.py:247-255: the batch streams (tf.data pipelines there) - re-iterable, one pass = count // batch batches
    pipe = arr.get_training_pipeline(2, 48, discard=None)
    for _ in range(2):
        passes = [b for b in pipe]
        assert len(passes) == 2 and all(b.shape == (2, 48, 48, 3) for b in passes)
    assert [b.shape for b in arr.get_validation_pipeline(1)] == [(1, 32, 32, 3)] * arr.count_validation


def _run_dp_workers(world):
    import torch.multiprocessing as mp
    from dp_worker import dp_worker
    res, procs = collect_from_workers(
        lambda ctx, port, q: [ctx.Process(target=dp_worker, args=(r, world, port, q)) for r in range(world)], world, 180)
    assert all(p.exitcode == 0 for p in procs)
    return sorted(res)


@pytest.mark.parametrize('world', [2, 8])
def test_data_parallel_plumbing_gloo(world):
    """parallel.py over gloo on CPU, 2 ranks and the 8 ranks of one MI355X node: three gradient buckets in flight (sum
    all-reduce), NaN-flag OR, contiguous shards of the global batch, rank 0's augmentation draws broadcast, and the per-rank
    random streams (awgn noise / dropout masks: every rank differs, rank 0 reproduces the single-process stream)."""
    import torch
    res = _run_dp_workers(world)
    total = float(sum(range(1, world + 1)))
    for r, out in enumerate(res):
        assert out[0] == r and out[1] == [total, total, total]     # every bucket of the flat gradient buffer is summed
        assert out[2] == 1                                        # NaN flag is OR-reduced
        assert out[3] == list(range(4 * r, 4 * r + 4))
        assert out[4] == [0.25, 7.0]                              # rank 0's augmentation draws reach every rank
    streams = [tuple(out[5]) for out in res]
    assert len(set(streams)) == world                             # no two ranks draw the same noise
    single = torch.randn((6,), generator=torch.Generator().manual_seed(5)).tolist()
    assert list(streams[0]) == single                             # rank 0 = the single-process stream


def test_forced_collectives_at_world_size_one():
    """parallel.force_collectives(): a one-rank group whose collectives really run (the hook the nccl world-1 GPU test and
    bench.py's dp1_nccl leg use) - sum over one rank is the identity, the world size stays 1 (Adam scale 1)."""
    import torch.multiprocessing as mp
    from dp_worker import forced_world1_worker
    ((ok, flat, flag, drawn, world),), procs = collect_from_workers(
        lambda ctx, port, q: [ctx.Process(target=forced_world1_worker, args=(q,))], 1, 120)
    assert procs[0].exitcode == 0 and ok
    assert flat == [float(i) for i in range(10)] and flag == 1 and drawn == [0.5, 2.0] and world == 1


def test_forced_one_rank_group_picks_another_port_when_its_pick_is_taken(monkeypatch):
    """parallel.init_from_env for a forced one-rank group chooses the rendezvous port itself; between probing a free number and the
    store listening on it another process can take it (EADDRINUSE, seen on a GPU box during test_data_parallel_step_nccl_world1):
    it picks again.  A port the LAUNCHER set is never second-guessed, and any other error passes through."""
    calls = []

    def fake_init(backend=None):
        calls.append(os.environ['MASTER_PORT'])
        if len(calls) < 3:
            raise RuntimeError('The server socket has failed to listen on any local network address. port: {}, useIpv6: false, '
                               'code: -98, name: EADDRINUSE, message: address already in use'.format(calls[-1]))

    for k in ('MASTER_PORT', 'RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(parallel.dist, 'init_process_group', fake_init)
    monkeypatch.setattr(parallel, '_FORCE', True)
    assert parallel.init_from_env('gloo') == 1
    assert len(calls) == 3 and os.environ['MASTER_PORT'] == calls[-1]
    # a port given by the launcher: one attempt, the error is the caller's
    calls.clear()
    monkeypatch.setenv('MASTER_PORT', '29500')
    with pytest.raises(RuntimeError):
        parallel.init_from_env('gloo')
    assert calls == ['29500']
    # an error that is not about the port: no retry
    monkeypatch.delenv('MASTER_PORT')
    calls.clear()
    monkeypatch.setattr(parallel.dist, 'init_process_group', lambda backend=None: (_ for _ in ()).throw(ValueError('bad backend')))
    with pytest.raises(ValueError):
        parallel.init_from_env('gloo')
    assert 'MASTER_PORT' not in os.environ
    for k in ('RANK', 'WORLD_SIZE'):
        os.environ.pop(k, None)
