"""Diagnostic: per-tensor cosine between the float32 and the throughput mode of the full channel with the learned codec
(continuous latent, no entropy term), for several loss weightings."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
importlib.import_module('neural-imaging_amd')
from neural_imaging_amd import ops
from neural_imaging_amd.models import compression
from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
from util import natural_images, bayer_from_rgb

dev = torch.device('cuda', 0)
rgb = natural_images(2, 128, 128, seed=9)
raw = bayer_from_rgb(rgb)
cos = lambda a, b: float(a.ravel() @ b.ravel() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
G = lambda m: {k: v.detach().cpu().numpy().astype(np.float64) for k, v in m._model.g.items()}
for manips in (['sharpen:1', 'resample:50', 'gaussian:0.83'], ['resample:50'], ['gaussian:0.83'], ['sharpen:1']):
    for lam_nip, lam_dcn in ((0.1, 0.1), (0.1, 0.0), (0.0, 0.1)):
        res = {}
        for mode in ('f32', 'bf16'):
            ops.set_compute(mode)
            dcn = compression.TwitterDCN(patch_size=128, rounding='identity', entropy_weight=0, device=dev)
            dist = {'downsampling': 'none', 'compression': 'dcn', 'compression_params': {'model': dcn}}
            wf = ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip', 'dcn'},
                                            raw_patch_size=64, device=dev)
            loss, parts = wf.training_step(raw, rgb, lambda_nip=lam_nip, lambda_dcn=lam_dcn, learning_rate=1e-4)
            res[mode] = ({k: float(v) for k, v in parts.items()}, G(wf.nip), G(dcn), G(wf.fan))
        ops.set_compute('f32')
        line = []
        for name, gi in (('nip', 1), ('dcn', 2), ('fan', 3)):
            cs = sorted((cos(a, res['bf16'][gi][k]), k) for k, a in res['f32'][gi].items() if a.size >= 16 and np.linalg.norm(a) > 0)
            norms = (sum(np.linalg.norm(a) ** 2 for a in res['f32'][gi].values()) ** 0.5, sum(np.linalg.norm(a) ** 2 for a in res['bf16'][gi].values()) ** 0.5)
            line.append('%s min %.3f (%s) med %.3f |g| %.3e/%.3e' % (name, cs[0][0], cs[0][1], cs[len(cs) // 2][0], norms[0], norms[1]))
        print(manips, 'lam', lam_nip, lam_dcn, res['f32'][0], '\n   ', ' | '.join(line), flush=True)
