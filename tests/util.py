"""Shared helpers for the parity tests (oracle on CPU float64 vs HIP kernels on the GPU)."""
import numpy as np
import torch


def natural_images(n, h, w, seed=0):
    """Natural-image-like synthetic RGB in [0,1]: low-pass blocks + ramp + noise, quantised to k/255."""
    rng = np.random.default_rng(seed)
    base = rng.random((n, h // 8 + 1, w // 8 + 1, 3))
    img = np.kron(base, np.ones((1, 8, 8, 1)))[:, :h, :w, :]
    ramp = np.linspace(0, 1, w)[None, None, :, None]
    img = 0.7 * img + 0.3 * ramp + rng.normal(0, 0.03, size=(n, h, w, 3))
    return (np.round(np.clip(img, 0, 1) * 255) / 255).astype(np.float32)


def bayer_from_rgb(rgb):
    """Position-ordered Bayer stack (N,h/2,w/2,4) of an RGB batch on a GBRG mosaic: planes taken at (0,0), (0,1), (1,0), (1,1)
    = G, B, R, G.  A synthetic RAW input for the learned pipelines (which do not care about the plane order); the parity
    tolerances of the workflow tests were tuned on it.  The reference's own order is stack_bayer() below."""
    g1 = rgb[:, 0::2, 0::2, 1]
    b = rgb[:, 0::2, 1::2, 2]
    r = rgb[:, 1::2, 0::2, 0]
    g2 = rgb[:, 1::2, 1::2, 1]
    return np.stack([g1, b, r, g2], axis=-1).astype(np.float32)


def stack_bayer(rgb):
    """RGGB-ordered stack of a GBRG mosaic like the reference's helpers/raw.py:204-225 `stack_bayer(image, 'GBRG')`: planes
    (R, G1, G2, B) taken at (1,0), (0,0), (1,1), (0,1) - what upsampling_kernel('gbrg') puts back in place."""
    return np.stack([rgb[:, 1::2, 0::2, 0], rgb[:, 0::2, 0::2, 1], rgb[:, 1::2, 1::2, 1], rgb[:, 0::2, 1::2, 2]],
                    axis=-1).astype(np.float32)


def to64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0, 0.0
    d = np.abs(a - b)
    return float(d.max()), float(d.max() / (np.abs(b).max() + 1e-30))


def assert_close(a, b, atol, rtol_max=None, what=''):
    """|a-b| <= atol  OR  (optionally) max|a-b| <= rtol_max * max|b| (gradient tensors have arbitrary scale)."""
    mx, rel = err(a, b)
    ok = mx <= atol or (rtol_max is not None and rel <= rtol_max)
    assert ok, '{}: max abs err {:.3e}, rel-to-max {:.3e} (atol {}, rtol_max {})'.format(what, mx, rel, atol, rtol_max)
    return mx, rel



def scene_images(n, h, w, seed=0, device='cpu'):
    """Synthetic photographs with the statistics the channel needs to LEARN on (bench.py's trained-parity leg,
    tools/train_parity.py): smooth colour gradients + a few soft-edged occluders + band-limited texture that is mostly
    luminance (the cross-channel correlation a demosaicer exploits) + a little sensor noise, quantised to k/255.
    Generated with torch on `device` (seeded per device type: a CPU and a GPU run draw different images) -> float32 tensor
    (n, h, w, 3).  natural_images() above stays the input of the parity tests (their tolerances were tuned on it)."""
    import torch.nn.functional as F
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(int(seed))
    rnd = lambda *s: torch.rand(*s, generator=g, device=dev)
    nrm = lambda *s: torch.randn(*s, generator=g, device=dev)

    def blur(x, sigma):                                  # x (n, c, h, w): separable gaussian, reflect borders
        r = int(3 * sigma + 0.5)
        k = torch.exp(-0.5 * (torch.arange(-r, r + 1, device=dev, dtype=torch.float32) / sigma) ** 2)
        k = k / k.sum()
        c = x.shape[1]
        x = F.conv2d(F.pad(x, (r, r, 0, 0), mode='reflect'), k.view(1, 1, 1, -1).repeat(c, 1, 1, 1), groups=c)
        return F.conv2d(F.pad(x, (0, 0, r, r), mode='reflect'), k.view(1, 1, -1, 1).repeat(c, 1, 1, 1), groups=c)

    img = F.interpolate(rnd(n, 3, h // 32 + 2, w // 32 + 2), scale_factor=32, mode='bilinear', align_corners=False)
    img = 0.15 + 0.7 * img[:, :, 16:16 + h, 16:16 + w]
    yy = torch.arange(h, device=dev, dtype=torch.float32).view(1, h, 1)
    xx = torch.arange(w, device=dev, dtype=torch.float32).view(1, 1, w)
    for _ in range(3):                                   # occluders: half planes and discs with ~1 px soft edges
        th = rnd(n, 1, 1) * (2 * np.pi)
        cx, cy = rnd(n, 1, 1) * w, rnd(n, 1, 1) * h
        disc = rnd(n, 1, 1) < 0.5
        rad = (0.1 + 0.3 * rnd(n, 1, 1)) * min(h, w)
        dx, dy = xx - cx, yy - cy
        dist = torch.where(disc, rad - torch.sqrt(dx * dx + dy * dy), torch.cos(th) * dx + torch.sin(th) * dy)
        mask = torch.sigmoid(dist / 0.7)
        delta = (rnd(n, 3, 1, 1) - 0.5) * 0.7
        img = img + mask[:, None] * delta
    amp = 0.02 + 0.06 * rnd(n, 1, 1, 1)
    lum = blur(nrm(n, 1, h, w), 1.2) * 3.0
    chroma = blur(nrm(n, 3, h, w), 2.0) * 5.0
    img = img + amp * lum + 0.3 * amp * chroma + 0.004 * nrm(n, 3, h, w)
    img = torch.round(img.clamp(0, 1) * 255) / 255
    return img.permute(0, 2, 3, 1).contiguous()


def bayer_from_rgb_t(rgb):
    """bayer_from_rgb() for a torch tensor (N,h,w,3) on any device."""
    return torch.stack([rgb[:, 0::2, 0::2, 1], rgb[:, 0::2, 1::2, 2], rgb[:, 1::2, 0::2, 0], rgb[:, 1::2, 1::2, 1]],
                       dim=-1).contiguous()


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def collect_from_workers(make_procs, n_results, timeout, attempts=3):
    """Start the processes make_procs(port, queue) returns and collect n_results answers from the queue.  A worker that dies
    without answering ends the wait at once (not after `timeout`), and - because the rendezvous port is probed, released and only
    then listened on by rank 0, so another process can take it in between - the whole group is started again on a fresh port, up to
    `attempts` times.  Returns (results, procs) with every process joined."""
    import queue
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    last = None
    for _ in range(attempts):
        q = ctx.Queue()
        procs = make_procs(ctx, free_port(), q)
        for p in procs:
            p.start()
        res, deadline = [], time.monotonic() + timeout
        while len(res) < n_results:
            try:
                res.append(q.get(timeout=1.0))
            except queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    try:                                                   # an answer put just before the exit
                        res.append(q.get(timeout=0.5))
                        continue
                    except queue.Empty:
                        break
                if time.monotonic() > deadline:
                    break
        if len(res) == n_results:
            for p in procs:
                p.join(timeout=120)
            return res, procs
        last = [p.exitcode for p in procs]
        for p in procs:                                                    # exactly the processes started here
            if p.is_alive():
                p.kill()
            p.join(timeout=30)
    raise RuntimeError('workers did not answer in {} attempt(s); exit codes of the last one: {}'.format(attempts, last))
