#!/usr/bin/env python3
"""Identity of the kernel sources a profile was taken on.

csrc_sha16() = first 16 hex digits of the sha256 over every file under neural-imaging_amd/csrc (*.hip, *.h, Makefile) and
include/nimg.h, in name order.  The counter-collection tools (tools/pmc_ring.sh, tools/pmc_step_total.py) write it into the
JSON they produce; bench.py prints a traffic figure from such a file ONLY if the stamp equals the sources it is running on
(no figures from profiles that predate the kernels).  `python tools/src_stamp.py <json>...` run in the
git checkout adds `git_head` (the GPU box has no .git) and reports whether each file is current.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, 'neural-imaging_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h')) + [os.path.join(d, 'Makefile'),
                   os.path.join(root, 'include', 'nimg.h')])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_if_current(path, root=ROOT):
    """-> (dict or None, source): the profile JSON if its csrc_sha16 stamp equals the current sources, else None."""
    src = {'file': os.path.relpath(path, root)}
    try:
        with open(path) as f:
            data = json.load(f)
    except (OSError, ValueError):
        return None, dict(src, status='missing')
    src.update(csrc_sha16=data.get('csrc_sha16'), git_head=data.get('git_head'))
    if data.get('csrc_sha16') != csrc_sha16(root):
        return None, dict(src, status='stale: taken on other kernel sources than this run ({})'.format(csrc_sha16(root)))
    return data, dict(src, status='current')


if __name__ == '__main__':
    try:
        head = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ''
    cur = csrc_sha16()
    print('csrc_sha16', cur)
    for p in sys.argv[1:]:
        with open(p) as f:
            d = json.load(f)
        if head and d.get('csrc_sha16') == cur and not d.get('git_head'):
            d['git_head'] = head + ' (+ working tree at collection time)'
            with open(p, 'w') as f:
                json.dump(d, f, indent=1)
        print(p, 'current' if d.get('csrc_sha16') == cur else 'STALE ({})'.format(d.get('csrc_sha16')))
