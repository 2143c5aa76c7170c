#!/usr/bin/env python
"""GPU box: randomised stress of the device lattice against the C oracle -- many sizes (1 .. 60 000 points,
ragged pairs), distributions (uniform frustum, surface patches, tight clusters, duplicated points, lines,
huge / tiny coordinates) and scale maps; every table of every level must match bit for bit.
    python tests/stress/stress_lattice.py [--cases 200] [--seed 0]
"""
import argparse, os, sys, time, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hplflownet_amd as H
from hplflownet_amd.synthetic import SCALES_FILTER_MAP
from oracle import lattice_oracle as LO


def cloud(rng, n, kind):
    p = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    p[:, 2] = rng.uniform(1.5, 35, n)
    if kind == 'surface':
        u, v = rng.uniform(-6, 6, n), rng.uniform(-3, 3, n)
        p = np.stack([u, v, 10 + 0.3 * np.sin(u) + 0.2 * v], 1).astype(np.float32)
    elif kind == 'cluster':
        c = rng.uniform(-5, 5, (max(1, n // 200), 3))
        p = (c[rng.randint(0, len(c), n)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    elif kind == 'dup':
        p = p[rng.randint(0, max(1, n // 5), n)]
    elif kind == 'line':
        p[:, :2] = 0
    elif kind == 'far':
        p *= 40
    elif kind == 'tiny':
        p *= 1e-3
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--native', action='store_true', help='the native builder (fused driver, staged fallback on overflow) instead of the Python stage driver')
    a = ap.parse_args()
    rng = np.random.RandomState(a.seed)
    maps = [SCALES_FILTER_MAP, SCALES_FILTER_MAP[:5], [[4., 1, -1, -1], [1.5, 1, 1, 1], [0.7, 1, 1, 1], [0.2, 1, 1, 1]]]
    kinds = ['frustum', 'surface', 'cluster', 'dup', 'line', 'far', 'tiny']
    t0 = time.time()
    bad = 0
    gens = {}
    for case in range(a.cases):
        kind = kinds[case % len(kinds)]
        big = case % 25 == 24
        n1 = int(rng.randint(20000, 60000)) if big else int(np.exp(rng.uniform(0, np.log(6000))))
        n2 = max(1, int(n1 * rng.uniform(0.5, 1.2)))
        sfm = maps[case % len(maps)]
        p1, p2 = cloud(rng, n1, kind), cloud(rng, n2, kind)
        gen = gens.setdefault(id(sfm), H.GenerateDataUnsymmetric(types.SimpleNamespace(dim=3, scales_filter_map=sfm), device='cuda'))
        if a.native:         # one builder per scale map: the vertex bounds adapt over the cases (and overflow now and then)
            t1, t2 = (torch.from_numpy(np.ascontiguousarray(p.T)).cuda() for p in (p1, p2))
            lat = gen.build_native(t1, t2)
            torch.cuda.synchronize()
        else:
            _, _, _, lat = gen([p1, p2, np.zeros_like(p1)])
        gd = LO.generate_data(p1, p2, sfm)
        for l, (x, y) in enumerate(zip(H.to_reference_format(lat), gd)):
            for k in y:
                vx = x[k].cpu().numpy() if torch.is_tensor(x[k]) else x[k]
                if not np.array_equal(np.asarray(vx), np.asarray(y[k])):
                    bad += 1
                    print('MISMATCH case %d kind %s n=(%d,%d) level %d key %s' % (case, kind, n1, n2, l, k))
        if case % 20 == 19:
            print('case %d ok so far (%d mismatches) %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
    if a.native:
        print('native builders: fused %s, staged fallbacks %s' % ([g.native_builder().fused for g in gens.values()], [g.native_builder().fallbacks for g in gens.values()]))
    print('DONE %d cases, %d mismatching tables, %.0f s' % (a.cases, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
