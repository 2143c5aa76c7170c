"""GPU: the fused lattice driver (csrc/lattice_fused.hip: the whole 7-level build enqueued at once, vertex counts kept on
the device, ONE read-back per pair) against the staged driver (one read-back per level) and the C oracle: every table
of every level bit-identical -- integer tables, CSR, row orders and per-tile index tables included -- on the sizes of
the BASELINE configs, ragged / tiny / degenerate clouds, under bounds that overflow (staged fallback) and in the
pipelined producer-thread form the bench uses."""
import types

import numpy as np
import pytest
import torch

from hplflownet_amd.synthetic import SCALES_FILTER_MAP, fill_module_, surface_pair, synthetic_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def make_gen(nsc, fused, monkeypatch, wide=True):
    import hplflownet_amd as H
    monkeypatch.setenv('HPL_LATTICE_FUSED', '1' if fused else '0')
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:nsc], evaluate=True, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    cls = H.HPLFlowNet if nsc == 7 else H.HPLFlowNetShallow
    m = cls(args)
    hint = m.lattice_hint() if wide else None
    gen = H.GenerateDataUnsymmetric(args, device=DEV, wide_up=hint)
    nb = gen.native_builder()
    assert nb.fused == fused
    return gen


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(DEV)


def assert_same_lattice(a, b, what=''):
    """a, b: NativeLattice -- every table the forward, the backward or the wire format can see."""
    import hplflownet_amd as H
    assert a.H == b.H, what
    for L, (x, y) in enumerate(zip(H.to_reference_format(a), H.to_reference_format(b))):
        for k in y:
            assert (torch.equal(x[k], y[k]) if torch.is_tensor(y[k]) else x[k] == y[k]), (what, L, k)
    for L, (x, y) in enumerate(zip(a.levels, b.levels)):
        assert all(torch.equal(p, q) for p, q in zip(x.pair.csr(), y.pair.csr())), (what, L, 'csr')
        assert torch.equal(x.emg_pair, y.emg_pair), (what, L)
        for nm, tx, ty in (('pair', x.blur.pair, y.blur.pair), ('up', x.blur[0], y.blur[0])):
            px, py = getattr(tx, '_perm', False), getattr(ty, '_perm', False)
            assert (px is None or px is False) == (py is None or py is False), (what, L, nm)
            if py is not None and py is not False:
                assert torch.equal(px, py), (what, L, nm, 'perm')
                for u, v in zip(tx._perm_tiles, ty._perm_tiles):
                    assert torch.equal(u, v), (what, L, nm, 'tiles')
            gx, gy = getattr(tx, '_groups', None), getattr(ty, '_groups', None)
            assert (gx is None) == (gy is None), (what, L, nm, 'groups')
            if gy:
                for (f0, f1, p), (g0, g1, q) in zip(gx, gy):
                    assert (f0, f1) == (g0, g1) and torch.equal(p, q), (what, L, nm, 'group perm')
                for (u0, u1), (v0, v1) in zip(tx._group_tiles, ty._group_tiles):
                    assert torch.equal(u0, v0) and torch.equal(u1, v1), (what, L, nm, 'group tiles')
        if y.corr2 is not None:
            assert torch.equal(x.corr2.t, y.corr2.t), (what, L, 'corr2')


CASES = [(7, 'frustum', 8192, 8192), (7, 'surface', 8192, 8000), (5, 'frustum', 4096, 4096), (7, 'frustum', 50, 37),
         (7, 'frustum', 16384, 16384), (7, 'frustum', 1, 1), (7, 'frustum', 2048, 300), (7, 'frustum', 32768, 32768)]


@pytest.mark.parametrize('nsc,kind,n1,n2', CASES)
def test_fused_equals_staged_bit_exact(nsc, kind, n1, n2, monkeypatch):
    gen_f = make_gen(nsc, True, monkeypatch)
    gen_s = make_gen(nsc, False, monkeypatch)
    pc1, pc2, _ = (surface_pair if kind == 'surface' else synthetic_pair)(max(n1, n2), 6)
    t1, t2 = dev(pc1[:n1]), dev(pc2[:n2])
    a = gen_f.build_native(t1, t2)
    b = gen_s.build_native(t1, t2)
    torch.cuda.synchronize()
    assert gen_f.native_builder().fallbacks == 0
    assert gen_f.native_builder().launches <= 60, gen_f.native_builder().launches      # staged: ~260
    assert_same_lattice(a, b, 'first build (default bounds)')
    # second build: the bounds now follow the counts seen (another arena layout), a different pair of the same size
    pc1b, pc2b, _ = (surface_pair if kind == 'surface' else synthetic_pair)(max(n1, n2), 7)
    t1b, t2b = dev(pc1b[:n1]), dev(pc2b[:n2])
    assert any(gen_f.native_builder().bounds)
    a2, b2 = gen_f.build_native(t1b, t2b), gen_s.build_native(t1b, t2b)
    torch.cuda.synchronize()
    assert_same_lattice(a2, b2, 'second build (observed bounds)')


def test_fused_equals_oracle():
    """straight against the C oracle (itself pinned to the reference's golden vectors), N = 1024 frustum"""
    import hplflownet_amd as H
    from oracle import lattice_oracle
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP, evaluate=True, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    gen = H.GenerateDataUnsymmetric(args, device=DEV)
    assert gen.native_builder().fused
    pc1, pc2, _ = synthetic_pair(1024, 0)
    lat = gen.build_native(dev(pc1), dev(pc2))
    gd = lattice_oracle.generate_data(pc1, pc2, SCALES_FILTER_MAP)
    for L, (x, d) in enumerate(zip(H.to_reference_format(lat), gd)):
        for k, v in d.items():
            got = x[k].cpu().numpy() if torch.is_tensor(x[k]) else x[k]
            assert np.array_equal(np.asarray(got).reshape(-1), np.asarray(v).reshape(-1)), (L, k)


def test_degenerate_clouds(monkeypatch):
    """all points identical; duplicates; a cloud of one point next to a big one"""
    gen_f = make_gen(7, True, monkeypatch)
    gen_s = make_gen(7, False, monkeypatch)
    pc1, pc2, _ = synthetic_pair(600, 3)
    same = np.repeat(pc1[:1], 300, axis=0)
    dup = np.concatenate([pc1[:200], pc1[:200], pc1[100:300]], axis=0)
    for what, (u, v) in {'identical': (same, pc2[:300]), 'duplicates': (dup, dup[::-1].copy()), 'one vs many': (pc1[:1], pc2)}.items():
        a, b = gen_f.build_native(dev(u), dev(v)), gen_s.build_native(dev(u), dev(v))
        torch.cuda.synchronize()
        assert_same_lattice(a, b, what)


def test_overflowing_bounds_fall_back_to_the_staged_driver(monkeypatch):
    gen_f = make_gen(7, True, monkeypatch)
    gen_s = make_gen(7, False, monkeypatch)
    nb = gen_f.native_builder()
    pc1, pc2, _ = synthetic_pair(3000, 11)
    t1, t2 = dev(pc1), dev(pc2)
    for bad_level in (0, 2, 6):
        nb.bounds = [0] * 8
        nb.seen = [0] * 8
        nb.bounds[bad_level] = 16                      # far below the real vertex count of that level
        before = nb.fallbacks
        a = gen_f.build_native(t1, t2)
        b = gen_s.build_native(t1, t2)
        torch.cuda.synchronize()
        assert nb.fallbacks == before + 1
        assert_same_lattice(a, b, 'overflow at level %d' % bad_level)
        assert nb.bounds[bad_level] >= 2 * max(a.H[bad_level])      # the next build fits
        a = gen_f.build_native(t1, t2)
        torch.cuda.synchronize()
        assert nb.fallbacks == before + 1
        assert_same_lattice(a, b, 'after the overflow at level %d' % bad_level)


def test_fused_pipeline_threaded_and_forward(monkeypatch):
    """the bench's form: producer thread, several pairs in flight on a side stream, forwards on the lattices"""
    import hplflownet_amd as H
    from hplflownet_amd.lattice import LatticePipeline
    gen = make_gen(5, True, monkeypatch)
    args = types.SimpleNamespace(dim=3, scales_filter_map=SCALES_FILTER_MAP[:5], evaluate=True, use_leaky=True,
                                 bcn_use_bias=True, bcn_use_norm=True, last_relu=False, DEVICE='cuda')
    m = H.HPLFlowNetShallow(args)
    fill_module_(m, 1.0, 'hash')
    m = m.to(DEV).eval()
    pairs = []
    for s_, n in enumerate([900, 300, 1500, 64, 700, 4096, 5]):
        p1, p2, _ = synthetic_pair(n, 50 + s_)
        pairs.append((dev(p1), dev(p2)))
    side = torch.cuda.Stream()
    for threaded in (False, True):
        pipe = LatticePipeline(gen, lambda i: pairs[i], 0, len(pairs), depth=3, stream=side, native=True, threaded=threaded)
        outs = []
        with torch.no_grad():
            for want in range(len(pairs)):
                (i, item), lat, ev = pipe.get()
                assert i == want
                torch.cuda.current_stream().wait_event(ev)
                outs.append(m(item[0][None], item[1][None], lat).clone())
            for i, (a, b) in enumerate(pairs):
                assert torch.equal(outs[i], m(a[None], b[None], gen.build(a, b)))
    assert gen.native_builder().fallbacks == 0


def test_rebuild_is_deterministic(monkeypatch):
    gen = make_gen(7, True, monkeypatch)
    pc1, pc2, _ = synthetic_pair(8192, 1)
    t1, t2 = dev(pc1), dev(pc2)
    a = gen.build_native(t1, t2)
    for _ in range(3):
        b = gen.build_native(t1, t2)
        torch.cuda.synchronize()
        assert_same_lattice(a, b, 'rebuild')


def test_abandoned_pipeline_stops_its_producer(monkeypatch):
    from hplflownet_amd.lattice import LatticePipeline
    gen = make_gen(5, True, monkeypatch)
    pairs = []
    for s_ in range(6):
        p1, p2, _ = synthetic_pair(500, 70 + s_)
        pairs.append((dev(p1), dev(p2)))
    side = torch.cuda.Stream()
    pipe = LatticePipeline(gen, lambda i: pairs[i], 0, len(pairs), depth=2, stream=side, native=True, threaded=True)
    (i, _), lat, ev = pipe.get()
    assert i == 0
    th = pipe._thread
    assert th is not None and th.is_alive()          # blocked: the queue is full, four pairs are still to come
    pipe.close()
    assert not th.is_alive()
    torch.cuda.synchronize()


def test_builder_protocol_errors(monkeypatch):
    """misuse of the hpl_lattice_* calls is reported, not executed: bounds while a build is in flight, a too small arena
    (HPL_ENOMEM with the size the fused layout needs known up front), advance without a build"""
    import ctypes
    from hplflownet_amd import _lib
    gen = make_gen(7, True, monkeypatch)
    nb = gen.native_builder()
    L = nb.lib
    h = nb.acquire()
    try:
        d = ctypes.c_int(0)
        assert L.hpl_lattice_advance(h, ctypes.byref(d)) != 0 and b'no build' in L.hpl_last_error()
        pc1, pc2, _ = synthetic_pair(2000, 5)
        t1, t2 = dev(pc1), dev(pc2)
        need = int(L.hpl_lattice_arena_bytes(h, 2000, 2000))
        assert need > 0
        small = torch.empty(need // 2, dtype=torch.uint8, device=DEV)
        assert L.hpl_lattice_begin(h, t1.data_ptr(), t2.data_ptr(), 2000, 2000, small.data_ptr(), small.numel(), _lib.stream()) == -4
        arena = torch.empty(need + 4096, dtype=torch.uint8, device=DEV)
        assert L.hpl_lattice_begin(h, t1.data_ptr(), t2.data_ptr(), 2000, 2000, arena.data_ptr(), arena.numel(), _lib.stream()) == 0
        bounds = (ctypes.c_int64 * 8)(*([1 << 20] * 8))
        assert L.hpl_lattice_set_bounds(h, bounds) != 0 and b'in progress' in L.hpl_last_error()
        while not d.value:
            assert L.hpl_lattice_advance(h, ctypes.byref(d)) == 0
        assert L.hpl_lattice_set_bounds(h, bounds) == 0                   # finished: allowed again
        st = (ctypes.c_int32 * 3)()
        assert L.hpl_lattice_stats(h, st) == 0 and st[0] <= 60 and st[1] == 1
        assert int(L.hpl_lattice_arena_bytes(h, 2000, 2000)) > need        # the bounds just set are larger than the defaults' cascade
        torch.cuda.synchronize()
    finally:
        nb.release(h)
