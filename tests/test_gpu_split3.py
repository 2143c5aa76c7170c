"""The split-operand gather-GEMM (csrc/gconv3.hip) against float64 and against the fp32-MFMA kernel, in both of its forms:
every fp32 operand as an exact bf16 triple on the bf16 matrix pipe (HPL_MATH=bf16x3: 6 partial products), and as a scaled
fp16 pair on the fp16 pipe (the default since round 5: 3 partial products, operands good to 2^-22).  The claim pinned here:
the error against the exact result is of the fp32 rounding class -- triples: not larger than the fp32 kernel's own; pairs:
largest error not larger than the fp32 kernel's, mean error within 1.4 x of it (a sequential fp32 dot product measures
1.4 x) -- on dense and gathered launches, with and without row orders / tile tables, including the tails (partial tiles,
partial slices, absent rows, taps straddling slices)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ref64(A, nbr, M, C, F, Wt, N, bias=None, leaky=None):
    A64, W64 = A.double(), Wt[:F * C, :N].double()
    y = torch.zeros((M, N), dtype=torch.float64, device=A.device)
    for f in range(F):
        if nbr is None:
            rows = A64[:M, :C]
        else:
            idx = nbr[f].long()
            rows = torch.where((idx >= 0)[:, None], A64[idx.clamp(min=0), :C], torch.zeros((), dtype=torch.float64, device=A.device))
        y += rows @ W64[f * C:(f + 1) * C]
    if bias is not None:
        y += bias.double()
    if leaky is not None:
        y = torch.where(y > 0, y, leaky * y)
    return y


def _table(M, rows_a, F, density, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    nbr = torch.randint(0, rows_a, (F, M), generator=g, dtype=torch.int32)
    nbr[torch.rand((F, M), generator=g) > density] = -1
    nbr[0] = torch.arange(M, dtype=torch.int32) % rows_a
    return nbr.to(DEV)


@pytest.mark.parametrize('M,C,F,N,density,order', [
    (8192, 64, 1, 256, 1.0, None),            # dense GEMM, K = 64
    (16500, 580, 8, 1024, 0.45, 'tiles'),     # the dominant launch's shape class: tap group of bcn1_, 128-row tile tables
    (16400, 324, 7, 512, 0.7, 'perm'),        # second group of bcn2_: K = 2268 is not a multiple of 32, prologue without tables
    (16385, 100, 15, 256, 0.5, 'tiles'),      # C % 8 = 4: 8-blocks straddle taps; one row in the last tile
    (16500, 36, 3, 320, 0.9, None),           # N = 320: 2.5 column tiles of 128
    (16600, 64, 8, 512, 0.8, 'tiles'),        # N = 512: the 256-wide tile variant
    (16500, 1024, 15, 580, 0.45, 'perm'),     # the data gradient of bcn1_ (mirrored gather): N = 580 = 4.5 tiles of 128
    (9433, 388, 15, 256, 0.8, 'splitk'),      # bcn3_: 74 row tiles -> split over K into 3 shares, partial tiles + fixed-order sum
    (8200, 132, 15, 256, 0.6, 'splitk'),      # 62 slices over 3 shares: a last share that is longer; C % 8 = 4
])
@pytest.mark.parametrize('planes', [2, 3])
def test_split3_matches_float64_like_the_fp32_kernel(M, C, F, N, density, order, planes):
    from hplflownet_amd import ops
    torch.manual_seed(M + C)
    rows_a = M if F == 1 else M + 37
    A = torch.randn(rows_a, C, device=DEV) * torch.exp(torch.randn(rows_a, 1, device=DEV))      # rows of very different scale
    k_rows = ops.round_up(F * C, 32)
    Wt = torch.zeros(k_rows, ops.round_up(N, 4), device=DEV)
    Wt[:F * C, :N] = torch.randn(F * C, N, device=DEV) / (F * C) ** 0.5
    bias = torch.randn(N, device=DEV)
    nbr = _table(M, rows_a, F, density, 3) if F > 1 else None
    perm = tiles = None
    if order in ('perm', 'tiles'):
        perm = ops.tap_order(nbr)
        if order == 'tiles':
            tiles = ops.tile_index(nbr, perm, BM=128)
    W3 = ops.weight_split3(Wt, planes=planes)
    if planes == 3:         # the planes are an exact decomposition of the image
        pl = W3.planes.view(torch.bfloat16).view(3, k_rows // 8, Wt.shape[1], 8).float()
        assert torch.equal(pl.sum(0).permute(0, 2, 1).reshape(k_rows, Wt.shape[1]), Wt)
    else:                   # hi + lo = w * s up to 2^-22 |w s| (lo below fp16's normal range: 2^-25 absolute), s = 2^k, amax * s in [2^14, 2^15)
        amax = float(W3.amax)
        assert amax == float(Wt.abs().max())
        sc = 2.0 ** (14 - int(np.floor(np.log2(amax))))
        pl = W3.planes.view(torch.float16).view(2, k_rows // 8, Wt.shape[1], 8).double()
        back = pl.sum(0).permute(0, 2, 1).reshape(k_rows, Wt.shape[1])
        ws = Wt.double() * sc
        assert float(pl[0].abs().max()) < 2.0 ** 15 + 16
        assert bool(((back - ws).abs() <= ws.abs() * 2.0 ** -22 + 2.0 ** -25).all())
    kw = dict(bias=bias, act=ops.ACT_LEAKY, row_perm=perm, tiles=tiles, split_k=order == 'splitk')
    y3 = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, **kw)
    y1 = ops.gconv_raw(A, nbr, M, C, F, Wt, N, **kw)
    assert not torch.equal(y3, y1), 'the split-operand kernel did not take this launch'
    ref = _ref64(A, nbr, M, C, F, Wt, N, bias, ops.LEAKY_RATE)
    scale = float(ref.abs().max())
    e3, e1 = float((y3.double() - ref).abs().max()), float((y1.double() - ref).abs().max())
    # error measured in units of the magnitude sum |a||w| of each output (what bounds an fp32 dot product's rounding)
    mag = _ref64(A.abs(), nbr, M, C, F, Wt.abs(), N) + bias.abs().double()
    r3 = float(((y3.double() - ref).abs() / mag).max()), float(((y3.double() - ref).abs() / mag).mean())
    r1 = float(((y1.double() - ref).abs() / mag).max()), float(((y1.double() - ref).abs() / mag).mean())
    print('M=%d K=%d N=%d: max|err| split3 %.3g fp32 %.3g (scale %.3g); err / sum|a||w|: split3 max %.3g mean %.3g, '
          'fp32 max %.3g mean %.3g' % (M, F * C, N, e3, e1, scale, r3[0], r3[1], r1[0], r1[1]))
    # fp32 rounding class in absolute terms (a few 2^-24 of the magnitude sum; an fp32 fma chain of K terms measures
    # 3e-7 .. 7e-7 here), and not worse than the fp32-MFMA kernel on the same launch: measured 0.6x (max) / 0.85x (mean)
    assert r3[0] < 1.5e-6 and r3[1] < 5e-8
    assert r3[1] <= (1.1 if planes == 3 else 1.4) * r1[1] + 1e-10 and r3[0] <= 1.25 * r1[0] + 1e-9
    assert e3 <= 1.5 * e1 + 3e-7 * scale       # (the largest single error of a launch: K = 108 measures 7.0e-5 vs 3.5e-5 at scale 132)


def test_split3_is_deterministic_and_order_independent():
    """Same bits on every run; the row order (which tile a row lands in) does not change a row's result."""
    from hplflownet_amd import ops
    torch.manual_seed(5)
    M, C, F, N = 16500, 132, 8, 256
    A = torch.randn(M, C, device=DEV)
    Wt = torch.randn(ops.round_up(F * C, 32), N, device=DEV)
    Wt[F * C:] = 0
    W3 = ops.weight_split3(Wt)
    assert W3.P == ops.SPLIT_PLANES
    nbr = _table(M, M, F, 0.6, 11)
    perm = ops.tap_order(nbr)
    a = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, row_perm=perm, tiles=ops.tile_index(nbr, perm, BM=128), split_k=False)
    b = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, row_perm=perm, split_k=False)
    c = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, split_k=False)
    d = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, row_perm=torch.randperm(M, device=DEV).int(), split_k=False)
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)


def test_split3_generic_epilogue_is_bit_identical():
    """The epilogue has a fast form (32-bit buffer addressing; what every launch of the model takes) and the generic form it falls
    back to for operands of 2 GB and more: HPL_SPLIT3_EPILOGUE=0 forces the generic one -- same bits, with bias, LeakyReLU, a
    residual (the second tap-group pass of a layer) and a second destination."""
    import os
    import subprocess
    import sys
    import tempfile
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from hplflownet_amd import ops\n"
        "torch.manual_seed(3)\n"
        "M, C, F, N = 16500, 64, 8, 512\n"
        "A = torch.randn(M + 11, C, device='cuda')\n"
        "Wt = torch.randn(ops.round_up(F * C, 32), N, device='cuda') / (F * C) ** 0.5\n"
        "g = torch.Generator(device='cpu').manual_seed(9)\n"
        "nbr = torch.randint(0, M + 11, (F, M), generator=g, dtype=torch.int32)\n"
        "nbr[torch.rand((F, M), generator=g) > 0.7] = -1\n"
        "nbr = nbr.cuda()\n"
        "perm = ops.tap_order(nbr)\n"
        "res = torch.randn(M, N, device='cuda')\n"
        "out2 = torch.zeros(9000, N, device='cuda')\n"
        "y = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=ops.weight_split3(Wt), bias=torch.randn(N, device='cuda'), act=ops.ACT_LEAKY,\n"
        "                  res=res, row_perm=perm, tiles=ops.tile_index(nbr, perm, BM=128), split_k=False, out2=out2, rows2=9000)\n"
        "torch.save((y.cpu(), out2.cpu()), sys.argv[1])\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    with tempfile.TemporaryDirectory() as d:
        for mode in ('1', '0'):
            f = os.path.join(d, 'y%s.pt' % mode)
            r = subprocess.run([sys.executable, '-c', code, f], env=dict(os.environ, HPL_SPLIT3_EPILOGUE=mode), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
            outs.append(torch.load(f))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].abs().max()) > 0 and torch.equal(outs[0][0][:9000], outs[0][1])


def test_amax_is_the_largest_magnitude():
    """hpl_amax (the scale of the fp16-pair operands) against torch: aligned and ragged views, zero and NaN matrices."""
    from hplflownet_amd import ops
    torch.manual_seed(1)
    X = torch.randn(5000, 132, device=DEV) * torch.exp(2 * torch.randn(5000, 1, device=DEV))
    for rows, cols, view in [(5000, 132, X), (4999, 131, X), (777, 64, X[11:, 4:]), (1, 1, X[3:, 7:]), (3000, 3, X[:, 1:])]:
        got = float(ops.amax(view, rows=rows, cols=cols))
        assert got == float(view[:rows, :cols].abs().max()), (rows, cols)
    assert float(ops.amax(torch.zeros(100, 8, device=DEV))) == 0.0
    Xn = X.clone()
    Xn[4321, 77] = float('nan')
    assert np.isnan(float(ops.amax(Xn)))
    assert float(ops.amax(Xn, cols=64)) == float(X[:, :64].abs().max())


@pytest.mark.parametrize('sa,sw', [(1.0, 1.0), (1e-25, 1e-8), (1e18, 1e-12), (3e4, 40.0), (1e-3, 1e-3)])
def test_fp16_pairs_follow_the_scale_of_their_operands(sa, sw):
    """The pair form scales both operands by powers of two taken from their largest magnitudes: activations of 1e-25 or 1e18
    (far outside fp16's range) give the same relative accuracy as activations of 1."""
    from hplflownet_amd import ops
    torch.manual_seed(7)
    M, C, F, N = 8192, 160, 1, 256
    A = torch.randn(M, C, device=DEV) * sa
    Wt = torch.randn(C, N, device=DEV) * (sw / C ** 0.5)
    y = ops.gconv_raw(A, None, M, C, F, Wt, N, Wt3=ops.weight_split3(Wt, planes=2), split_k=False)
    y1 = ops.gconv_raw(A, None, M, C, F, Wt, N, split_k=False)
    ref = A.double() @ Wt.double()
    mag = A.abs().double() @ Wt.abs().double()
    r, r1 = float(((y.double() - ref).abs() / mag).max()), float(((y1.double() - ref).abs() / mag).max())
    print('scales %g %g: err / sum|a||w| pairs %.3g fp32 %.3g' % (sa, sw, r, r1))
    assert not torch.equal(y, y1) and torch.isfinite(y).all()
    assert r < 1.25 * r1 + 1e-9 and r < 6e-7


@pytest.mark.parametrize('outlier', [1e3, 1e5, 1e8])
def test_fp16_pairs_with_an_outlier_in_the_matrix(outlier):
    """ONE scale per matrix: an element `outlier` times larger than the rest pushes the small elements' lo halves towards fp16's
    subnormals.  Up to 2^18 x nothing is lost; beyond that every element keeps an ABSOLUTE error of 2^-40 of the largest magnitude
    (per product 2^-40 amax |w|) -- the bound include/hpl_bcl.h states -- instead of a relative one.  Zeros stay zeros."""
    from hplflownet_amd import ops
    torch.manual_seed(8)
    M, C, F, N = 8192, 128, 1, 256
    A = torch.randn(M, C, device=DEV)
    A[5, 9] = outlier
    A[100:200] = 0
    Wt = torch.randn(C, N, device=DEV) / C ** 0.5
    y = ops.gconv_raw(A, None, M, C, F, Wt, N, Wt3=ops.weight_split3(Wt, planes=2), split_k=False)
    ref = A.double() @ Wt.double()
    mag = A.abs().double() @ Wt.abs().double()
    err = (y.double() - ref).abs()
    bound = 6e-7 * mag + C * 2.0 ** -39 * outlier * float(Wt.abs().max())
    assert bool((err <= bound).all()), float((err / bound).max())
    assert float(y[100:200].abs().max()) == 0.0
    if outlier <= 2.0 ** 17:
        assert float((err / mag.clamp(min=1e-30)).max()) < 6e-7


@pytest.mark.parametrize('M,C,F,N,splitk', [(16500, 64, 1, 512, False), (16400, 96, 8, 320, False), (9433, 388, 15, 256, True), (3000, 40, 1, 48, False)])
def test_a_launch_can_leave_the_largest_magnitude_of_its_result(M, C, F, N, splitk):
    """hpl_gconv_desc.y_amax: the scale of the NEXT wide launch from this launch's epilogue (fast split epilogue: in registers;
    split-K, generic or fp32-MFMA launches: one more pass) -- always the exact largest |Y| after bias, residual and LeakyReLU."""
    from hplflownet_amd import ops
    torch.manual_seed(M)
    A = torch.randn(M + 5, C, device=DEV)
    Wt = torch.zeros(ops.round_up(F * C, 32), ops.round_up(N, 4), device=DEV)
    Wt[:F * C, :N] = torch.randn(F * C, N, device=DEV) / (F * C) ** 0.5
    nbr = _table(M, M + 5, F, 0.7, 2) if F > 1 else None
    perm = ops.tap_order(nbr) if F > 1 else None
    res = torch.randn(M, N, device=DEV) * 3
    slot = torch.zeros(1, device=DEV)
    W3 = ops.weight_split3(Wt) if N >= 256 else None
    y = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, bias=torch.randn(N, device=DEV), act=ops.ACT_LEAKY, res=res, row_perm=perm,
                      split_k=splitk, y_amax=slot)
    assert float(slot) == float(y.abs().max())
    g = torch.randn(M, N, device=DEV)
    slot.zero_()
    dx = ops.leaky_bwd(g, y, amax=slot)
    assert torch.equal(dx, g * torch.where(y > 0, 1.0, ops.LEAKY_RATE)) and float(slot) == float(dx.abs().max())


def _guard_word(X):
    """the range-guard word the kernels publish: ~bits of the smallest non-zero row maximum (0: every row zero)"""
    rm = X.abs().max(dim=1).values
    rm = rm[rm > 0]
    if rm.numel() == 0:
        return 0
    return (~int(rm.min().view(torch.int32)) & 0xffffffff)


def test_amax_rows_leaves_the_smallest_nonzero_row_maximum():
    """hpl_amax_rows: the largest magnitude AND the guard word (aligned / ragged views, wide rows that need several column passes,
    zero rows skipped, an all-zero matrix -> 0)."""
    from hplflownet_amd import ops
    torch.manual_seed(3)
    X = torch.randn(5000, 1028, device=DEV) * torch.exp(6 * torch.randn(5000, 1, device=DEV))
    X[17] = 0
    X[4000:4100] = 0
    for rows, cols, view in [(5000, 1028, X), (4999, 131, X), (777, 64, X[11:, 4:]), (1, 1, X[3:, 7:]), (3000, 3, X[:, 1:]), (5000, 580, X)]:
        a, g = ops.amax_rows(view, rows=rows, cols=cols)
        sub = view[:rows, :cols]
        assert float(a) == float(sub.abs().max()), (rows, cols)
        assert (int(g) & 0xffffffff) == _guard_word(sub), (rows, cols)
    a, g = ops.amax_rows(torch.zeros(100, 8, device=DEV))
    assert float(a) == 0.0 and int(g) == 0


def _rowwise(y, ref, mag):
    """largest error of every output row relative to the row's own sum |a||w| (its largest entry): the per-ROW accuracy"""
    return ((y.double() - ref).abs().max(dim=1).values / mag.max(dim=1).values.clamp(min=1e-300))


@pytest.mark.parametrize('case', ['quiet_rows', 'channels', 'gathered'])
def test_range_guard_keeps_quiet_rows_fp32_class(case):
    """The fp16-pair form scales a matrix by ONE power of two.  Trained-like activations put magnitudes of different origin into one
    matrix: per-channel log-normal scales (sigma = 4) with one hot channel x 1e6 ('channels': every row carries the hot channel,
    so every row's outputs are accurate relative to ITS sum |a||w| -- no second pass needed, and none taken), and -- the case the
    guard exists for -- ROWS that are quiet as a whole ('quiet_rows': per-row scales over 10 decades; 'gathered': the same through a
    neighbour table, where an output row sums rows of different loudness).  Row-wise relative error <= 4 x the fp32-MFMA kernel's
    (measured: below it), where the unguarded pair form is off by orders of magnitude on the quiet rows."""
    from hplflownet_amd import ops
    torch.manual_seed(11)
    M, C, N = (16500, 128, 256) if case == 'gathered' else (8192, 256, 256)      # (129 row tiles: the gathered launch fills the GPU, i.e. takes the pair form)
    F = 8 if case == 'gathered' else 1
    A = torch.randn(M, C, device=DEV)
    if case == 'channels':
        A *= torch.exp(4 * torch.randn(1, C, device=DEV))
        A[:, 37] *= 1e6
    else:
        A *= 10.0 ** (10 * torch.rand(M, 1, device=DEV) - 7)          # rows from 1e-7 to 1e3 (2^33: inside the 2^41 the second pass covers)
        A[50:60] = 0
    Wt = torch.zeros(ops.round_up(F * C, 32), N, device=DEV)
    Wt[:F * C] = torch.randn(F * C, N, device=DEV) / (F * C) ** 0.5
    W3 = ops.weight_split3(Wt, planes=2)
    nbr = _table(M, M, F, 0.6, 5) if F > 1 else None
    perm = ops.tap_order(nbr) if F > 1 else None
    tiles = ops.tile_index(nbr, perm, BM=128) if F > 1 else None
    kw = dict(row_perm=perm, tiles=tiles, split_k=False)
    trips = torch.zeros(1, dtype=torch.int32, device=DEV)
    y = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, guard_trips=trips, **kw)
    y_off = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=W3, guard=False, **kw)
    y32 = ops.gconv_raw(A, nbr, M, C, F, Wt, N, **kw)
    ref = _ref64(A, nbr, M, C, F, Wt, N)
    W64 = Wt[:F * C, :N].abs().double()
    mag = torch.zeros((M, N), dtype=torch.float64, device=DEV)
    for f in range(F):
        rows = A.abs().double() if nbr is None else torch.where((nbr[f] >= 0)[:, None], A.abs().double()[nbr[f].long().clamp(min=0)], torch.zeros((), dtype=torch.float64, device=DEV))
        mag += rows @ W64[f * C:(f + 1) * C]
    live = mag.max(dim=1).values > 0
    e, e_off, e32 = [float(_rowwise(x, ref, mag)[live].max()) for x in (y, y_off, y32)]
    print('%s: row-wise err / sum|a||w|: guarded pairs %.3g, unguarded %.3g, fp32 MFMA %.3g; second passes %d' % (case, e, e_off, e32, int(trips)))
    assert torch.isfinite(y).all() and float(y[~live].abs().max() if (~live).any() else 0.0) == 0.0
    assert e <= 4 * e32 and e < 2e-6
    if case == 'channels':
        assert int(trips) == 0 and torch.equal(y, y_off)              # nothing to guard: the launch is the round-5 launch, bit for bit
    else:
        assert int(trips) == 1 and e_off > 100 * e                    # the guard is what makes the quiet rows right


def test_range_guard_word_from_the_epilogue_is_conservative():
    """hpl_gconv_desc.y_guard: the wide epilogue leaves the guard word of what it stores (row maxima over 16 columns of a lane
    block: never above the true row maximum) -- between the true smallest row maximum and 2^-8 of it for these activations, and
    exactly hpl_amax_rows' word when the launch reduces Y in a second pass (split-K)."""
    from hplflownet_amd import ops
    torch.manual_seed(12)
    for M, C, F, N, splitk in [(16500, 64, 1, 512, False), (9433, 388, 15, 256, True)]:
        A = torch.randn(M + 5, C, device=DEV) * 10.0 ** (6 * torch.rand(M + 5, 1, device=DEV) - 3)
        Wt = torch.zeros(ops.round_up(F * C, 32), N, device=DEV)
        Wt[:F * C] = torch.randn(F * C, N, device=DEV) / (F * C) ** 0.5
        nbr = _table(M, M + 5, F, 0.7, 2) if F > 1 else None
        perm = ops.tap_order(nbr) if F > 1 else None
        slot, g = torch.zeros(1, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
        y = ops.gconv_raw(A, nbr, M, C, F, Wt, N, Wt3=ops.weight_split3(Wt, planes=2), act=ops.ACT_LEAKY, row_perm=perm, split_k=splitk,
                          y_amax=slot, y_guard=g)
        assert float(slot) == float(y.abs().max())
        rm = y.abs().max(dim=1).values
        true_min = float(rm[rm > 0].min())
        got = torch.tensor([~int(g) & 0x7fffffff], dtype=torch.int32).view(torch.float32).item()
        if splitk:
            assert got == true_min
        else:
            assert true_min * 2.0 ** -8 <= got <= true_min, (got, true_min)
