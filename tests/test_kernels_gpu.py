"""Per-kernel parity on a real MI355X: every C-ABI entry point against a plain PyTorch fp64/fp32 CPU
statement of the same op (and the NMS against the CPU oracle, bit-exact).  Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from phnet_amd import hip_ops
    return hip_ops


def dev(t):
    return t.to("cuda").contiguous()


def nhwc(t):   # NCHW cpu -> NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def close(a, b, tol=2e-4, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, (what, err, scale)


# ------------------------------------------------------------------------------------------------ NMS
def _random_lanes(K, n_off, seed):
    r = np.random.default_rng(seed)
    rows = np.zeros((K, 5 + n_off), np.float32)
    base = r.uniform(0, 800, (K, 1)) + r.normal(0, 12, (K, n_off)).cumsum(1)
    if K > 1:
        base[K // 2:] = base[:K - K // 2] + r.normal(0, 30, (K - K // 2, 1))
    rows[:, 5:] = base
    rows[:, 2] = r.uniform(-0.3, 1.2, K)
    rows[:, 4] = r.uniform(-2, 40, K)
    scores = (r.permutation(K).astype(np.float32) + 0.5) / K
    return rows, scores


@pytest.mark.parametrize("K,n_off,top_k", [(1, 36, 4), (2, 36, 4), (63, 36, 4), (64, 36, 4), (65, 36, 4), (240, 36, 4),
                                           (240, 36, 1000), (240, 72, 4), (500, 36, 7), (700, 72, 3)])
def test_lane_nms_bit_exact_vs_oracle(ops, K, n_off, top_k):
    from oracle import lane_nms as ON
    rows, scores = _random_lanes(K, n_off, K + n_off)
    keep, num, parent = ops.lane_nms(dev(torch.from_numpy(rows)), dev(torch.from_numpy(scores)), 50.0, top_k)
    rk, rn, rp = ON.lane_nms(rows, scores, 50.0, top_k)
    assert int(num) == rn
    assert keep.cpu().numpy().tolist() == rk.tolist()
    assert parent.cpu().numpy().tolist() == rp.tolist()


def test_lane_nms_batched_ragged_counts(ops):
    from oracle import lane_nms as ON
    F_, K = 6, 240
    counts = [240, 0, 1, 100, 64, 239]
    rows = np.zeros((F_, K, 41), np.float32); scores = np.zeros((F_, K), np.float32)
    for f in range(F_):
        rows[f], scores[f] = _random_lanes(K, 36, 100 + f)
    keep, num, parent = ops.lane_nms(dev(torch.from_numpy(rows)), dev(torch.from_numpy(scores)), 50.0, 4,
                                     counts=torch.tensor(counts, dtype=torch.int32, device="cuda"))
    for f, c in enumerate(counts):
        rk, rn, rp = ON.lane_nms(rows[f, :c], scores[f, :c], 50.0, 4)
        assert int(num[f]) == rn
        assert keep[f, :c].cpu().numpy().tolist() == rk.tolist()
        assert parent[f, :c].cpu().numpy().tolist() == rp.tolist()


def test_lane_nms_nan_and_tied_scores(ops):
    """Scores with NaNs, infinities and exact ties: the in-kernel ranking is a total order (NaN above every number as in
    ATen's descending sort, ties by index), so the rank -> row table is fully written and the result equals the oracle's."""
    from oracle import lane_nms as ON
    for K, seed in ((7, 0), (240, 1), (65, 2)):
        rows, scores = _random_lanes(K, 36, 900 + seed)
        r = np.random.default_rng(seed)
        scores[r.choice(K, max(1, K // 8), replace=False)] = np.nan
        scores[r.choice(K, max(1, K // 8), replace=False)] = 0.25
        scores[r.integers(K)] = np.inf
        scores[r.integers(K)] = -np.inf
        keep, num, parent = ops.lane_nms(dev(torch.from_numpy(rows)), dev(torch.from_numpy(scores)), 50.0, 1000)
        rk, rn, rp = ON.lane_nms(rows, scores, 50.0, 1000)
        assert int(num) == rn and keep.cpu().numpy().tolist() == rk.tolist() and parent.cpu().numpy().tolist() == rp.tolist()
        assert sorted(keep[:rn].cpu().tolist()) == sorted(set(keep[:rn].cpu().tolist()))      # every keeper is a distinct row


def test_lane_nms_empty(ops):
    keep, num, parent = ops.lane_nms(torch.zeros(0, 41, device="cuda"), torch.zeros(0, device="cuda"), 50.0, 4)
    assert int(num) == 0 and keep.numel() == 0


# ------------------------------------------------------------------------------------------------ ROI pooling
@pytest.mark.parametrize("h,w", [(10, 25), (20, 50), (40, 100), (2, 5)])
def test_roi_pool_fwd_bwd_vs_grid_sample(ops, h, w):
    torch.manual_seed(h * w)
    B, N, P, C = 1, 240, 36, 64
    fmap = torch.randn(B, C, h, w, dtype=torch.float64, requires_grad=True)
    xs = (torch.rand(B, N, P, dtype=torch.float64) * 1.6 - 0.3).requires_grad_(True)   # some anchors leave the map
    ys = torch.flip(1 - torch.arange(P, dtype=torch.float64) / (P - 1), dims=[0])
    gx = torch.flip(xs, dims=[2]) * 2 - 1
    gy = (ys * 2 - 1).view(1, 1, P).expand(B, N, P)
    ref = F.grid_sample(fmap, torch.stack([gx, gy], -1), align_corners=True).permute(0, 2, 3, 1)    # [B,N,P,C]
    gout = torch.randn_like(ref)
    ref.backward(gout)
    out = ops.roi_pool_fwd(nhwc(fmap.detach().float()), dev(xs.detach().float()), dev(ys.float()))
    close(out, ref, 1e-5)
    dmap = torch.zeros(B, h, w, C, device="cuda")
    dxs = ops.roi_pool_bwd(dev(gout.float()), nhwc(fmap.detach().float()), dev(xs.detach().float()), dev(ys.float()), dmap, True)
    close(dmap, fmap.grad.permute(0, 2, 3, 1), 1e-4)
    close(dxs, xs.grad, 1e-4)


# ------------------------------------------------------------------------------------------------ conv / linear
CONV_CASES = [
    # N, Hi, Wi, Ci, Co, R, stride, pad
    (2, 16, 20, 64, 64, 3, 1, 1),
    (5, 20, 50, 64, 128, 3, 2, 1),
    (2, 20, 50, 64, 128, 1, 2, 0),
    (1, 10, 25, 512, 512, 3, 1, 1),        # split-K path (few tiles, long K)
    (2, 33, 47, 4, 64, 7, 2, 3),           # stem geometry (3 channels padded to 4), ragged sizes
    (3, 9, 13, 128, 64, 1, 1, 0),
    (1, 80, 200, 64, 64, 3, 1, 1),         # 128x64 tile path
    (240, 1, 1, 1024, 8192, 1, 1, 0),      # hyper-net linear
    (240, 1, 1, 4608, 1024, 1, 1, 0),      # long-K linear -> split-K
    (240, 1, 1, 64, 36, 1, 1, 0),          # ragged Co (36 offsets)
    (7, 1, 1, 128, 4, 1, 1, 0),
]


@pytest.fixture
def f32_mfma(ops):
    """GEMM kernels on the f32-input MFMA (v_mfma_f32_32x32x2_f32) - the round-1 arithmetic, still selectable."""
    ops.set_mma_mode("f32")
    yield
    ops.set_mma_mode(ops.DEFAULT_MMA)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(ops, f32_mfma, case):
    N, Hi, Wi, Ci, Co, R, stride, pad = case
    torch.manual_seed(sum(case))
    x = torch.randn(N, Ci, Hi, Wi, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, R, R, dtype=torch.float64) / (Ci * R * R) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, dtype=torch.float64)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    xd, wd = nhwc(x.detach().float()), nhwc(w.detach().float())
    y = ops.conv2d_fwd(xd, wd, dev(b.float()), stride, pad)
    close(y, ref.permute(0, 2, 3, 1), 2e-5)
    yr = ops.conv2d_fwd(xd, wd, dev(b.float()), stride, pad, relu=True)
    close(yr, F.relu(ref).permute(0, 2, 3, 1), 2e-5)
    gyd = nhwc(gy.float())
    dx = ops.conv2d_dgrad(gyd, wd, (Hi, Wi), stride, pad)
    close(dx, x.grad.permute(0, 2, 3, 1), 2e-5)
    dw = ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad)
    close(dw, w.grad.permute(0, 2, 3, 1), 2e-5)
    dw2 = ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad, dw=dw.clone(), accumulate=True)
    close(dw2, 2 * w.grad.permute(0, 2, 3, 1), 2e-5)
    db = torch.full((Co,), 7.0, device="cuda")                       # fused bias gradient = column sums of dy
    dw3 = ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad, dbias=db)
    close(dw3, w.grad.permute(0, 2, 3, 1), 2e-5)
    close(db, gy.sum(dim=(0, 2, 3)), 2e-5)
    ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad, dw=dw3, accumulate=True, dbias=db)
    close(db, 2 * gy.sum(dim=(0, 2, 3)), 2e-5)


@pytest.fixture
def split_bf16(ops):
    """GEMM kernels in split-bf16 arithmetic for the duration of one test (process-global switch, restored afterwards)."""
    ops.set_mma_mode("split_bf16")
    yield
    ops.set_mma_mode(ops.DEFAULT_MMA)


# every code path of CONV_CASES (tiles, split-K, ragged, stem, linears) once more in the opt-in split-bf16 arithmetic:
# two bf16 terms per operand, three bf16 MFMAs per product, f32 accumulation - ~2^-16 per product, 4-5e-6 of the output
# scale measured on the hot shapes (tests/tools/bench_mma.py); held to 5e-5 here (f32-input MFMA: 2e-5)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad_split_bf16(ops, split_bf16, case):
    N, Hi, Wi, Ci, Co, R, stride, pad = case
    torch.manual_seed(sum(case) + 1)
    x = torch.randn(N, Ci, Hi, Wi, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, R, R, dtype=torch.float64) / (Ci * R * R) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, dtype=torch.float64)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    xd, wd, gyd = nhwc(x.detach().float()), nhwc(w.detach().float()), nhwc(gy.float())
    close(ops.conv2d_fwd(xd, wd, dev(b.float()), stride, pad, relu=True), F.relu(ref).permute(0, 2, 3, 1), 5e-5)
    close(ops.conv2d_dgrad(gyd, wd, (Hi, Wi), stride, pad), x.grad.permute(0, 2, 3, 1), 5e-5)
    db = torch.full((Co,), 7.0, device="cuda")
    dw = ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad, dbias=db)
    close(dw, w.grad.permute(0, 2, 3, 1), 5e-5)
    close(db, gy.sum(dim=(0, 2, 3)), 2e-5)                           # the bias gradient stays an f32 sum
    ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad, dw=dw, accumulate=True)
    close(dw, 2 * w.grad.permute(0, 2, 3, 1), 5e-5)


@pytest.mark.parametrize("case", CONV_CASES[:8])
def test_conv_split3_bf16_is_as_accurate_as_f32_mfma(ops, case):
    """Exact three-term bf16 split, 6 bf16 MFMAs per product: the f32 kernels' own tolerance (2e-5)."""
    N, Hi, Wi, Ci, Co, R, stride, pad = case
    torch.manual_seed(sum(case) + 2)
    x = torch.randn(N, Ci, Hi, Wi, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, R, R, dtype=torch.float64) / (Ci * R * R) ** 0.5).requires_grad_(True)
    ref = F.conv2d(x, w, None, stride=stride, padding=pad)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    xd, wd, gyd = nhwc(x.detach().float()), nhwc(w.detach().float()), nhwc(gy.float())
    ops.set_mma_mode("split3_bf16")
    try:
        close(ops.conv2d_fwd(xd, wd, None, stride, pad), ref.permute(0, 2, 3, 1), 2e-5)
        close(ops.conv2d_dgrad(gyd, wd, (Hi, Wi), stride, pad), x.grad.permute(0, 2, 3, 1), 2e-5)
        close(ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad), w.grad.permute(0, 2, 3, 1), 2e-5)
    finally:
        ops.set_mma_mode(ops.DEFAULT_MMA)


@pytest.fixture
def bf16x3(ops):
    """GEMM kernels in the staged exact-split arithmetic (three bf16 planes in LDS, 6 bf16 MFMAs per product)."""
    ops.set_mma_mode("bf16x3")
    yield
    ops.set_mma_mode(ops.DEFAULT_MMA)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad_bf16x3(ops, bf16x3, case):
    """EVERY code path of CONV_CASES (tiles, split-K, ragged edges, stem, linears, strided and K-strided operands through the
    hardware-transposed LDS reads) in the staged three-term split, held to the f32-input kernels' own tolerance (2e-5 of the
    output scale against fp64): x = hi + mid + lo is an exact decomposition and only products below 2^-24 are dropped."""
    N, Hi, Wi, Ci, Co, R, stride, pad = case
    torch.manual_seed(sum(case) + 3)
    x = torch.randn(N, Ci, Hi, Wi, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, R, R, dtype=torch.float64) / (Ci * R * R) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, dtype=torch.float64)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    xd, wd, gyd = nhwc(x.detach().float()), nhwc(w.detach().float()), nhwc(gy.float())
    close(ops.conv2d_fwd(xd, wd, dev(b.float()), stride, pad), ref.permute(0, 2, 3, 1), 2e-5)
    close(ops.conv2d_fwd(xd, wd, dev(b.float()), stride, pad, relu=True), F.relu(ref).permute(0, 2, 3, 1), 2e-5)
    close(ops.conv2d_dgrad(gyd, wd, (Hi, Wi), stride, pad), x.grad.permute(0, 2, 3, 1), 2e-5)
    db = torch.full((Co,), 7.0, device="cuda")
    dw = ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad, dbias=db)
    close(dw, w.grad.permute(0, 2, 3, 1), 2e-5)
    close(db, gy.sum(dim=(0, 2, 3)), 2e-5)
    ops.conv2d_wgrad(gyd, xd, wd.shape, stride, pad, dw=dw, accumulate=True)
    close(dw, 2 * w.grad.permute(0, 2, 3, 1), 2e-5)


@pytest.mark.parametrize("case", [(2, 16, 20, 64, 64), (3, 5, 17, 64, 128), (4, 2, 5, 512, 512), (1, 1, 70, 64, 64), (5, 20, 50, 256, 64),
                                  (1, 10, 25, 512, 512), (4, 3, 23, 128, 64), (3, 8, 20, 128, 128), (2, 2, 2, 64, 64), (1, 40, 100, 64, 192)])
def test_packed_weight_3x3_kernel_vs_fp64(ops, bf16x3, case):
    """conv3p_kernel (csrc/conv3p.hip: 3x3 / stride 1 / pad 1 on weights packed into bf16 planes in MFMA fragment order), forward
    and data gradient: image rows narrower / wider than the 128-pixel block, blocks that span image rows and frames, one- and
    two-row images (every row a border row), ragged pixel counts, split-K and unsplit plans, bias / addend / ReLU epilogues,
    the BatchNorm statistics rows, the batched pack launch - against fp64 at the tolerance of the generic kernel (2e-5 of the
    output scale) and against the generic entry points."""
    N, Hi, Wi, Ci, Co = case
    torch.manual_seed(sum(case) + 5)
    x = torch.randn(N, Ci, Hi, Wi, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, dtype=torch.float64)
    res = torch.randn(N, Co, Hi, Wi, dtype=torch.float64)
    ref = F.conv2d(x, w, None, stride=1, padding=1)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    xd, wd, gyd = nhwc(x.detach().float()), nhwc(w.detach().float()), nhwc(gy.float())
    assert ops.conv3p_applies(N * Hi * Wi, Ci, Co) and ops.conv3p_applies(N * Hi * Wi, Co, Ci)
    pf, pd = ops.conv3p_pack(wd, False), ops.conv3p_pack(wd, True)
    plan = ops.Conv3pPackPlan([wd, wd * 2.0])                       # the one-launch form over several weights
    plan.refresh()
    assert torch.equal(plan.images[0][False], pf) and torch.equal(plan.images[0][True], pd)
    assert torch.equal(plan.images[1][False].view(torch.int16), ops.conv3p_pack(wd * 2.0, False).view(torch.int16))
    want = ref.permute(0, 2, 3, 1)
    close(ops.conv3p(xd, pf, Co), want, 2e-5)
    close(ops.conv3p(xd, pf, Co, bias=dev(b.float()), addend=nhwc(res.float()), relu=True),
          F.relu(ref + b[None, :, None, None] + res).permute(0, 2, 3, 1), 2e-5)
    close(ops.conv3p(gyd, pd, Ci, dgrad=True), x.grad.permute(0, 2, 3, 1), 2e-5)
    close(ops.conv3p(gyd, pd, Ci, dgrad=True, addend=xd), (x.grad + x.detach()).permute(0, 2, 3, 1), 2e-5)
    if Co & (Co - 1) == 0:
        y, (part, nblk) = ops.conv3p(xd, pf, Co, stats=True)
        close(y, want, 2e-5)
        sums = part[:nblk * 2 * Co * 4].view(torch.float32).view(nblk, 2, Co).double().sum(0).cpu()
        rd = ref.detach()
        assert float((sums[0] - rd.sum((0, 2, 3))).abs().max()) <= 2e-5 * float(rd.abs().sum((0, 2, 3)).max())
        assert float((sums[1] - (rd ** 2).sum((0, 2, 3))).abs().max()) <= 2e-5 * float((rd ** 2).sum((0, 2, 3)).max())
    # and the entry points it replaces in the trunk schedule
    close(ops.conv3p(xd, pf, Co), ops.conv2d_fwd(xd, wd, None, 1, 1).double().cpu(), 2e-5)
    close(ops.conv3p(gyd, pd, Ci, dgrad=True), ops.conv2d_dgrad(gyd, wd, (Hi, Wi), 1, 1).double().cpu(), 2e-5)


@pytest.mark.parametrize("case", [(1200, 1024, 1024), (257, 1152, 1024), (300, 1024, 1152), (1200, 2048, 512)])
def test_many_row_linear_weight_gradient_kernel_vs_fp64(ops, bf16x3, case):
    """wgrad1s_kernel (csrc/wgrad1s.hip: 128 x 128 tiles, 8 consumer + 4 producer waves) - the weight and bias gradient of a Linear
    layer over many rows: ragged row counts (a last step of one row), overwrite and accumulate, with and without the bias
    gradient, against fp64 and against the generic kernel (phnet_tune_wgrad bit 6 switches this one off)."""
    from phnet_amd._lib import lib
    P, Ci, Co = case
    torch.manual_seed(P + Ci + Co)
    x = torch.randn(P, Ci, dtype=torch.float64)
    gy = torch.randn(P, Co, dtype=torch.float64)
    want, want_b = gy.t() @ x, gy.sum(0)
    xd, gyd = x.float().cuda().view(P, 1, 1, Ci), gy.float().cuda().view(P, 1, 1, Co)
    shape = (Co, 1, 1, Ci)
    db = torch.full((Co,), 7.0, device="cuda")
    dw = ops.conv2d_wgrad(gyd, xd, shape, 1, 0, dbias=db)
    close(dw.view(Co, Ci), want, 2e-5)
    close(db, want_b, 2e-5)
    ops.conv2d_wgrad(gyd, xd, shape, 1, 0, dw=dw, dbias=db, accumulate=True)
    close(dw.view(Co, Ci), 2 * want, 2e-5)
    close(db, 2 * want_b, 2e-5)
    dw_nb = ops.conv2d_wgrad(gyd, xd, shape, 1, 0)
    close(dw_nb.view(Co, Ci), want, 2e-5)
    assert lib().phnet_tune_wgrad(1 | 64, 768) == 0                      # the generic kernel on the same operands
    try:
        dw_generic = ops.conv2d_wgrad(gyd, xd, shape, 1, 0)
    finally:
        assert lib().phnet_tune_wgrad(1, 768) == 0
    close(dw_generic.view(Co, Ci), want, 2e-5)
    close(dw_nb, dw_generic, 1e-5)


@pytest.mark.parametrize("case", [(2, 16, 20, 64, 64), (3, 5, 17, 64, 128), (2, 7, 16, 128, 64), (1, 1, 70, 64, 64), (5, 20, 50, 256, 64),
                                  (1, 10, 25, 512, 512), (4, 3, 23, 64, 64), (4, 1, 20, 64, 64), (3, 2, 16, 64, 64)])
def test_wgrad_three_taps_kernel_vs_fp64(ops, bf16x3, case):
    """conv_wgrad3x3_kernel (3x3 / stride 1 / pad 1, the three taps of a filter row from one staged pixel block): image rows
    narrower / wider than the 16-pixel block, blocks that span image rows and frames, one-row images (every row is the top and
    the bottom row), ragged pixel counts, bias gradient, accumulation - against fp64 at the tolerance of the generic kernel,
    and against the generic kernel itself (phnet_tune_wgrad bit 3 switches the three-taps kernel off)."""
    from phnet_amd._lib import lib
    N, Hi, Wi, Ci, Co = case
    torch.manual_seed(sum(case) + 11)
    x = torch.randn(N, Ci, Hi, Wi, dtype=torch.float64)
    w = (torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5).requires_grad_(True)
    ref = F.conv2d(x, w, None, stride=1, padding=1)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    xd, gyd = nhwc(x.float()), nhwc(gy.float())
    want = w.grad.permute(0, 2, 3, 1)
    shape = (Co, 3, 3, Ci)
    db = torch.full((Co,), 7.0, device="cuda")
    dw = ops.conv2d_wgrad(gyd, xd, shape, 1, 1, dbias=db)
    close(dw, want, 2e-5)
    close(db, gy.sum(dim=(0, 2, 3)), 2e-5)
    ops.conv2d_wgrad(gyd, xd, shape, 1, 1, dw=dw, dbias=db, accumulate=True)
    close(dw, 2 * want, 2e-5)
    close(db, 2 * gy.sum(dim=(0, 2, 3)), 2e-5)
    # without a bias gradient the producer / consumer variant runs (csrc/wgrad3s.hip: four extra waves stage, twelve multiply)
    dw_pc = ops.conv2d_wgrad(gyd, xd, shape, 1, 1)
    close(dw_pc, want, 2e-5)
    ops.conv2d_wgrad(gyd, xd, shape, 1, 1, dw=dw_pc, accumulate=True)
    close(dw_pc, 2 * want, 2e-5)
    assert lib().phnet_tune_wgrad(1 | 32, 768) == 0                      # bit 5: the all-waves-stage kernel without a bias gradient
    try:
        close(ops.conv2d_wgrad(gyd, xd, shape, 1, 1), want, 2e-5)
    finally:
        assert lib().phnet_tune_wgrad(1, 768) == 0
    assert lib().phnet_tune_wgrad(1 | 8, 768) == 0                       # the generic kernel on the same operands
    try:
        dw_generic = ops.conv2d_wgrad(gyd, xd, shape, 1, 1)
    finally:
        assert lib().phnet_tune_wgrad(1, 768) == 0
    close(dw_generic, want, 2e-5)
    for flags, target in ((1, 64), (1, 1024), (1 | 32, 64), (1 | 32 | 16, 256), (1 | 32 | 16, 1024)):      # other splits of the pixel range, down to
        assert lib().phnet_tune_wgrad(flags, -target) == 0                   # one step; bit 4: 32-pixel instead of 16-pixel steps (bit 5 kernel)
        try:
            close(ops.conv2d_wgrad(gyd, xd, shape, 1, 1), want, 2e-5)
        finally:
            assert lib().phnet_tune_wgrad(1, -256) == 0


@pytest.mark.parametrize("case", [(2, 16, 20, 64, 64), (3, 5, 17, 16, 36), (2, 7, 16, 48, 64), (1, 1, 70, 32, 128), (5, 20, 50, 256, 64),
                                  (1, 10, 25, 512, 512), (4, 3, 2, 64, 64), (1, 80, 200, 64, 64)])
def test_conv3x3_three_taps_forward_and_dgrad_vs_fp64(ops, bf16x3, case):
    """conv3x3s1_kernel (3x3 / stride 1 / pad 1: a staged 66-pixel block serves the three taps of a filter row): blocks that span
    image rows and frames, one-row and two-pixel-wide images, pixel counts that are no multiple of the tile, channel counts of 1-3
    chunks and ragged output channels, split-K, bias / addend / ReLU epilogue and the BatchNorm statistics - against fp64 at the
    generic kernel's tolerance and against the generic kernel (phnet_tune_force_k_tile(-5) switches this one off)."""
    from phnet_amd._lib import lib
    N, Hi, Wi, Ci, Co = case
    torch.manual_seed(sum(case) + 17)
    x = torch.randn(N, Ci, Hi, Wi, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Co, Ci, 3, 3, dtype=torch.float64) / (Ci * 9) ** 0.5)
    b = torch.randn(Co, dtype=torch.float64)
    ref = F.conv2d(x, w, b, stride=1, padding=1)
    gy = torch.randn_like(ref)
    ref.backward(gy)
    xd, wd, gyd = nhwc(x.detach().float()), nhwc(w.float()), nhwc(gy.float())
    add = torch.randn(N, Hi, Wi, Co, device="cuda")
    def run():
        y = ops.conv2d_fwd(xd, wd, dev(b.float()), 1, 1)
        yr = ops.conv2d_fwd(xd, wd, dev(b.float()), 1, 1, relu=True, addend=add)
        st = ops.conv2d_fwd(xd, wd, None, 1, 1, stats=True) if Co & (Co - 1) == 0 else None
        dx = ops.conv2d_dgrad(gyd, wd, (Hi, Wi), 1, 1)
        dxa = ops.conv2d_dgrad(gyd, wd, (Hi, Wi), 1, 1, addend=xd)
        return y, yr, st, dx, dxa
    y, yr, st, dx, dxa = run()
    close(y, ref.detach().permute(0, 2, 3, 1), 2e-5)
    close(yr, F.relu(ref.detach().permute(0, 2, 3, 1) + add.cpu().double()), 2e-5)
    close(dx, x.grad.permute(0, 2, 3, 1), 2e-5)
    close(dxa, x.grad.permute(0, 2, 3, 1) + x.detach().permute(0, 2, 3, 1), 2e-5)
    if st is not None:
        y0, (part, nblk) = st
        raw = (ref.detach() - b.view(1, -1, 1, 1)).permute(0, 2, 3, 1).reshape(-1, Co)
        sums = part.view(torch.float32)[:nblk * 2 * Co].view(nblk, 2, Co).double().sum(0).cpu()
        assert float((sums[0] - raw.sum(0)).abs().max()) <= 1e-4 * float(raw.abs().sum(0).max())
        assert float((sums[1] - (raw * raw).sum(0)).abs().max()) <= 1e-4 * float((raw * raw).sum(0).max())
    assert lib().phnet_tune_force_k_tile(-5) == 0
    try:
        g = run()
    finally:
        assert lib().phnet_tune_force_k_tile(-6) == 0
    for a_, b_ in ((y, g[0]), (yr, g[1]), (dx, g[3]), (dxa, g[4])):
        assert float((a_ - b_).abs().max()) <= 4e-5 * float(b_.abs().max())


def test_bf16x3_split_is_exact_on_adversarial_values(ops, bf16x3):
    """Operands chosen so that a two-term split would visibly lose bits: every mantissa bit set, magnitudes from 2^-100 to
    2^100 in one row, exact cancellation.  A 1x1 'convolution' with K = 64 against fp64."""
    torch.manual_seed(0)
    M, K, N = 256, 64, 64
    mant = 1.0 + (2.0 ** 23 - 1) / 2.0 ** 23                                         # 24 mantissa bits set
    x = torch.full((M, K), mant, dtype=torch.float64) * torch.where(torch.rand(M, K) < 0.5, -1.0, 1.0).double()
    x[:, ::7] *= 2.0 ** -20
    x[1] *= 2.0 ** 60
    x[2] *= 2.0 ** -60
    w = torch.randn(N, K, dtype=torch.float64).float().double()
    w[:, 1::5] = mant
    ref = x @ w.t()
    y = ops.linear_fwd(dev(x.float()), dev(w.float()), None)
    err = (y.cpu().double() - ref).abs() / (x.abs() @ w.abs().t())                   # relative to the sum of |products|
    assert float(err.max()) <= 3e-7, float(err.max())


def test_split_bf16_large_problem_takes_the_large_tiles(ops, split_bf16):
    """Problems with thousands of tiles run on 128x128 / 128x64 tiles in split-bf16 mode (conv.hip pick_tile)."""
    torch.manual_seed(5)
    for (N, Hi, Wi, Ci, Co) in [(4, 40, 100, 128, 128), (8, 80, 200, 64, 64)]:
        x = torch.randn(N, Hi, Wi, Ci, device="cuda"); w = torch.randn(Co, 3, 3, Ci, device="cuda") / (9 * Ci) ** 0.5
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, 1, 1).permute(0, 2, 3, 1)
        close(ops.conv2d_fwd(x, w, None, 1, 1), ref, 5e-5)
        gy = torch.randn_like(x[..., :Co])
        refd = torch.nn.grad.conv2d_input((N, Ci, Hi, Wi), w.permute(0, 3, 1, 2).double(), gy.permute(0, 3, 1, 2).double(), 1, 1)
        close(ops.conv2d_dgrad(gy, w, (Hi, Wi), 1, 1), refd.permute(0, 2, 3, 1), 5e-5)


@pytest.mark.parametrize("M,K,N", [(240, 128, 384), (240, 192, 44), (37, 64, 64), (240, 2304, 576)])
def test_linear_backward_fused_launch_split_bf16(ops, split_bf16, M, K, N):
    torch.manual_seed(M + K + N + 1)
    x = torch.randn(M, K, dtype=torch.float64)
    w = torch.randn(N, K, dtype=torch.float64) / K ** 0.5
    dy = torch.randn(M, N, dtype=torch.float64)
    y = torch.relu(torch.randn(M, N, dtype=torch.float64))
    gm = dy * (y > 0)
    dw0, db0 = torch.randn(N, K, dtype=torch.float64), torch.randn(N, dtype=torch.float64)
    dwd, dbd = dev(dw0.float()), dev(db0.float())
    dx = ops.linear_bwd(dev(dy.float()), dev(x.float()), dev(w.float()), dwd, dbd, accumulate=True, relu_y=dev(y.float()))
    close(dx, gm @ w, 5e-5); close(dwd, dw0 + gm.t() @ x, 5e-5); close(dbd, db0 + gm.sum(0), 3e-5)


def test_linear_wrappers(ops):
    torch.manual_seed(1)
    x = torch.randn(240, 2304, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(576, 2304, dtype=torch.float64) / 48).requires_grad_(True)
    b = torch.randn(576, dtype=torch.float64)
    ref = F.linear(x, w, b)
    g = torch.randn_like(ref)
    ref.backward(g)
    close(ops.linear_fwd(dev(x.detach().float()), dev(w.detach().float()), dev(b.float())), ref, 2e-5)
    close(ops.linear_dgrad(dev(g.float()), dev(w.detach().float())), x.grad, 2e-5)
    close(ops.linear_wgrad(dev(g.float()), dev(x.detach().float())), w.grad, 2e-5)
    close(ops.colsum(dev(g.float())), g.sum(0), 2e-5)


def test_stem_layout_helpers(ops):
    x = torch.randn(3, 3, 17, 23)
    y = ops.nchw3_to_nhwc4(dev(x))
    assert torch.equal(y[..., :3].cpu(), x.permute(0, 2, 3, 1)) and float(y[..., 3].abs().max()) == 0.0
    w = torch.randn(64 * 49, 3)
    p = ops.pad_channels(dev(w), 4)
    assert torch.equal(p[:, :3].cpu(), w) and float(p[:, 3].abs().max()) == 0.0
    assert torch.equal(ops.pad_channels(p, 3).cpu(), w)


# ------------------------------------------------------------------------------------------------ BN / pool / FPN
@pytest.mark.parametrize("shape,relu,res", [((5, 64, 40, 50), True, False), ((5, 128, 20, 25), False, True),
                                            ((2, 512, 5, 7), True, True), ((3, 256, 9, 11), False, False)])
def test_batchnorm_train_fwd_bwd(ops, shape, relu, res):
    torch.manual_seed(shape[1])
    N, C, H, W = shape
    x = (torch.randn(shape, dtype=torch.float64) * 2 + 0.5).requires_grad_(True)
    r = torch.randn(shape, dtype=torch.float64, requires_grad=True) if res else None
    g = (torch.rand(C, dtype=torch.float64) + 0.5).requires_grad_(True)
    b = torch.randn(C, dtype=torch.float64, requires_grad=True)
    rm, rv = torch.randn(C, dtype=torch.float64) * 0.1, torch.rand(C, dtype=torch.float64) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm_ref, rv_ref, g, b, True, 0.1, 1e-5)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    gy = torch.randn_like(y)
    y.backward(gy)
    rmd, rvd = dev(rm.float()), dev(rv.float())
    xd = nhwc(x.detach().float())
    rd = nhwc(r.detach().float()) if res else None
    yd, sm, si = ops.bn_fwd(xd, dev(g.detach().float()), dev(b.detach().float()), rmd, rvd, True, 1e-5, 0.1, rd, relu)
    close(yd, y.permute(0, 2, 3, 1), 1e-5)
    close(rmd, rm_ref, 1e-5); close(rvd, rv_ref, 1e-5)
    dres = torch.zeros_like(xd) if res else None
    dx, dg, db = ops.bn_bwd(nhwc(gy.float()), xd, yd, sm, si, dev(g.detach().float()), relu, dres)
    close(dx, x.grad.permute(0, 2, 3, 1), 2e-5)
    close(dg, g.grad, 2e-5); close(db, b.grad, 2e-5)
    if res:
        close(dres, r.grad.permute(0, 2, 3, 1), 1e-6)


@pytest.mark.parametrize("case", [(5, 20, 50, 64, 64, 3, 1, 1), (1, 10, 25, 512, 512, 3, 1, 1), (5, 20, 50, 64, 128, 3, 2, 1),
                                  (2, 33, 47, 4, 64, 7, 2, 3), (1, 80, 200, 64, 64, 3, 1, 1), (3, 9, 13, 128, 256, 1, 1, 0)])
@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_conv_epilogue_statistics_feed_batchnorm(ops, case, mode):
    """phnet_conv2d_fwd_fused(stats): the per-channel (sum, sum of squares) partials written by the GEMM epilogue - or by the
    split-K reduce when the plan splits K - finalised by phnet_bn_finalize_partials give the same BatchNorm forward (output,
    saved mean / invstd, running statistics) as the separate statistics pass; and the fused residual + ReLU epilogue
    (eval-mode BatchNorm folded into weights and bias) equals conv -> affine -> add -> relu."""
    N, Hi, Wi, Ci, Co, R, stride, pad = case
    torch.manual_seed(sum(case))
    ops.set_mma_mode(mode)
    try:
        x = torch.randn(N, Hi, Wi, Ci, device="cuda")
        w = torch.randn(Co, R, R, Ci, device="cuda") / (Ci * R * R) ** 0.5
        g, b = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda")
        c0 = ops.conv2d_fwd(x, w, None, stride, pad)
        c1, partials = ops.conv2d_fwd(x, w, None, stride, pad, stats=True)
        assert torch.equal(c0, c1)
        part, nblk = partials
        sums = part.view(torch.float32)[: nblk * 2 * Co].view(nblk, 2, Co).double().sum(0)
        flat = c0.reshape(-1, Co).double()
        close(sums[0], flat.sum(0), 1e-5); close(sums[1], (flat * flat).sum(0), 1e-5)
        rm0, rv0 = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        rm1, rv1 = rm0.clone(), rv0.clone()
        res = torch.randn_like(c0)
        # (the fused partials first: the separate statistics pass reuses the same scratch slot)
        y1, sm1, si1 = ops.bn_fwd(c1, g, b, rm1, rv1, True, 1e-5, 0.1, res, True, partials=partials)
        y0, sm0, si0 = ops.bn_fwd(c0, g, b, rm0, rv0, True, 1e-5, 0.1, res, True)
        close(y1, y0, 1e-5); close(sm1, sm0, 1e-6); close(si1, si0, 1e-5); close(rm1, rm0, 1e-6); close(rv1, rv0, 1e-5)
        # eval fold: relu(conv(x, w * s) + (beta - mean * s) + res) in ONE launch
        mean, var = torch.randn(Co, device="cuda") * 0.1, torch.rand(Co, device="cuda") + 0.5
        sc = g * torch.rsqrt(var + 1e-5)
        fused = ops.conv2d_fwd(x, (w.reshape(Co, -1) * sc[:, None]).reshape(w.shape).contiguous(), (b - mean * sc).contiguous(),
                               stride, pad, relu=True, addend=res)
        want = torch.relu((c0.double() - mean.double()) * sc.double() + b.double() + res.double())
        close(fused, want, 2e-5)
    finally:
        ops.set_mma_mode(ops.DEFAULT_MMA)


def test_batchnorm_eval(ops):
    x = torch.randn(2, 64, 8, 10, dtype=torch.float64)
    g, b = torch.rand(64, dtype=torch.float64) + 0.5, torch.randn(64, dtype=torch.float64)
    rm, rv = torch.randn(64, dtype=torch.float64) * 0.1, torch.rand(64, dtype=torch.float64) + 0.5
    ref = F.relu(F.batch_norm(x, rm, rv, g, b, False, 0.1, 1e-5))
    y, _, _ = ops.bn_fwd(nhwc(x.float()), dev(g.float()), dev(b.float()), dev(rm.float()), dev(rv.float()), False, 1e-5, 0.1, None, True)
    close(y, ref.permute(0, 2, 3, 1), 1e-5)


@pytest.mark.parametrize("shape", [(2, 64, 16, 20), (1, 64, 17, 23), (5, 64, 32, 80)])
def test_maxpool_fwd_bwd(ops, shape):
    torch.manual_seed(3)
    x = torch.relu(torch.randn(shape, dtype=torch.float64)).requires_grad_(True)     # zeros -> ties, like after ReLU
    y = F.max_pool2d(x, 3, 2, 1)
    gy = torch.randn_like(y)
    y.backward(gy)
    yd, arg = ops.maxpool_fwd(nhwc(x.detach().float()))
    close(yd, y.permute(0, 2, 3, 1), 1e-7)
    dx = ops.maxpool_bwd(nhwc(gy.float()), arg, tuple(nhwc(x.detach().float()).shape))
    close(dx, x.grad.permute(0, 2, 3, 1), 1e-6)


@pytest.mark.parametrize("H,W,h,w", [(40, 100, 20, 50), (8, 20, 4, 10), (9, 21, 4, 10)])
def test_upsample_add_fwd_bwd(ops, H, W, h, w):
    fine = torch.randn(2, 64, H, W, dtype=torch.float64, requires_grad=True)
    coarse = torch.randn(2, 64, h, w, dtype=torch.float64, requires_grad=True)
    ref = fine + F.interpolate(coarse, size=(H, W), mode="nearest")
    g = torch.randn_like(ref)
    ref.backward(g)
    out = ops.upsample_add_(nhwc(fine.detach().float()), nhwc(coarse.detach().float()))
    close(out, ref.permute(0, 2, 3, 1), 1e-6)
    dc = ops.upsample_add_bwd_(nhwc(g.float()), torch.zeros(2, h, w, 64, device="cuda"))
    close(dc, coarse.grad.permute(0, 2, 3, 1), 1e-6)


def test_syncbn_split_path_equals_fused_path_on_one_rank(ops):
    """The SyncBatchNorm route (local sums -> all-reduce -> apply) with no process group must reproduce the fused route."""
    torch.manual_seed(5)
    x = torch.randn(3, 12, 14, 64, device="cuda") * 2 + 0.3
    g, b = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
    rm1, rv1 = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    rm2, rv2 = rm1.clone(), rv1.clone()
    res = torch.randn_like(x)
    y1, sm1, si1 = ops.bn_fwd(x, g, b, rm1, rv1, True, 1e-5, 0.1, res, True)
    y2, sm2, si2, total = ops.bn_fwd_sync(x, g, b, rm2, rv2, 1e-5, 0.1, res, True)
    assert total.is_cuda and total.dtype == torch.float64 and float(total) == x.numel() // 64     # the count never leaves the device
    close(y2, y1, 1e-5); close(sm2, sm1, 1e-6); close(si2, si1, 1e-5); close(rm2, rm1, 1e-6); close(rv2, rv1, 1e-5)
    dy = torch.randn_like(x)
    d1 = torch.zeros_like(x); d2 = torch.zeros_like(x)
    dx1, dg1, db1 = ops.bn_bwd(dy, x, y1, sm1, si1, g, True, d1)
    dx2, dg2, db2 = ops.bn_bwd_sync(dy, x, y2, sm2, si2, g, True, total, d2)
    close(dx2, dx1, 1e-5); close(dg2, dg1, 1e-5); close(db2, db1, 1e-5); close(d2, d1, 1e-7)
    # arena mode: parameter gradients are ADDED to the given destinations
    ag, ab = torch.ones(64, device="cuda"), torch.ones(64, device="cuda")
    ops.bn_bwd_sync(dy, x, y2, sm2, si2, g, True, total, None, dgamma=ag, dbeta=ab, param_accumulate=True)
    close(ag - 1, dg1, 1e-5); close(ab - 1, db1, 1e-5)


@pytest.mark.parametrize("n_valid,seed", [(3, 0), (4, 1), (1, 2), (0, 3), (2, 4)])
def test_lane_assign_matches_scipy_on_oracle_cost(ops, n_valid, seed):
    """Device-side cost matrix + exact matching vs the oracle's cost (dynamic_assign.py:128-185) + scipy Hungarian."""
    from oracle import phnet_cpu as O
    from tests import synth
    g = O.Geometry()
    r = np.random.default_rng(seed)
    tgt = synth.make_targets(g, 1, n_lanes=4)[0]
    order = r.permutation(4)
    for j in order[n_valid:]:
        tgt[j] = -1e5; tgt[j, 0] = 1; tgt[j, 1] = 0                         # invalid rows anywhere in the label block
    pri, _ = O.priors_from_embeddings(O.initial_anchor_embeddings(g), g)
    pred = pri.clone()
    pred[:, :2] = torch.from_numpy(r.normal(0, 1, (240, 2)).astype(np.float32))
    pred[:, 2:5] += torch.from_numpy(r.normal(0, 0.02, (240, 3)).astype(np.float32))
    pred[:, 5] = 0.6
    pred[:, 6:] += torch.from_numpy(r.normal(0, 0.01, (240, 36)).astype(np.float32))
    rows, srt, nv, cost = ops.lane_assign(dev(pred), dev(tgt), g.img_w, g.img_h, want_cost=True)
    assert int(nv) == n_valid
    vmask = tgt[:, 1] == 1
    if n_valid == 0:
        assert rows.cpu().tolist() == [-1] * 4 and srt.cpu().tolist() == [-1] * 4
        return
    ref_cost = O.assignment_cost(pred, tgt[vmask], g)
    close(cost.cpu()[:, vmask], ref_cost, 1e-4)
    assert torch.isinf(cost.cpu()[:, ~vmask]).all()
    rr, cc = O.hungarian(ref_cost)
    want = [-1] * 4
    cols = torch.where(vmask)[0]
    for a, b in zip(rr.tolist(), cc.tolist()):
        want[int(cols[b])] = a
    assert rows.cpu().tolist() == want
    assert srt.cpu().tolist() == sorted([w for w in want if w >= 0]) + [-1] * (4 - n_valid)


@pytest.mark.parametrize("rows,L,relu,res", [(240, 2304, True, True), (240, 2304, False, False), (8640, 128, True, False),
                                             (8640, 64, True, False), (240, 64, False, False), (240, 128, False, False), (7, 1000, True, True)])
def test_layernorm_fwd_bwd(ops, rows, L, relu, res):
    torch.manual_seed(rows + L)
    x = (torch.randn(rows, L, dtype=torch.float64) * 1.5 + 0.2).requires_grad_(True)
    w = (torch.rand(L, dtype=torch.float64) + 0.5).requires_grad_(True)
    b = torch.randn(L, dtype=torch.float64, requires_grad=True)
    r = torch.randn(rows, L, dtype=torch.float64, requires_grad=True) if res else None
    y = F.layer_norm(x, [L], w, b, 1e-5)
    if res:
        y = y + r
    if relu:
        y = F.relu(y)
    g = torch.randn_like(y)
    y.backward(g)
    yd, mean, rstd = ops.layernorm_fwd(dev(x.detach().float()), dev(w.detach().float()), dev(b.detach().float()), 1e-5,
                                       dev(r.detach().float()) if res else None, relu)
    close(yd, y, 1e-5)
    dx, dres, dw, db = ops.layernorm_bwd(dev(g.float()), dev(x.detach().float()), yd, dev(w.detach().float()), mean, rstd, relu, res)
    close(dx, x.grad, 2e-5); close(dw, w.grad, 2e-5); close(db, b.grad, 2e-5)
    if res:
        close(dres, r.grad, 1e-6)


@pytest.mark.parametrize("N,P,K,J", [(240, 36, 64, 128), (240, 36, 128, 64), (7, 36, 32, 64), (3, 5, 64, 32), (241, 20, 64, 128), (5, 16, 128, 64)])
def test_dynamic_head_bmm_layernorm_relu(ops, N, P, K, J):
    """relu(LayerNorm(x[n] @ w[n])) per anchor (dynamic_head.py:40-51) vs an fp64 torch reference."""
    torch.manual_seed(N + K)
    x = torch.randn(N, P, K, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(N, K, J, dtype=torch.float64) / K ** 0.5).requires_grad_(True)
    ga = (torch.rand(J, dtype=torch.float64) + 0.5).requires_grad_(True)
    be = torch.randn(J, dtype=torch.float64, requires_grad=True)
    y = F.relu(F.layer_norm(torch.bmm(x, w), [J], ga, be, 1e-5))
    g = torch.randn_like(y)
    y.backward(g)
    xd, wd, gd, bd = (dev(t.detach().float()) for t in (x, w, ga, be))
    yd, stats = ops.dyn_bmm_ln_relu_fwd(xd, wd, gd, bd, 1e-5)
    close(yd, y, 2e-5)
    y_inf, none = ops.dyn_bmm_ln_relu_fwd(xd, wd, gd, bd, 1e-5, save_stats=False)
    assert none is None and torch.equal(y_inf, yd)
    from phnet_amd._lib import lib
    if lib().phnet_dyn_mfma_applies(P, K, J):
        # matrix-pipe forward: one wavefront per (anchor, 16-row fragment) by default, per anchor with the switch - the same bits
        assert lib().phnet_tune_dyn_mfma(3) == 0
        try:
            y_a, st_a = ops.dyn_bmm_ln_relu_fwd(xd, wd, gd, bd, 1e-5)
        finally:
            assert lib().phnet_tune_dyn_mfma(1) == 0
        assert torch.equal(y_a, yd) and torch.equal(st_a, stats)
    dx, dw, dg, db = ops.dyn_bmm_ln_relu_bwd(dev(g.float()), xd, wd, yd, stats, gd, 1e-5)
    close(dx, x.grad, 5e-5); close(dw, w.grad, 5e-5); close(dg, ga.grad, 5e-5); close(db, be.grad, 5e-5)
    if lib().phnet_dyn_mfma_applies(P, K, J):
        # matrix-pipe backward: four wavefronts per anchor by default, one with the switch - the same arithmetic up to the compiler's
        # contraction of the LayerNorm-backward expressions and the order in which the row fragments' affine sums are folded
        assert lib().phnet_tune_dyn_mfma(3) == 0
        try:
            dx_a, dw_a, dg_a, db_a = ops.dyn_bmm_ln_relu_bwd(dev(g.float()), xd, wd, yd, stats, gd, 1e-5)
        finally:
            assert lib().phnet_tune_dyn_mfma(1) == 0
        close(dx_a, dx, 2e-6); close(dw_a, dw, 2e-6); close(dg_a, dg, 2e-6); close(db_a, db, 2e-6)
    # accumulate mode adds to the destinations; dx may be skipped
    acc_g, acc_b = torch.ones_like(gd), torch.ones_like(gd)
    dx2, dw2, _, _ = ops.dyn_bmm_ln_relu_bwd(dev(g.float()), xd, wd, yd, stats, gd, 1e-5, need_dx=False,
                                             dgamma=acc_g, dbeta=acc_b, accumulate=True)
    assert dx2 is None and torch.equal(dw2, dw)
    close(acc_g - 1.0, ga.grad, 5e-5); close(acc_b - 1.0, be.grad, 5e-5)


def test_dwconv3x3_fwd_bwd(ops):
    torch.manual_seed(9)
    N, C, P = 240, 64, 36
    x = torch.randn(1, N, C, P, dtype=torch.float64, requires_grad=True)
    w = torch.randn(N, 1, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.randn(N, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, b, padding=1, groups=N)
    g = torch.randn_like(y)
    y.backward(g)
    xd, wd = dev(x.detach().float()[0]), dev(w.detach().float())
    close(ops.dwconv3x3(xd, wd, dev(b.detach().float())), y[0], 1e-5)
    close(ops.dwconv3x3(dev(g.float()[0]), wd, None, flip=True), x.grad[0], 1e-5)
    dw, db = ops.dwconv3x3_wgrad(dev(g.float()[0]), xd)
    close(dw, w.grad, 2e-5); close(db, b.grad, 2e-5)


@pytest.mark.parametrize("Lq,Lk,masked,drop", [(240, 240, False, False), (240, 40, True, False), (240, 5, True, False),
                                                (240, 40, True, True), (240, 240, False, True), (7, 3, False, False)])
def test_fused_attention_fwd_bwd(ops, Lq, Lk, masked, drop):
    torch.manual_seed(Lq + Lk)
    H, E = 8, 128
    q = torch.randn(Lq, E, dtype=torch.float64, requires_grad=True)
    k = torch.randn(Lk, E, dtype=torch.float64, requires_grad=True)
    v = torch.randn(Lk, E, dtype=torch.float64, requires_grad=True)
    valid = torch.ones(Lk, dtype=torch.bool)
    if masked:
        valid[torch.randperm(Lk)[: max(1, Lk // 3)]] = False
        valid[-1] = True
    keep = (torch.rand(H, Lq, Lk) >= 0.1) if drop else None
    scale = 1.0 / 0.9 if drop else 1.0
    qh = q.view(Lq, H, 16).transpose(0, 1) * 0.25
    kh, vh = k.view(Lk, H, 16).transpose(0, 1), v.view(Lk, H, 16).transpose(0, 1)
    logits = (qh @ kh.transpose(1, 2)).masked_fill(~valid[None, None, :], float("-inf"))
    att = torch.softmax(logits, dim=-1)
    if drop:
        att = att * keep.double() * scale
    ref = (att @ vh).transpose(0, 1).reshape(Lq, E)
    g = torch.randn_like(ref)
    ref.backward(g)
    qd, kd, vd = dev(q.detach().float()), dev(k.detach().float()), dev(v.detach().float())
    vu8 = valid.to(torch.uint8).cuda() if masked else None
    ku8 = keep.to(torch.uint8).cuda().contiguous() if drop else None
    out, lse = ops.attention_fwd(qd, kd, vd, H, vu8, ku8, scale)
    close(out, ref, 2e-5)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attention_bwd(qd, kd, vd, out, dev(g.float()), lse, H, dq, dk, dv, vu8, ku8, scale)
    close(dq, q.grad, 5e-5); close(dk, k.grad, 5e-5); close(dv, v.grad, 5e-5)
    # packed self-attention layout: q|k|v column blocks of one [L,3E] buffer, gradients into a packed buffer
    if Lq == Lk and not masked:
        pk = torch.cat([qd, kd, vd], dim=1).contiguous()
        o2, lse2 = ops.attention_fwd(pk[:, :E], pk[:, E:2 * E], pk[:, 2 * E:], H, None, ku8, scale)
        close(o2, ref, 2e-5)
        dpk = torch.zeros_like(pk)
        ops.attention_bwd(pk[:, :E], pk[:, E:2 * E], pk[:, 2 * E:], o2, dev(g.float()), lse2, H, dpk[:, :E], dpk[:, E:2 * E], dpk[:, 2 * E:], None, ku8, scale)
        close(dpk[:, :E], q.grad, 5e-5); close(dpk[:, E:2 * E], k.grad, 5e-5); close(dpk[:, 2 * E:], v.grad, 5e-5)


def _rng(state_value, call, p):
    return (torch.tensor([state_value], dtype=torch.int64, device="cuda"), call, p)


def test_attention_in_kernel_dropout_matches_explicit_mask(ops):
    """The counter-based dropout of the attention kernels: recover the mask it drew (one-hot V exposes the dropped attention
    weights), then the explicit-mask path - itself checked against torch above - must reproduce forward and backward."""
    torch.manual_seed(5)
    H, E, Lq, Lk, p = 8, 128, 240, 16, 0.25
    q, k = torch.randn(Lq, E, device="cuda"), torch.randn(Lk, E, device="cuda")
    onehot = torch.eye(16, device="cuda").repeat(1, H).contiguous()                  # V[k, h*16+d] = (k == d)
    rng = _rng(123456789, 7, p)
    probe, _ = ops.attention_fwd(q, k, onehot, H, rng=rng)                           # probe[q, h*16+k] = dropped weight (h,q,k)
    keep = (probe.view(Lq, H, Lk) > 0).permute(1, 0, 2).contiguous().to(torch.uint8)
    frac = float(keep.float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac
    per_head = keep.float().mean(dim=(1, 2))
    assert float((per_head - (1 - p)).abs().max()) < 0.05
    v, g = torch.randn(Lk, E, device="cuda"), torch.randn(Lq, E, device="cuda")
    o1, l1 = ops.attention_fwd(q, k, v, H, rng=rng)
    o2, l2 = ops.attention_fwd(q, k, v, H, None, keep, 1.0 / (1.0 - p))
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    grads = []
    for kw in (dict(rng=rng), dict(keep=keep, keep_scale=1.0 / (1.0 - p))):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attention_bwd(q, k, v, o1, g, l1, H, dq, dk, dv, **kw)
        grads.append((dq, dk, dv))
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    # another step counter or another site id draws another mask; p = 0 draws none
    o3, _ = ops.attention_fwd(q, k, v, H, rng=_rng(123456790, 7, p))
    o4, _ = ops.attention_fwd(q, k, v, H, rng=_rng(123456789, 8, p))
    o5, _ = ops.attention_fwd(q, k, v, H, rng=_rng(123456789, 7, 0.0))
    o6, _ = ops.attention_fwd(q, k, v, H)
    assert not torch.equal(o1, o3) and not torch.equal(o1, o4) and torch.equal(o5, o6)


def test_flat_adamw_matches_torch_adamw(ops):
    """phnet_amd.optim.FlatAdamW (one launch over the flat arenas) against torch.optim.AdamW with the reference's grouping
    (weight decay on matrices only), including a channels_last conv weight whose flat view keeps the OHWI strides."""
    from phnet_amd.optim import FlatAdamW, split_decay

    def make():
        torch.manual_seed(4)
        net = torch.nn.Sequential(torch.nn.Conv2d(8, 12, 3, bias=False), torch.nn.BatchNorm2d(12), torch.nn.Flatten(), torch.nn.Linear(12 * 36, 7)).cuda()      # (a conv bias in front of BN would have a pure-noise gradient, which Adam amplifies to +-lr)
        net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
        return net
    ref, net = make(), make()
    decay, no_decay = split_decay(ref)
    kw = dict(lr=5e-3, betas=(0.9, 0.99), eps=1e-8)
    topt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.05}, {"params": no_decay, "weight_decay": 0.0}], **kw)
    fopt, arena = FlatAdamW.for_model(net, weight_decay=0.05, **kw)
    try:
        assert not net[0].weight.is_contiguous() and net[0].weight.is_contiguous(memory_format=torch.channels_last)
        assert (arena.numel + 3) // 4 * 4 == arena.flat.numel() == arena.flat_params.numel()
        for step in range(4):
            x = torch.randn(5, 8, 8, 8, device="cuda")
            topt.zero_grad(set_to_none=True)
            fopt.zero_grad()
            ref(x).square().mean().backward()
            loss = net(x).square().mean()
            grads = torch.autograd.grad(loss, list(net.parameters()))
            for p, gr in zip(net.parameters(), grads):
                p.grad.add_(gr)                                      # what the HIP backward kernels do: accumulate into the arena
            topt.step()
            fopt.step()
            for (k, a), b in zip(ref.named_parameters(), net.parameters()):
                close(b, a, 2e-6)
        assert int(fopt.step_count) == 4
        # the caller step of the reference around it: per-iteration cosine schedule (trainOL.py:121-124,228) and GradScaler
        # (trainOL.py:225-227); the last two steps replay ONE captured optimizer launch - the schedule reaches it through lr_dev
        tsch = torch.optim.lr_scheduler.CosineAnnealingLR(topt, T_max=10, eta_min=1e-5)
        fsch = torch.optim.lr_scheduler.CosineAnnealingLR(fopt, T_max=10, eta_min=1e-5)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
        graph = None
        for step in range(5):
            x = torch.randn(5, 8, 8, 8, device="cuda")
            topt.zero_grad(set_to_none=True)
            fopt.zero_grad()
            ref(x).square().mean().backward()
            if step < 2:                                              # eager, through the GradScaler
                scaler.scale(net(x).square().mean()).backward()
                scaler.step(fopt)
                scaler.update()
            else:
                net(x).square().mean().backward()
                if graph is None:
                    st = torch.cuda.Stream()
                    st.wait_stream(torch.cuda.current_stream())
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(st):
                        with torch.cuda.graph(graph, stream=st):
                            fopt.step()
                    torch.cuda.current_stream().wait_stream(st)
                fopt.sync_lr()
                graph.replay()
            topt.step()
            tsch.step(); fsch.step()
            assert abs(fopt.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) < 1e-12
            for (k, a), b in zip(ref.named_parameters(), net.parameters()):
                close(b, a, 5e-6)
        assert int(fopt.step_count) == 9
        # ---- checkpoints travel in torch.optim.AdamW's own layout, both ways (trainOL.py:128,182) --------------------------
        tsd, fsd = topt.state_dict(), fopt.state_dict()
        assert set(fsd) == {"state", "param_groups"} and sorted(fsd["state"]) == sorted(tsd["state"])
        assert [g["params"] for g in fsd["param_groups"]] == [g["params"] for g in tsd["param_groups"]]
        for i in tsd["state"]:
            assert float(fsd["state"][i]["step"]) == float(tsd["state"][i]["step"]) == 9
            close(fsd["state"][i]["exp_avg"], tsd["state"][i]["exp_avg"], 1e-5)
            close(fsd["state"][i]["exp_avg_sq"], tsd["state"][i]["exp_avg_sq"], 1e-5)
        # (a) a torch.optim.AdamW built on our parameters resumes from OUR checkpoint and (b) a fresh FlatAdamW resumes from
        # TORCH's checkpoint: both then take the same next step as the original torch optimizer
        net_b = make()
        net_b.load_state_dict(ref.state_dict())
        fopt_b, arena_b = FlatAdamW.for_model(net_b, lr=1.0, betas=(0.5, 0.5), eps=1e-3, weight_decay=0.5)   # all overwritten by the load
        net_c = make()
        net_c.load_state_dict(ref.state_dict())
        dc, ndc = split_decay(net_c)
        topt_c = torch.optim.AdamW([{"params": dc, "weight_decay": 0.9}, {"params": ndc, "weight_decay": 0.0}], lr=1.0)
        try:
            fopt_b.load_state_dict(tsd)
            topt_c.load_state_dict(fsd)
            assert abs(fopt_b.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) < 1e-12 and fopt_b.param_groups[0]["weight_decay"] == 0.05
            assert tuple(fopt_b.param_groups[0]["betas"]) == (0.9, 0.99) and int(fopt_b.step_count) == 9
            x = torch.randn(5, 8, 8, 8, device="cuda")
            topt.zero_grad(set_to_none=True); fopt_b.zero_grad(); topt_c.zero_grad(set_to_none=True)
            ref(x).square().mean().backward()
            net_b(x).square().mean().backward()
            net_c(x).square().mean().backward()
            topt.step(); fopt_b.step(); topt_c.step()
            for a, b, c in zip(ref.parameters(), net_b.parameters(), net_c.parameters()):
                close(b, a, 5e-6); close(c, a, 5e-6)
            # hyper-parameters are read from param_groups at every step; what the one-launch kernel cannot honour is refused
            fopt_b.param_groups[1]["weight_decay"] = 0.1
            with pytest.raises(ValueError):
                fopt_b.step()
        finally:
            arena_b.release()
    finally:
        arena.release()


@pytest.mark.parametrize("L", [128, 64, 200])
def test_dropout_add_layernorm_fused(ops, L):
    """t = res + dropout(x), h = LN(t): exact vs torch at p = 0 (values and all gradients, with and without a gradient on
    the residual stream); at p > 0 the backward redraws the forward's mask and kept entries carry the 1/(1-p) scale."""
    torch.manual_seed(L)
    rows = 240
    x = torch.randn(rows, L, dtype=torch.float64, requires_grad=True)
    res = torch.randn(rows, L, dtype=torch.float64, requires_grad=True)
    w = (torch.rand(L, dtype=torch.float64) + 0.5).requires_grad_(True)
    b = torch.randn(L, dtype=torch.float64, requires_grad=True)
    t_ref = res + x
    h_ref = F.layer_norm(t_ref, [L], w, b, 1e-5)
    dt, dh = torch.randn_like(t_ref), torch.randn_like(h_ref)
    (t_ref * dt).sum().add((h_ref * dh).sum()).backward()
    xd, rd, wd, bd = (dev(v.detach().float()) for v in (x, res, w, b))
    t, h, mean, rstd = ops.dropout_add_ln_fwd(xd, rd, wd, bd, 1e-5)
    close(t, t_ref, 1e-6); close(h, h_ref, 2e-5)
    dres, dx, dw, db = ops.dropout_add_ln_bwd(dev(dh.float()), dev(dt.float()), t, wd, mean, rstd)
    close(dres, res.grad, 3e-5); close(dx, x.grad, 3e-5); close(dw, w.grad, 3e-5); close(db, b.grad, 3e-5)
    dres0, dx0, _, _ = ops.dropout_add_ln_bwd(dev(dh.float()), None, t, wd, mean, rstd)
    close(dres0, res.grad - dt, 3e-5)
    assert torch.equal(dres0, dx0)
    p = 0.3
    rng = _rng(77, 5, p)
    t2, h2, mean2, rstd2 = ops.dropout_add_ln_fwd(xd, rd, wd, bd, 1e-5, rng)
    kept = (t2 - rd) != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.02
    close((t2 - rd)[kept], (xd / (1 - p))[kept], 1e-5)
    close(h2, F.layer_norm(t2.double().cpu(), [L], w.detach(), b.detach(), 1e-5), 2e-5)
    dres2, dx2, _, _ = ops.dropout_add_ln_bwd(dev(dh.float()), dev(dt.float()), t2, wd, mean2, rstd2, rng)
    assert torch.equal(dx2 != 0, kept & (dres2 != 0))
    close(dx2[kept], (dres2 / (1 - p))[kept], 1e-5)


@pytest.mark.parametrize("M,K,N", [(240, 128, 384), (240, 192, 44), (240, 256, 128), (37, 64, 64), (256, 2304, 64), (240, 2304, 576)])
def test_linear_backward_fused_launch(ops, M, K, N):
    """dx, dW (accumulated) and dbias of a few-rows Linear layer from ONE launch vs fp64 torch."""
    assert ops.linear_bwd_fusable(M, K, N)
    assert not ops.linear_bwd_fusable(300, K, N) and not ops.linear_bwd_fusable(M, K, 1024)
    torch.manual_seed(M + K + N)
    x = torch.randn(M, K, dtype=torch.float64)
    w = torch.randn(N, K, dtype=torch.float64) / K ** 0.5
    dy = torch.randn(M, N, dtype=torch.float64)
    dw0, db0 = torch.randn(N, K, dtype=torch.float64), torch.randn(N, dtype=torch.float64)
    dwd, dbd = dev(dw0.float()), dev(db0.float())
    dx = ops.linear_bwd(dev(dy.float()), dev(x.float()), dev(w.float()), dwd, dbd, accumulate=True)
    close(dx, dy @ w, 3e-5)
    close(dwd, dw0 + dy.t() @ x, 3e-5)
    close(dbd, db0 + dy.sum(0), 3e-5)
    dw1 = torch.empty_like(dwd)
    dx1 = ops.linear_bwd(dev(dy.float()), dev(x.float()), dev(w.float()), dw1, None, accumulate=False)
    assert torch.equal(dx1, dx)
    close(dw1, dy.t() @ x, 3e-5)
    # layer that ended in a ReLU: the mask of its saved output is applied while dy is staged
    y = torch.relu(torch.randn(M, N, dtype=torch.float64))
    gm = dy * (y > 0)
    dw2, db2 = torch.empty_like(dwd), torch.empty_like(dbd)
    dx2 = ops.linear_bwd(dev(dy.float()), dev(x.float()), dev(w.float()), dw2, db2, accumulate=False, relu_y=dev(y.float()))
    close(dx2, gm @ w, 3e-5); close(dw2, gm.t() @ x, 3e-5); close(db2, gm.sum(0), 3e-5)


def test_memory_tokens_gate_tail_blend(ops):
    """The three small fused pieces of the per-frame loop against their tensor-op definitions."""
    torch.manual_seed(21)
    N, E = 240, 128
    feat = torch.randn(N, 1, E, device="cuda")
    for rows in ([3, 17, 200, -1, -1, -1, -1, -1], [-1] * 8, [0, 1, 2, 3, 4, 5, 6, 239], []):
        r = torch.tensor(rows, dtype=torch.int64, device="cuda")
        tok, valid = ops.memory_tokens(feat, r)
        ok = r >= 0
        pos = feat[r.clamp(min=0)] * ok[:, None, None].float()
        rest = (feat.double().sum(0, keepdim=True) - pos.double().sum(0, keepdim=True)) / (N - int(ok.sum()))
        assert valid.dtype == torch.bool and valid.tolist() == ok.tolist() + [True]
        assert torch.equal(tok[:-1], pos)
        close(tok[-1:], rest, 1e-5)
    K_ = 576
    h = torch.randn(N, K_, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(1, K_, dtype=torch.float64) / K_ ** 0.5).requires_grad_(True)
    b = torch.tensor([0.1], dtype=torch.float64, requires_grad=True)
    ref = torch.sigmoid(F.relu(F.linear(h, w, b)))[:, 0]
    g = torch.randn_like(ref)
    ref.backward(g)
    hd, wd, bd = dev(h.detach().float()), dev(w.detach().float().view(-1)), dev(b.detach().float())
    out = ops.gate_tail_fwd(hd, wd, bd)
    close(out, ref, 1e-6)
    assert float((out == 0.5).float().mean()) > 0.2                       # closed ReLUs are part of the case
    dh, dw, db = ops.gate_tail_bwd(dev(g.float()), out, hd, wd)
    close(dh, h.grad, 1e-6); close(dw, w.grad.view(-1), 1e-5); close(db, b.grad, 1e-5)
    aw, ab = torch.ones_like(wd), torch.ones_like(bd)
    dh2, _, _ = ops.gate_tail_bwd(dev(g.float()), out, hd, wd, need_dh=False, dw=aw, db=ab, accumulate=True)
    assert dh2 is None
    close(aw - 1, w.grad.view(-1), 1e-5); close(ab - 1, b.grad, 1e-5)
    W, P = 78, 36
    gate = torch.rand(1, N, 1, device="cuda")
    a, bl = torch.randn(1, N, W, device="cuda"), torch.randn(1, N, W, device="cuda")
    idx = (torch.linspace(0, 1, steps=P) * 71).long().cuda()
    pri, on_map = ops.blend_priors(gate, a, bl, idx)
    want = (1 - gate) * a + gate * bl
    close(pri, want, 1e-6)
    assert torch.equal(on_map, pri[..., 6 + idx])


@pytest.mark.parametrize("p", [0.0, 0.1, 0.5])
def test_dropout_add_and_gelu_dropout(ops, p):
    """res + dropout(x) and dropout(gelu(x)) (transformer.py:275-298): exact at p = 0, and at p > 0 the forward and backward
    masks agree, the keep rate is 1-p and kept values carry the 1/(1-p) scale."""
    torch.manual_seed(11)
    n = 240 * 256
    x, res, dy = (torch.randn(n, device="cuda") for _ in range(3))
    rng = _rng(42, 3, p) if p > 0 else None
    y = ops.dropout_add(x, res, rng)
    d = y - res
    kept = d != 0
    if p == 0:
        assert torch.equal(y, res + x)
    else:
        assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
        close(d[kept], x[kept] / (1 - p), 1e-6)
        dx = ops.dropout_add(dy, None, rng)
        assert torch.equal(dx != 0, kept)
        close(dx[kept], dy[kept] / (1 - p), 1e-6)
        assert not torch.equal(ops.dropout_add(x, res, _rng(43, 3, p)), y)
    xg = x.double().cpu().requires_grad_(True)
    ref = F.gelu(xg)
    ref.backward(dy.double().cpu())
    yg = ops.gelu_dropout_fwd(x, rng)
    dxg = ops.gelu_dropout_bwd(dy, x, rng)
    if p == 0:
        close(yg, ref, 1e-6); close(dxg, xg.grad, 1e-6)
    else:
        m = ops.dropout_add(torch.ones_like(x), None, rng) != 0                      # same site -> same bits
        close(yg[m], (ref.detach() / (1 - p)).float().cuda()[m], 1e-6)
        close(dxg[m], (xg.grad / (1 - p)).float().cuda()[m], 1e-6)
        assert float(yg[~m].abs().max()) == 0.0 and float(dxg[~m].abs().max()) == 0.0


@pytest.mark.parametrize("seed,thr", [(0, 0.5), (1, 0.5), (2, 0.9), (3, 0.999999), (4, 0.0)])
def test_lane_decode_matches_oracle_decode(ops, seed, thr):
    """Fused decode vs the oracle's decode_frame (softmax threshold + NMS rows + oracle NMS + gather)."""
    from oracle import lane_nms as ON
    from oracle import phnet_cpu as O
    g = O.Geometry(conf_threshold=thr)
    r = np.random.default_rng(seed)
    pri, _ = O.priors_from_embeddings(O.initial_anchor_embeddings(g), g)
    lines = pri.clone()
    lines[:, :2] = torch.from_numpy(r.normal(0, 2, (240, 2)).astype(np.float32))
    lines[:, 2:5] += torch.from_numpy(r.normal(0, 0.02, (240, 3)).astype(np.float32))
    lines[:, 5] = torch.from_numpy(r.uniform(0.2, 0.9, 240).astype(np.float32))
    lines[:, 6:] += torch.from_numpy(r.normal(0, 0.01, (240, 36)).astype(np.float32))
    ref = O.decode_frame(lines.clone(), g, ON.lane_nms)
    out = ops.lane_decode(dev(lines), thr, g.nms_thres, g.max_lanes, g.img_w)
    assert (out["keep_mask"].cpu().numpy().astype(bool) == ref["keep_inds"].numpy()).all()
    n = int(out["num"])
    assert n == len(ref["keep"])
    assert out["keep_c"][:n].cpu().tolist() == ref["keep"].tolist()
    want_anchor = torch.where(ref["keep_inds"])[0][ref["keep"]].tolist()
    assert out["anchors"][:n].cpu().tolist() == want_anchor
    assert out["anchors_sorted"][:n].cpu().tolist() == sorted(want_anchor)
    assert out["anchors"][n:].cpu().tolist() == [-1] * (4 - n)
    if n:
        assert torch.equal(out["kept_rows"][:n].cpu(), ref["kept_rows"])


# ------------------------------------------------------------------------------------------------ routing gate stack
def _gate_reference(x, params, eps, kink=None):
    """libs/models/Router.py:72-75 spelled in fp64 tensor ops: pre_norm, then 4 x relu(DWblock(s) + s) with
    DWblock = dwconv3x3 -> LN([C,P]) -> ReLU -> dwconv3x3 -> LN([C,P]); x [B,N,C,P], per-anchor filters [N,1,3,3].
    kink (optional list): receives, per ReLU, the smallest |pre-activation| of every plane [B,N]."""
    n, (c, p) = x.shape[1], x.shape[2:]
    s = F.layer_norm(x, (c, p), params[0], params[1], eps)
    for b in range(4):
        w1, b1, g1, e1, w2, b2, g2, e2 = params[2 + 8 * b: 10 + 8 * b]
        t = F.conv2d(s, w1, b1, padding=1, groups=n)
        t = F.layer_norm(t, (c, p), g1, e1, eps)
        if kink is not None:
            kink.append(t.detach().abs().amin(dim=(2, 3)))
        t = F.conv2d(torch.relu(t), w2, b2, padding=1, groups=n)
        t = F.layer_norm(t, (c, p), g2, e2, eps) + s
        if kink is not None:
            kink.append(t.detach().abs().amin(dim=(2, 3)))
        s = torch.relu(t)
    return s


def _gate_params(N, C, P, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=g, dtype=torch.float64) * scale       # noqa: E731
    params = [1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P)]
    for _ in range(4):
        params += [rn(N, 1, 3, 3, scale=0.4), 0.1 * rn(N), 1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P),
                   rn(N, 1, 3, 3, scale=0.4), 0.1 * rn(N), 1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P)]
    return [t.requires_grad_(True) for t in params]


@pytest.mark.parametrize("B,N,C,P", [(1, 240, 64, 36), (5, 240, 64, 36), (2, 7, 16, 24), (1, 3, 64, 36)])
def test_gate_stack_fwd_bwd_vs_fp64_statement(ops, B, N, C, P):
    """phnet_gate_stack_fwd / _bwd (+ gate_ln_grad_reduce and, for B > 1, the per-plane filter-gradient reduce) against the
    reference's gate stack (Router.py:39-63,72-75) in fp64 autograd: output, and the gradient of all 34 parameters, both as an
    overwrite and accumulated onto existing values (the gradient-arena mode).  B > 1 = planes of several frames sharing the
    per-anchor filters (stage 0 of a clip)."""
    torch.manual_seed(100 + N)
    eps = 1e-5
    x = torch.randn(B, N, C, P, dtype=torch.float64)
    params = _gate_params(N, C, P, seed=N + C)
    kink = []
    ref = _gate_reference(x, params, eps, kink)
    gout = torch.randn_like(ref)
    # A plane with a pre-activation within fp32 rounding of a ReLU kink has no well-defined fp32 gradient: whichever side an
    # implementation's rounding lands on switches that element's whole gradient path (on the 5 x 240 case the fp64 statement
    # holds pre-activations of 4e-8 and 1.2e-7; an LN output carries ~3e-7 of fp32 rounding).  Such planes (a few per thousand
    # at this margin) get a zero upstream gradient, so the comparison is decided by arithmetic, not by the luck of a rounding.
    at_kink = torch.stack(kink).amin(dim=0) < 3e-6                          # [B,N]
    assert float(at_kink.double().mean()) < 0.05
    gout[at_kink] = 0.0
    ref.backward(gout)
    xd = dev(x.float().reshape(B * N, C, P))
    pd = [dev(t.detach().float()) for t in params]
    out, saved = ops.gate_stack_fwd(xd, pd, eps, True, anchors=N)
    close(out.view(B, N, C, P), ref, 2e-5)
    out_inf, none = ops.gate_stack_fwd(xd, pd, eps, False, anchors=N)
    assert none is None and torch.equal(out_inf, out)                        # inference launch = same arithmetic
    gd = dev(gout.float().reshape(B * N, C, P))
    grads = [torch.full_like(t, float("nan")) for t in pd]
    ops.gate_stack_bwd(gd, xd, out, pd, saved, grads, eps, False, anchors=N)
    torch.cuda.synchronize()
    # the depth-wise conv biases sit in front of a LayerNorm over the whole [C,P] plane, which removes a constant shift: their
    # true gradient is ZERO (fp64 autograd: 1e-13) and what any fp32 implementation returns is the rounding noise of a
    # C*P-term sum of O(1) values - judged against that noise floor, everything else against 5e-5 of the tensor's scale
    noise = 2e-7 * C * P * B
    is_conv_bias = lambda i: i >= 2 and (i - 2) % 4 == 1                                      # noqa: E731
    for i, (got, p) in enumerate(zip(grads, params)):
        assert torch.isfinite(got).all(), i
        if is_conv_bias(i):
            assert float(p.grad.abs().max()) < 1e-9 and float(got.abs().max()) <= noise, (i, float(got.abs().max()), noise)
        else:
            close(got.view(p.shape), p.grad, 5e-5, f"parameter {i} (overwrite)")
    acc = [torch.ones_like(t) for t in pd]
    ops.gate_stack_bwd(gd, xd, out, pd, saved, acc, eps, True, anchors=N)
    for i, (got, p) in enumerate(zip(acc, params)):
        if is_conv_bias(i):
            assert float((got - 1.0).abs().max()) <= noise + 1e-6, i
        else:
            close(got.view(p.shape) - 1.0, p.grad, 5e-5, f"parameter {i} (accumulate)")


def test_gate_stack_through_the_module_matches_reference_gate(ops):
    """AdaptiveRouter4Lane.forward (gate stack + Linear 2304->576 + ReLU + Linear 576->1 + ReLU + sigmoid, Router.py:72-81)
    against the same module spelled with torch.nn in fp64, incl. gradients of the MLP and the stack parameters."""
    from phnet_amd.libs.models.Router import AdaptiveRouter4Lane
    torch.manual_seed(5)
    N, C, P = 240, 64, 36
    gate = AdaptiveRouter4Lane(num_priors=N, features_channels=C, num_points=P, stages=3)
    with torch.no_grad():
        for name, p in gate.named_parameters():
            if not name.startswith("layers."):                              # LayerNorm affine and depth-wise filters off their init
                p.add_(0.1 * torch.randn_like(p))
    ref = __import__("copy").deepcopy(gate).double()
    gate = gate.cuda()
    x = torch.randn(2, N, C, P)
    stage = 1
    s = ref.pre_norm[stage](x.double())
    for blk in ref.DWNets[stage]:
        s = torch.relu(blk(s) + s)
    want = torch.sigmoid(ref.layers[stage](s.flatten(2)))
    got = gate(x.cuda(), stage)
    close(got, want, 1e-5)
    g = torch.randn_like(want)
    want.backward(g)
    got.backward(g.float().cuda())
    have = dict(gate.named_parameters())
    for k, p in ref.named_parameters():
        if p.grad is None:
            assert have[k].grad is None, k
            continue
        if ".DWNets." in k and k.endswith((".0.bias", ".3.bias")):           # conv bias in front of a LayerNorm: true gradient 0
            assert float(p.grad.abs().max()) < 1e-9 and float(have[k].grad.abs().max()) <= 2e-7 * C * P * 2 * 10, k
            continue
        close(have[k].grad, p.grad, 1e-4)


# ------------------------------------------------------------------------------------------------ lane prior update
def _lane_update_reference(priors, head, ys, img_w, img_h, S):
    """libs/models/Router4OL.py:329-344 in fp64 (out-of-place): returns (predictions, prediction_lines)."""
    import math
    cls, reg, off = head[:, :2], head[:, 2:6], head[:, 6:6 + S]
    sy, sx, th = (priors[:, 2 + i] + torch.tanh(reg[:, i]) for i in range(3))
    xs = (sx[:, None] * (img_w - 1) + ((1 - ys[None, :] - sy[:, None]) * img_h / torch.tan(th[:, None] * math.pi + 1e-5))) / (img_w - 1)
    lines = torch.cat([cls, sy[:, None], sx[:, None], th[:, None], reg[:, 3:4], xs], dim=1)
    preds = torch.cat([lines[:, :6], xs + off], dim=1)
    return preds, lines


@pytest.mark.parametrize("N,S,HW", [(240, 36, 44), (240, 72, 80), (1200, 36, 44), (5, 36, 42)])
def test_lane_update_fwd_bwd_vs_fp64_statement(ops, N, S, HW):
    """phnet_lane_update_fwd / _bwd against the reference's prior update (tanh on start/theta, length replaced, xs recomputed
    through 1/tan(theta*pi + 1e-5), offsets added to the output copy only).  theta is kept off tan's poles (anchors within
    1e-2 of horizontal are noise-dominated in any fp32 implementation); gradients for every combination of upstream
    gradients the autograd node sees (preds only, lines only, both)."""
    g = torch.Generator().manual_seed(N + S)
    img_w, img_h = 800.0, 320.0
    priors = torch.zeros(N, 6 + S, dtype=torch.float64)
    priors[:, 2] = torch.rand(N, generator=g, dtype=torch.float64) * 0.8
    priors[:, 3] = torch.rand(N, generator=g, dtype=torch.float64)
    priors[:, 4] = 0.12 + 0.76 * torch.rand(N, generator=g, dtype=torch.float64)           # theta*pi in [0.38, 2.76]
    priors[:, 6:] = torch.randn(N, S, generator=g, dtype=torch.float64)                    # overwritten by the update
    head = torch.randn(N, HW, generator=g, dtype=torch.float64) * 0.5
    head[:, 4] *= 0.05                                                                     # keep theta + tanh(.) inside (0.05, 0.95)
    ys = torch.linspace(1, 0, S, dtype=torch.float64)
    pr, hd = priors.clone().requires_grad_(True), head.clone().requires_grad_(True)
    preds, lines = _lane_update_reference(pr, hd, ys, img_w, img_h, S)
    pd, hdv, ysd = dev(priors.float()), dev(head.float()), dev(ys.float())
    got_p, got_l = ops.lane_update_fwd(pd, hdv, ysd, img_w, img_h)
    close(got_p, preds, 2e-5); close(got_l, lines, 2e-5)
    assert torch.equal(got_p[:, :6], got_l[:, :6])
    gp, gl = torch.randn_like(preds), torch.randn_like(lines)
    for use_p, use_l in ((True, False), (False, True), (True, True)):
        pr.grad = hd.grad = None
        loss = (preds * gp).sum() * use_p + (lines * gl).sum() * use_l
        loss.backward(retain_graph=True)
        dhead, dpri = ops.lane_update_bwd(dev(gp.float()) if use_p else None, dev(gl.float()) if use_l else None,
                                          got_l, hdv, ysd, img_w, img_h, True)
        close(dhead[:, :6 + S], hd.grad[:, :6 + S], 5e-5)
        assert float(dhead[:, 6 + S:].abs().max()) == 0.0 if HW > 6 + S else True
        close(dpri, pr.grad, 5e-5)
        dhead2, none = ops.lane_update_bwd(dev(gp.float()) if use_p else None, dev(gl.float()) if use_l else None,
                                           got_l, hdv, ysd, img_w, img_h, False)
        assert none is None and torch.equal(dhead2, dhead)


def test_dropout_masks_per_item_batched_equals_item_by_item():
    """csrc/common.h "Items": a launch over a batch of items (item_rows = rows per item) and one launch per item (item0 = its
    number) draw the same masks - elementwise kernels and the attention core (forward and backward)."""
    from phnet_amd import hip_ops as K
    dev = "cuda"
    state = torch.tensor([123456789], dtype=torch.int64, device=dev)
    items, rows, L = 5, 37, 128
    r = np.random.default_rng(3)
    x = torch.from_numpy(r.standard_normal((items * rows, L)).astype(np.float32)).to(dev)
    res = torch.from_numpy(r.standard_normal((items * rows, L)).astype(np.float32)).to(dev)
    w = torch.from_numpy(r.uniform(0.5, 1.5, L).astype(np.float32)).to(dev)
    b = torch.from_numpy(r.normal(0, 0.1, L).astype(np.float32)).to(dev)
    whole = (state, 11, 0.3, 0, rows)
    y = K.dropout_add(x, res, whole)
    gl = K.gelu_dropout_fwd(x, whole)
    t, h, _, _ = K.dropout_add_ln_fwd(x, res, w, b, 1e-5, whole)
    assert 0.2 < float((y == res).float().mean()) < 0.4                     # ~30 % dropped
    for i in range(items):
        one = (state, 11, 0.3, i, 0)
        sl = slice(i * rows, (i + 1) * rows)
        assert torch.equal(K.dropout_add(x[sl].contiguous(), res[sl].contiguous(), one), y[sl])
        assert torch.equal(K.gelu_dropout_fwd(x[sl].contiguous(), one), gl[sl])
        ti, hi, _, _ = K.dropout_add_ln_fwd(x[sl].contiguous(), res[sl].contiguous(), w, b, 1e-5, one)
        assert torch.equal(ti, t[sl]) and torch.equal(hi, h[sl])
    assert not torch.equal(K.dropout_add(x[:rows].contiguous(), res[:rows].contiguous(), (state, 11, 0.3, 1, 0)), y[:rows])
    # attention: batch entry b of a batched launch == a one-entry launch with item0 = b
    H, D, lq, lk = 8, 16, 48, 20
    q = torch.from_numpy(r.standard_normal((items * lq, H * D)).astype(np.float32)).to(dev)
    kv = torch.from_numpy(r.standard_normal((items * lk, 2 * H * D)).astype(np.float32)).to(dev)
    o, lse = K.attention_fwd(q, kv[:, :H * D], kv[:, H * D:], H, None, rng=(state, 12, 0.3, 0, 0), batch=items)
    do = torch.from_numpy(r.standard_normal((items * lq, H * D)).astype(np.float32)).to(dev)
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    K.attention_bwd(q, kv[:, :H * D], kv[:, H * D:], o, do, lse, H, dq, dkv[:, :H * D], dkv[:, H * D:], None, rng=(state, 12, 0.3, 0, 0), batch=items)
    for i in range(items):
        qs, ks = slice(i * lq, (i + 1) * lq), slice(i * lk, (i + 1) * lk)
        qi, kvi = q[qs].contiguous(), kv[ks].contiguous()
        oi, lsei = K.attention_fwd(qi, kvi[:, :H * D], kvi[:, H * D:], H, None, rng=(state, 12, 0.3, i, 0))
        assert torch.equal(oi, o[qs])
        dqi, dkvi = torch.empty_like(qi), torch.empty_like(kvi)
        K.attention_bwd(qi, kvi[:, :H * D], kvi[:, H * D:], oi, do[qs].contiguous(), lsei, H, dqi, dkvi[:, :H * D], dkvi[:, H * D:], None,
                        rng=(state, 12, 0.3, i, 0))
        assert torch.equal(dqi, dq[qs]) and torch.equal(dkvi, dkv[ks])


def _close_k(a, b, tol, what=""):
    close(a, b, tol)


@pytest.mark.parametrize("e,ff", [(128, 256), (256, 512)])
def test_rowchain_vs_fp64_statement(e, ff):
    """csrc/rowchain.hip: the three forms a decoder layer uses (LayerNorm + projection; out-projection + residual + LayerNorm +
    projection; out-projection + residual + LayerNorm + feed-forward + residual + LayerNorm (+ projection)) against torch fp64,
    on a row count that is not a multiple of the 16-row blocks."""
    from phnet_amd import hip_ops as K
    r_ = np.random.default_rng(e)
    R = 37
    def T(*shape, s=1.0):
        return torch.from_numpy((r_.standard_normal(shape) * s).astype(np.float32)).cuda()
    x, res = T(R, e), T(R, e)
    wa, ba = T(e, e, s=e ** -0.5), T(e, s=0.1)
    l1, l2 = (T(e).abs() + 0.5, T(e, s=0.1)), (T(e).abs() + 0.5, T(e, s=0.1))
    w1, b1, w2, b2 = T(ff, e, s=e ** -0.5), T(ff, s=0.1), T(e, ff, s=ff ** -0.5), T(e, s=0.1)
    wg, bg = T(3 * e, e, s=e ** -0.5), T(3 * e, s=0.1)
    d = lambda t: t.double()                                                             # noqa: E731
    # form 1
    _, _, y = K.rowchain_fwd(x, ln1=l1, wg=wg, bg=bg, want_t=False)
    ref = F.linear(F.layer_norm(d(x), [e], d(l1[0]), d(l1[1])), d(wg), d(bg))
    _close_k(y, ref, 2e-5, "form 1")
    # form 2
    t, _, y = K.rowchain_fwd(x, resid=res, wa=wa, ba=ba, ln1=l1, wg=wg[:e], bg=bg[:e])
    t_ref = d(res) + F.linear(d(x), d(wa), d(ba))
    _close_k(t, t_ref, 2e-5, "form 2 t")
    _close_k(y, F.linear(F.layer_norm(t_ref, [e], d(l1[0]), d(l1[1])), d(wg[:e]), d(bg[:e])), 2e-5, "form 2 y")
    # form 3 (with and without the trailing projection)
    for tail in (True, False):
        t, h, y = K.rowchain_fwd(x, resid=res, wa=wa, ba=ba, ln1=l1, ffn=(w1, b1, w2, b2), ln2=l2, wg=wg if tail else None,
                                 bg=bg if tail else None, want_h=True)
        h1 = F.layer_norm(t_ref, [e], d(l1[0]), d(l1[1]))
        t2 = t_ref + F.linear(F.gelu(F.linear(h1, d(w1), d(b1))), d(w2), d(b2))
        h2 = F.layer_norm(t2, [e], d(l2[0]), d(l2[1]))
        _close_k(t, t2, 2e-5, "form 3 t"); _close_k(h, h2, 2e-5, "form 3 h")
        if tail:
            _close_k(y, F.linear(h2, d(wg), d(bg)), 2e-5, "form 3 y")
        else:
            assert y is None


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_decoder_row_chain_forward_equals_unfused_forward(p):
    """TransformerDecoder without autograd (fused row chains) against the same decoder under autograd (one launch per operation):
    same values, and with dropout the same masks (both draw them from DropoutStream sites in the same order)."""
    from phnet_amd import functional as PF
    from phnet_amd.libs.models.utils.transformer import TransformerDecoder, TransformerDecoderLayer
    torch.manual_seed(0)
    layer = TransformerDecoderLayer(d_model=128, nhead=8, dim_feedforward=256, dropout=p, activation="gelu", normalize_before=True)
    dec = TransformerDecoder(layer, 2, torch.nn.LayerNorm(128)).cuda().train()
    r_ = np.random.default_rng(1)
    for batch, lk in ((1, 40), (3, 25)):
        tgt = torch.from_numpy(r_.standard_normal((batch * 240, 128)).astype(np.float32)).cuda()
        mem = torch.from_numpy(r_.standard_normal((batch * lk, 128)).astype(np.float32)).cuda()
        valid = torch.from_numpy(r_.uniform(size=batch * lk) > 0.3).cuda()
        valid[::lk] = True
        outs = []
        for fused in (False, True):
            with PF.DropoutStream.items(512, 2 if batch == 1 else 0, 0 if batch == 1 else 240):
                if fused:
                    with torch.no_grad():
                        outs.append(dec(tgt, mem, valid, batch=batch))
                else:
                    outs.append(dec(tgt.clone().requires_grad_(), mem, valid, batch=batch).detach())
        _close_k(outs[1], outs[0], 1e-4, f"decoder fused vs unfused, p = {p}, batch {batch}")


def test_assembled_tower_weights_and_scattered_gradients():
    """csrc/towers.hip: the six operands of the 3-GEMM tower chain against torch.cat / torch.block_diag, and the gradients of a
    chain through them against autograd on the cat / block_diag spelling."""
    from phnet_amd import functional as PF
    r_ = np.random.default_rng(4)
    C, outs = 64, (2, 4, 36)
    def P(*shape):
        return torch.nn.Parameter(torch.from_numpy(r_.standard_normal(shape).astype(np.float32) * 0.2).cuda())
    params = [x for o in outs for x in (P(C, C), P(C), P(C, C), P(C), P(o, C), P(o))]
    tow = [params[6 * t:6 * t + 6] for t in range(3)]
    x = torch.from_numpy(r_.standard_normal((50, C)).astype(np.float32)).cuda()
    w1, b1, w2, b2, wh, bh = PF.assemble_towers(C, outs, None, params)
    rw1 = torch.cat([t[0] for t in tow]); rb1 = torch.cat([t[1] for t in tow])
    rw2 = torch.block_diag(*[t[2] for t in tow]); rb2 = torch.cat([t[3] for t in tow])
    rwh = torch.block_diag(*[t[4] for t in tow]); rbh = torch.cat([t[5] for t in tow])
    assert torch.equal(w1, rw1) and torch.equal(b1, rb1) and torch.equal(w2, rw2) and torch.equal(b2, rb2)
    assert wh.shape == (44, 3 * C) and torch.equal(wh[:42], rwh) and not wh[42:].any() and torch.equal(bh[:42], rbh) and not bh[42:].any()
    g = torch.from_numpy(r_.standard_normal((50, 44)).astype(np.float32)).cuda()
    y = PF.linear(PF.linear(PF.linear(x, w1, b1, relu=True), w2, b2, relu=True), wh, bh)
    (y * g).sum().backward()
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    yr = F.linear(F.relu(F.linear(F.relu(F.linear(x.double(), rw1.double(), rb1.double())), rw2.double(), rb2.double())), rwh.double(), rbh.double())
    (yr * g[:, :42].double()).sum().backward()
    for a, p in zip(got, params):
        close(a, p.grad, 2e-5)


@pytest.mark.parametrize("c", [64, 128])
def test_tower_chain_forward_vs_fp64_statement(c):
    """csrc/rowchain.hip tower_chain_kernel: the three towers, their heads and the lane prior update of a branch in one launch
    against the same chain in torch fp64 (Router4OL.py:308-345)."""
    from phnet_amd import hip_ops as K
    r_ = np.random.default_rng(c)
    S, R = 36, 53
    outs = (2, 4, S)
    def P(*shape, s=0.2):
        return torch.from_numpy((r_.standard_normal(shape) * s).astype(np.float32)).cuda()
    params = [x for o in outs for x in (P(c, c, s=c ** -0.5), P(c), P(c, c, s=c ** -0.5), P(c), P(o, c, s=0.3 * c ** -0.5), P(o, s=0.05))]
    x = P(R, c, s=1.0)
    pri = torch.zeros(R, 6 + S)
    pri[:, 2] = torch.from_numpy(r_.uniform(0, 0.5, R)); pri[:, 3] = torch.from_numpy(r_.uniform(0.1, 0.9, R)); pri[:, 4] = torch.from_numpy(r_.uniform(0.15, 0.85, R))
    pri = pri.float().cuda()
    ys = torch.linspace(1, 0, S).cuda()
    preds, lines = K.tower_chain_fwd(x, params, outs, pri, ys, 800, 320)
    d = lambda t: t.double()                                                             # noqa: E731
    heads = []
    for t in range(3):
        w1, b1, w2, b2, wh, bh = params[6 * t:6 * t + 6]
        heads.append(F.linear(F.relu(F.linear(F.relu(F.linear(d(x), d(w1), d(b1))), d(w2), d(b2))), d(wh), d(bh)))
    head = torch.cat(heads, dim=1)
    ref_p, ref_l = _lane_update_reference(d(pri), head, d(ys), 800, 320, S)
    # the tan() of the prior update amplifies the GEMMs' 1e-6 (f32 vs f64) near its poles: judge x columns against the row scale
    for got, ref in ((preds, ref_p), (lines, ref_l)):
        got, ref = got.cpu().double(), ref.cpu()
        assert float((got[:, :6] - ref[:, :6]).abs().max()) <= 2e-5
        scale = 1.0 + ref[:, 6:].abs().amax(dim=1, keepdim=True)
        assert float(((got[:, 6:] - ref[:, 6:]).abs() / scale).max()) <= 2e-4


def test_lane_assign_tokens_equals_the_two_launches():
    from phnet_amd import hip_ops as K
    from tests import synth
    from oracle import phnet_cpu as O
    g = O.Geometry()
    r_ = np.random.default_rng(8)
    base = O.priors_from_embeddings(O.initial_anchor_embeddings(g), g)[0]
    for counts in ((3,), (0,), (4,), (1,)):
        tgt = synth.make_targets(g, 1, counts=counts)[0].cuda()
        pred = (base + torch.from_numpy(r_.normal(0, 0.02, base.shape).astype(np.float32))).cuda()
        pred[:, :2] = torch.from_numpy(r_.normal(0, 1, (240, 2)).astype(np.float32)).cuda()
        feat = torch.from_numpy(r_.standard_normal((240, 128)).astype(np.float32)).cuda()
        _, srt, _ = K.lane_assign(pred, tgt, g.img_w, g.img_h)
        tok_ref, val_ref = K.memory_tokens(feat, srt)
        tok = torch.full((5, 128), 7.0, device="cuda"); val = torch.zeros(5, dtype=torch.bool, device="cuda")
        srt2 = K.lane_assign_tokens(pred, tgt, g.img_w, g.img_h, feat, (tok, val))
        assert torch.equal(srt2, srt) and torch.equal(val, val_ref)
        close(tok, tok_ref.reshape(5, 128), 1e-6)
