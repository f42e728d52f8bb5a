"""GPU parity tests for the fused DoubleMHA kernels: HIP path (module API -> ctypes -> C ABI)
against the CPU oracle and the golden vectors captured from the reference.

Tolerances (fp32 path): BASELINE.json asks for embeddings within 1e-3 relative; the pooling
kernel itself is held to ~1e-5 absolute on outputs and 1e-6 on alignments.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden, rnd

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _pool():
    from doubleattentionspeakerverification_amd import poolings
    return poolings


def _lib():
    from doubleattentionspeakerverification_amd import _lib
    return _lib.load()


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def _x(g, i):
    key = "c%d_x" % i
    if key in g.files:
        return g[key]
    B, T, D, H = g["c%d_shape" % i]
    return rnd(int(g["c%d_seed" % i]), int(B), int(T), int(D))


def _module(D, H, q, a, mask_prob=0.2):
    m = _pool().DoubleMHA(D, H, mask_prob=mask_prob).to(DEV)
    with torch.no_grad():
        m.utteranceAttention.query.copy_(_t(q))
        m.headsAttention.att.copy_(_t(a))
    return m


# ------------------------------------------------------------------ golden vectors (reference outputs)
@pytest.mark.parametrize("i", range(6))
def test_golden_eval_forward(i):
    g = load_golden("g1_pooling_eval")
    B, T, D, H = (int(v) for v in g["c%d_shape" % i])
    m = _module(D, H, g["c%d_query" % i], g["c%d_att" % i]).eval()
    x = _t(_x(g, i))
    with torch.no_grad():
        out, al = m(x)
        al2, hw = m.getAlignments(x)
        ctx, _ = m.utteranceAttention(x)
    assert out.shape == (B, D // H) and al.shape == (B, T, H) and hw.shape == (B, H, 1)
    np.testing.assert_allclose(out.cpu().numpy(), g["c%d_out" % i], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(al.cpu().numpy(), g["c%d_align" % i], rtol=0, atol=1e-6)
    np.testing.assert_allclose(al2.cpu().numpy(), g["c%d_align" % i], rtol=0, atol=1e-6)
    np.testing.assert_allclose(hw.cpu().numpy(), g["c%d_head_w" % i], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ctx.cpu().numpy(), g["c%d_ctx" % i], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["big", "relu"])
def test_golden_stress_inputs(tag):
    g = load_golden("g1_pooling_eval")
    m = _module(64, 4, g["s_query"], g["s_att"]).eval()
    with torch.no_grad():
        out, al = m(_t(g["s_%s_x" % tag]))
    np.testing.assert_allclose(out.cpu().numpy(), g["s_%s_out" % tag], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(al.cpu().numpy(), g["s_%s_align" % tag], rtol=0, atol=2e-6)


def test_golden_single_head_attention():
    g = load_golden("g1_pooling_eval")
    m = _pool().Attention(48).to(DEV).eval()
    with torch.no_grad():
        m.att.copy_(_t(g["a_att"]))
        ct, sc = m(_t(g["a_x"]))
    assert sc.shape == (3, 23, 1)
    np.testing.assert_allclose(ct.cpu().numpy(), g["a_ct"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(sc.cpu().numpy(), g["a_score"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("i", range(3))
def test_golden_train_masked_forward_backward(i):
    g = load_golden("g2_pooling_train")
    B, T, D, H = (int(v) for v in g["c%d_shape" % i])
    m = _module(D, H, g["c%d_query" % i], g["c%d_att" % i], mask_prob=float(g["c%d_mask_prob" % i])).train()
    x = _t(g["c%d_x" % i]).requires_grad_(True)
    out, al = m(x, head_keep=_t(g["c%d_keep" % i], torch.bool))
    (out * _t(g["c%d_grad_out" % i])).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["c%d_out" % i], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(al.cpu().numpy(), g["c%d_align" % i], rtol=0, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["c%d_grad_x" % i], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(m.utteranceAttention.query.grad.cpu().numpy(), g["c%d_grad_query" % i], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.headsAttention.att.grad.cpu().numpy(), g["c%d_grad_att" % i], rtol=1e-4, atol=1e-5)


def test_golden_eval_backward():
    g = load_golden("g2_pooling_train")
    m = _module(64, 4, g["e_query"], g["e_att"]).eval()
    x = _t(g["e_x"]).requires_grad_(True)
    out, _ = m(x)
    (out * _t(g["e_grad_out"])).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["e_out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["e_grad_x"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(m.utteranceAttention.query.grad.cpu().numpy(), g["e_grad_query"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.headsAttention.att.grad.cpu().numpy(), g["e_grad_att"], rtol=1e-4, atol=1e-5)


def test_all_heads_masked_row_is_nan_like_reference():
    pool = _pool()
    x = _t(rnd(1, 3, 9, 32))
    keep = torch.ones(3, 4, dtype=torch.bool)
    keep[1] = False
    out, _, hw, _ = pool.double_mha(x, _t(rnd(2, 8, 4)), _t(rnd(3, 8, 1)), head_keep=keep.to(DEV))
    out = out.cpu()
    assert torch.isnan(out[1]).all() and not torch.isnan(out[0]).any() and not torch.isnan(out[2]).any()


# ------------------------------------------------------------------ random shapes vs the closed-form oracle
SHAPES = [
    # B, T, D, H        (dh = D/H decides the kernel variant: vectors/row -> (G, NI))
    (2, 5, 64, 1),       # dh 64  -> G16 NI1, single head
    (3, 33, 256, 4),     # dh 64
    (2, 70, 512, 4),     # dh 128 -> G16 NI2
    (2, 19, 5120, 32),   # dh 160 -> G8 NI5  (default model)
    (1, 38, 10240, 32),  # dh 320 -> G16 NI5 (VGG3L)
    (2, 40, 2560, 32),   # dh 80  -> G4 NI5
    (2, 21, 96, 2),      # dh 48  -> G4 NI3
    (3, 17, 36, 3),      # dh 12  -> 3 vectors: generic kernel
    (2, 9, 30, 5),       # dh 6   -> unaligned: generic kernel
    (1, 1, 8, 2),        # one frame
    (5, 700, 128, 2),    # long T: several passes
    (9, 64, 128, 8),     # dh 16 -> G4 NI1, B not a multiple of 8
]


@pytest.mark.parametrize("B,T,D,H", SHAPES)
def test_forward_backward_vs_oracle(B, T, D, H):
    pool = _pool()
    dh = D // H
    x, q, a = rnd(11, B, T, D), rnd(12, dh, H, scale=0.3), rnd(13, dh, 1, scale=0.3)
    go = rnd(14, B, dh)
    keep = np.ones((B, H), bool)
    if H > 1:
        keep[:, 1] = False
    xt = _t(x).requires_grad_(True)
    qt = _t(q).requires_grad_(True)
    at = _t(a).requires_grad_(True)
    out, al, hw, ctx = pool.double_mha(xt, qt, at, head_keep=_t(keep, torch.bool))
    (out * _t(go)).sum().backward()
    f = oracle.dmha_closed(x, q, a, head_keep=keep)
    b = oracle.dmha_backward_closed(x, q, a, go, head_keep=keep)
    np.testing.assert_allclose(out.detach().cpu().numpy(), f["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(al.cpu().numpy(), f["align"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(hw.cpu().numpy()[..., 0], f["head_w"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ctx.cpu().numpy(), f["ctx"], rtol=1e-5, atol=1e-5)
    scale = max(1.0, np.abs(b["grad_query"]).max())
    np.testing.assert_allclose(xt.grad.cpu().numpy(), b["grad_x"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(qt.grad.cpu().numpy(), b["grad_query"], rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(at.grad.cpu().numpy(), b["grad_att"], rtol=1e-4, atol=2e-5 * scale)


def test_mha_only_forward_backward_vs_oracle():
    pool = _pool()
    B, T, D, H = 3, 29, 256, 4
    x, q = rnd(21, B, T, D), rnd(22, D // H, H, scale=0.3)
    gc = rnd(23, B, D)
    xt, qt = _t(x).requires_grad_(True), _t(q).requires_grad_(True)
    ctx, al = pool.multi_head_attention(xt, qt)
    (ctx * _t(gc)).sum().backward()
    xo = torch.from_numpy(x).requires_grad_(True)
    qo = torch.from_numpy(q).requires_grad_(True)
    co, po = oracle.mha_opseq(xo, qo)
    (co * torch.from_numpy(gc)).sum().backward()
    np.testing.assert_allclose(ctx.detach().cpu().numpy(), co.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(al.cpu().numpy(), po.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), xo.grad.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(qt.grad.cpu().numpy(), qo.grad.numpy(), rtol=1e-4, atol=2e-5)


def _dbg_double_mha(lib, x, q, a):
    """One forward through the C ABI of the DEVELOPER build (libdasv_hip_dbg.so: same sources + the tuning knobs the product
    library does not export) — returns (out, align, variant)."""
    B, T, D = x.shape
    dh, H = q.shape
    f32 = dict(dtype=torch.float32, device=x.device)
    out, align = torch.empty(B, dh, **f32), torch.empty(B, T, H, **f32)
    hw, ctx, ml = torch.empty(B, H, **f32), torch.empty(B, H, dh, **f32), torch.empty(B, H, 2, **f32)
    ws = torch.zeros(lib.dasv_dmha_workspace_bytes(B, T, H, dh), dtype=torch.uint8, device=x.device)
    rc = lib.dasv_dmha_fwd(0, x.data_ptr(), q.data_ptr(), a.data_ptr(), None, None, out.data_ptr(), align.data_ptr(), hw.data_ptr(),
                           ctx.data_ptr(), ml.data_ptr(), ws.data_ptr(), ws.numel(), B, T, H, dh, T * D, D,
                           torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out, align, lib.dasv_dmha_last_variant().decode()


def test_generic_kernel_matches_vector_kernel():
    from doubleattentionspeakerverification_amd import _lib as libmod
    dbg = libmod.debug_probe()
    x, q, a = _t(rnd(31, 4, 45, 256)), _t(rnd(32, 64, 4, scale=0.3)), _t(rnd(33, 64, 1, scale=0.3)).view(-1)
    out1, al1, v1 = _dbg_double_mha(dbg, x, q, a)
    assert "vec<" in v1
    dbg.dasv_debug_set_tuning(0, 0, 0, 1, 0)
    try:
        out2, al2, v2 = _dbg_double_mha(dbg, x, q, a)
        assert "generic" in v2
    finally:
        dbg.dasv_debug_set_tuning(0, 0, 0, 0, 0)
    np.testing.assert_allclose(out1.cpu().numpy(), out2.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(al1.cpu().numpy(), al2.cpu().numpy(), atol=1e-6)
    # the product library exports no knob at all
    assert not hasattr(_lib(), "dasv_debug_set_tuning") and not hasattr(_lib(), "dasv_debug_read_floor")


@pytest.mark.parametrize("nw,wpu,rows", [(4, 1, 0), (4, 2, 8), (4, 4, 2), (8, 8, 0), (8, 2, 0), (2, 1, 8), (1, 1, 0), (4, 4, 8)])
def test_workgroup_shapes_agree(nw, wpu, rows):
    """Every (waves per workgroup, waves per head) split must give the same answer."""
    from doubleattentionspeakerverification_amd import _lib as libmod
    dbg = libmod.debug_probe()
    x, q, a = rnd(41, 3, 150, 512), rnd(42, 64, 8, scale=0.3), rnd(43, 64, 1, scale=0.3)
    f = oracle.dmha_closed(x, q, a)
    dbg.dasv_debug_set_tuning(nw, wpu, 0, 0, rows)
    try:
        out, al, _ = _dbg_double_mha(dbg, _t(x), _t(q), _t(a).view(-1))
    finally:
        dbg.dasv_debug_set_tuning(0, 0, 0, 0, 0)
    np.testing.assert_allclose(out.cpu().numpy(), f["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(al.cpu().numpy(), f["align"], atol=1e-6)


# ------------------------------------------------------------------ ragged batches, strides, dtypes
def test_lengths_equal_running_each_utterance_alone():
    pool = _pool()
    B, T, D, H = 5, 60, 256, 4
    x = rnd(51, B, T, D)
    lens = np.array([60, 1, 37, 59, 13], dtype=np.int32)
    for bi, n in enumerate(lens):
        x[bi, n:] = 777.0  # garbage in the padding must not matter
    q, a = rnd(52, D // H, H, scale=0.3), rnd(53, D // H, 1, scale=0.3)
    xt = _t(x).requires_grad_(True)
    out, al, hw, _ = pool.double_mha(xt, _t(q), _t(a), lengths=torch.from_numpy(lens))
    go = rnd(54, B, D // H)
    (out * _t(go)).sum().backward()
    for bi, n in enumerate(lens):
        f = oracle.dmha_closed(x[bi:bi + 1, :n], q, a)
        bw = oracle.dmha_backward_closed(x[bi:bi + 1, :n], q, a, go[bi:bi + 1])
        np.testing.assert_allclose(out[bi].detach().cpu().numpy(), f["out"][0], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(al[bi, :n].cpu().numpy(), f["align"][0], atol=1e-6)
        assert float(al[bi, n:].abs().max()) == 0.0 if n < T else True
        np.testing.assert_allclose(xt.grad[bi, :n].cpu().numpy(), bw["grad_x"][0], rtol=1e-4, atol=5e-6)
        if n < T:
            assert float(xt.grad[bi, n:].abs().max()) == 0.0


def test_row_strided_input_is_read_in_place():
    pool = _pool()
    big = _t(rnd(61, 3, 20, 160))
    x = big[:, :, :128]            # row stride 160, features contiguous
    assert not x.is_contiguous()
    q, a = rnd(62, 32, 4, scale=0.3), rnd(63, 32, 1, scale=0.3)
    out, al, _, _ = pool.double_mha(x, _t(q), _t(a))
    f = oracle.dmha_closed(x.cpu().numpy(), q, a)
    np.testing.assert_allclose(out.cpu().numpy(), f["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(al.cpu().numpy(), f["align"], atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,D,H", [(3, 50, 512, 8), (2, 19, 5120, 32), (2, 23, 40, 4)])
def test_half_precision_inputs_fp32_math(dtype, B, T, D, H):
    """16-bit frame embeddings are widened in registers; all arithmetic stays fp32, so the result must
    match the oracle evaluated on the rounded inputs to fp32 accuracy."""
    pool = _pool()
    dh = D // H
    x16 = _t(rnd(71, B, T, D)).to(dtype)
    q, a, go = rnd(72, dh, H, scale=0.3), rnd(73, dh, 1, scale=0.3), rnd(74, B, dh)
    xt = x16.clone().requires_grad_(True)
    out, al, hw, ctx = pool.double_mha(xt, _t(q), _t(a))
    (out * _t(go)).sum().backward()
    xr = x16.float().cpu().numpy()
    f = oracle.dmha_closed(xr, q, a)
    b = oracle.dmha_backward_closed(xr, q, a, go)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), f["out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(al.cpu().numpy(), f["align"], atol=1e-6)
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3   # grad_x is rounded to the 16-bit type on store
    assert xt.grad.dtype == dtype
    np.testing.assert_allclose(xt.grad.float().cpu().numpy(), b["grad_x"], rtol=tol, atol=tol * np.abs(b["grad_x"]).max())


# ------------------------------------------------------------------ full benchmark shape + in-launch exchange
def test_config2_full_size_against_oracle_and_properties():
    """BASELINE config 2: B=64, T=300, D=512, H=8 fp32."""
    pool = _pool()
    rs = np.random.RandomState(1234)
    x = rs.standard_normal((64, 300, 512)).astype(np.float32)
    q = (rs.standard_normal((64, 8)) * np.sqrt(2.0 / (64 + 8))).astype(np.float32)
    a = (rs.standard_normal((64, 1)) * np.sqrt(2.0 / (64 + 1))).astype(np.float32)
    out, al, hw, ctx = pool.double_mha(_t(x), _t(q), _t(a))
    f = oracle.dmha_closed(x, q, a)
    o = out.cpu().numpy()
    rel = np.abs(o - f["out"]).max() / np.abs(f["out"]).max()
    assert rel < 1e-5, rel
    np.testing.assert_allclose(al.cpu().numpy(), f["align"], atol=1e-7 + 1e-6)
    # size-independent properties
    np.testing.assert_allclose(al.sum(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    np.testing.assert_allclose(hw.sum(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    xv = _t(x).view(64, 300, 8, 64)
    assert bool((ctx <= xv.max(dim=1).values + 1e-5).all()) and bool((ctx >= xv.min(dim=1).values - 1e-5).all())
    # shift invariance of both softmaxes: adding a constant vector along q's null space changes nothing... use
    # the cheaper check: permuting frames permutes alignments and leaves the output unchanged
    perm = torch.randperm(300, device=DEV)
    out_p, al_p, _, _ = pool.double_mha(_t(x)[:, perm].contiguous(), _t(q), _t(a))
    np.testing.assert_allclose(out_p.cpu().numpy(), o, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(al_p.cpu().numpy(), al[:, perm].cpu().numpy(), atol=1e-6)


def test_head_exchange_is_never_stale_across_repeated_launches():
    """The last-arriving workgroup of each utterance re-reads the other heads' context vectors inside the
    launch.  Re-launch many times on the same buffers with changing inputs: `out` must always equal the head
    stage recomputed from the `ctx` of the SAME launch (a stale read would reuse an older launch's vector)."""
    pool = _pool()
    B, T, D, H = 64, 300, 512, 8
    q, a = _t(rnd(82, 64, 8, scale=0.3)), _t(rnd(83, 64, 1, scale=0.3))
    xs = [_t(rnd(90 + i, B, T, D, scale=1.0 + 0.5 * i)) for i in range(4)]
    worst = 0.0
    for it in range(200):
        out, _, hw, ctx = pool.double_mha(xs[it % 4], q, a, want_align=(it % 2 == 0))
        u = (ctx @ a.view(-1, 1)).squeeze(-1)
        w = torch.softmax(u, dim=-1)
        ref = (ctx * w.unsqueeze(-1)).sum(dim=1)
        worst = max(worst, float((out - ref).abs().max()))
    assert worst < 1e-5, worst


def test_cabi_argument_errors_are_codes_not_crashes():
    lib = _lib()
    x = torch.zeros(2, 4, 8, device=DEV)
    q = torch.zeros(4, 2, device=DEV)
    ctx = torch.zeros(2, 2, 4, device=DEV)
    args = lambda xp, cp, B=2: (0, xp, q.data_ptr(), None, None, None, None, None, None, cp, None, None, 0, B, 4, 2, 4, 32, 8, None)  # noqa: E731
    assert lib.dasv_dmha_fwd(*args(None, ctx.data_ptr())) == -1
    assert lib.dasv_dmha_fwd(*args(x.data_ptr(), None)) == -1
    assert lib.dasv_dmha_fwd(*args(x.data_ptr(), ctx.data_ptr(), B=0)) == -2
    bad = list(args(x.data_ptr(), ctx.data_ptr()))
    bad[0] = 9
    assert lib.dasv_dmha_fwd(*bad) == -5
    # att given but no workspace
    a = torch.zeros(4, device=DEV)
    out = torch.zeros(2, 4, device=DEV)
    bad = list(args(x.data_ptr(), ctx.data_ptr()))
    bad[3], bad[6] = a.data_ptr(), out.data_ptr()
    assert lib.dasv_dmha_fwd(*bad) == -3
    torch.cuda.synchronize()


@pytest.mark.parametrize("T,H,dh", [(20011, 2, 6), (45003, 2, 6), (45003, 2, 64)])
def test_very_long_utterances_pool_like_the_reference(T, H, dh):
    """the reference pools any number of frames.  20 k frames need more than the default 64 KB of dynamic LDS on
    the shape-agnostic kernel; 45 k frames do not fit the 160 KB at all and park their scores in the workspace."""
    import oracle
    from doubleattentionspeakerverification_amd import poolings
    rs = np.random.RandomState(T)
    x = rs.standard_normal((2, T, H * dh)).astype(np.float32)
    q = (rs.standard_normal((dh, H)) * 0.3).astype(np.float32)
    a = (rs.standard_normal((dh, 1)) * 0.3).astype(np.float32)
    ref = oracle.dmha_closed(x, q, a)
    out, align, hw, _ = poolings.double_mha(torch.from_numpy(x).to(DEV), torch.from_numpy(q).to(DEV), torch.from_numpy(a).to(DEV))
    np.testing.assert_allclose(out.cpu().numpy(), ref["out"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(align.cpu().numpy(), ref["align"], rtol=2e-3, atol=1e-8)
    np.testing.assert_allclose(hw.cpu().numpy().reshape(2, H), ref["head_w"].reshape(2, H), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,T,C,H", [(3, 19, 512, 32), (5, 19, 1024, 32), (2, 7, 1024, 16), (9, 50, 256, 16)])
def test_nhwc_map_is_pooled_in_place_like_the_transposed_frames(B, T, C, H):
    """CNNs.py:88-89 folded into the kernels' addressing: pooling the front-end's [B,C,T',F'] channels_last map in place
    (forward and backward) must equal pooling the materialised [B,T',C*F'] frames, and the closed-form oracle."""
    import oracle
    pool = _pool()
    F = 5
    dh = C * F // H
    rs = np.random.RandomState(C + T)
    frames = rs.standard_normal((B, T, C * F)).astype(np.float32)            # reference order: feature = c*F + f
    q = (rs.standard_normal((dh, H)) * 0.2).astype(np.float32)
    a = (rs.standard_normal((dh, 1)) * 0.2).astype(np.float32)
    go = rs.standard_normal((B, dh)).astype(np.float32)
    keep = rs.randint(0, 4, size=(B, H)) > 0
    keep[:, 0] = True
    lens = rs.randint(1, T + 1, size=B).astype(np.int32)
    lens[0] = T
    # the same numbers as a [B,C,T,F] map in channels_last memory
    h4 = _t(frames).view(B, T, C, F).permute(0, 2, 1, 3).contiguous(memory_format=torch.channels_last)
    assert h4.stride() == (T * F * C, 1, F * C, C)
    assert pool.nhwc_frames_supported(h4, H)
    mod = _module(C * F, H, q, a)
    mod.train()
    res = {}
    for tag in ("map", "frames"):
        qt = mod.utteranceAttention.query
        at = mod.headsAttention.att
        for p_ in (qt, at):
            p_.grad = None
        if tag == "map":
            x = h4.clone().requires_grad_(True)
            out, al = mod.pool_map(x, head_keep=_t(keep, torch.bool), lengths=_t(lens, torch.int32), want_align=True)
            assert "nhwc" in _lib().dasv_dmha_last_variant().decode()
        else:
            x = _t(frames).requires_grad_(True)
            out, al = mod(x, head_keep=_t(keep, torch.bool), lengths=_t(lens, torch.int32))
        (out * _t(go)).sum().backward()
        gx = x.grad
        if tag == "map":
            assert gx.stride() == h4.stride()
            gx = gx.permute(0, 2, 1, 3).reshape(B, T, C * F)
        res[tag] = [v.detach().cpu().numpy() for v in (out, al, gx, qt.grad, at.grad)]
    for k, (m, f) in enumerate(zip(res["map"], res["frames"])):
        np.testing.assert_allclose(m, f, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(f).max())), err_msg=str(k))
    # and against the oracle, utterance by utterance at its own length
    for b in range(B):
        n = int(lens[b])
        ref = oracle.dmha_closed(frames[b:b + 1, :n], q, a, head_keep=keep[b:b + 1])
        np.testing.assert_allclose(res["map"][0][b], ref["out"][0], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(res["map"][1][b, :n], ref["align"][0], rtol=0, atol=1e-6)
        assert float(np.abs(res["map"][1][b, n:]).max(initial=0.0)) == 0.0


def test_one_cached_workspace_serves_any_sequence_of_batch_sizes():
    """the arrival counters used to sit in the first round_up(B,64)*4 bytes of the workspace and the head
    scores / long-T raw scores behind them — at a B-dependent offset — while poolings caches ONE buffer per (device, stream).
    A small-B launch (and a long-T one, whose raw scores make the buffer large enough to be reused for ever) wrote where a later
    B > 64 launch expects zeroed counters: no last arriver was elected and rows of `out` came back uninitialised.  The counter
    region now has a fixed size.  B = 2 with long T, then B = 8, then B = 130, then 200 and 3 again, same stream, each against
    the oracle."""
    import oracle
    pool = _pool()
    pool.reset_workspaces()
    H, dh = 4, 16
    rs = np.random.RandomState(9)
    q = (rs.standard_normal((dh, H)) * 0.3).astype(np.float32)
    a = (rs.standard_normal((dh, 1)) * 0.3).astype(np.float32)
    for B, T in ((2, 45003), (8, 40), (130, 33), (200, 7), (3, 19), (70, 300)):
        x = rs.standard_normal((B, T, H * dh)).astype(np.float32)
        out, align, hw, _ = pool.double_mha(torch.from_numpy(x).to(DEV), torch.from_numpy(q).to(DEV), torch.from_numpy(a).to(DEV))
        ref = oracle.dmha_closed(x, q, a)
        np.testing.assert_allclose(out.cpu().numpy(), ref["out"], rtol=2e-4, atol=2e-6, err_msg=str((B, T)))
        np.testing.assert_allclose(hw.cpu().numpy().reshape(B, H), ref["head_w"].reshape(B, H), rtol=1e-4, atol=1e-6)
    # and the C ABI's own statement of it: the counter region does not depend on B
    lib = _lib()
    per_utt = H * 4 + H * (4 + dh) * 4          # head scores + one partial {M, S, 0, 0 | V[dh]} per head at most
    assert lib.dasv_dmha_workspace_bytes(1, 10, H, dh) - per_utt == lib.dasv_dmha_workspace_bytes(130, 10, H, dh) - 130 * per_utt == 16384 * 4


def test_nhwc_map_of_a_very_long_utterance_falls_back_to_the_frames_path():
    """the in-place NHWC kernel keeps a score row per head in LDS; a map too long for it (here 45,000 pooled
    frames = two hours of audio) must not raise — pool_map permutes and takes the shape-agnostic path, like the reference,
    which pools any length (poolings.py:73-80)."""
    import oracle
    pool = _pool()
    B, T, C, F, H = 1, 45000, 256, 5, 16
    dh = C * F // H
    rs = np.random.RandomState(3)
    frames = rs.standard_normal((B, T, C * F)).astype(np.float32)
    q = (rs.standard_normal((dh, H)) * 0.2).astype(np.float32)
    a = (rs.standard_normal((dh, 1)) * 0.2).astype(np.float32)
    h4 = _t(frames).view(B, T, C, F).permute(0, 2, 1, 3).contiguous(memory_format=torch.channels_last)
    assert pool.nhwc_frames_supported(h4, H)
    mod = _module(C * F, H, q, a).eval()
    with torch.no_grad():
        out = mod.pool_map(h4)
    ref = oracle.dmha_closed(frames, q, a)
    np.testing.assert_allclose(out.cpu().numpy(), ref["out"], rtol=2e-4, atol=2e-6)
