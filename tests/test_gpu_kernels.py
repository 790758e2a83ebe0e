"""GPU: kernel-level checks of the hand-written HIP pieces against plain torch fp32 references (through the C ABI)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loss_limbs_value(lossbuf):
    """[..., limbs] int64 -> float64 value of each sum (bf_debed_last / bf_pm2nchw: base-2^48 digits in units of 2^-112)."""
    out = torch.zeros(lossbuf.shape[:-1], dtype=torch.float64, device=lossbuf.device)
    for k in range(lossbuf.shape[-1]):
        out += lossbuf[..., k].double() * 2.0 ** (48 * k - 112)
    return out


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def K():
    from bubbleformer_amd import kernels
    return kernels


TOL = {torch.float32: 2e-6, torch.bfloat16: 1.2e-2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K_", [(200, 136, 96), (128, 128, 64), (37 * 8, 24, 40), (1000, 392, 264)])
def test_gemm_nt_plain_bias(K, dtype, M, N, K_):
    from bubbleformer_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(M, K_, device="cuda", generator=g).to(dtype)
    w = torch.randn(N, K_, device="cuda", generator=g).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    c = torch.empty(M, N, device="cuda", dtype=dtype)
    K.gemm(dtype, M, N, K_, K.operand(a, K_), K.operand(w, K_), K.epilogue(c, N, bias=bias))
    ref = a.float() @ w.float().t() + bias
    assert _rel(c.float(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_dA_layout_xc_and_aux(K, dtype):
    """dA = dC @ W with W [N][K] read as the XC operand; epilogue adds a residual / multiplies gelu'."""
    from bubbleformer_amd import _lib as L
    M, N, K_ = 264, 200, 136          # dC [M][N], W [N][K_] -> out [M][K_]
    g = torch.Generator(device="cuda").manual_seed(2)
    dc = torch.randn(M, N, device="cuda", generator=g).to(dtype)
    w = torch.randn(N, K_, device="cuda", generator=g).to(dtype)
    aux = torch.randn(M, K_, device="cuda", generator=g).to(dtype)
    out = torch.empty(M, K_, device="cuda", dtype=dtype)
    K.gemm(dtype, M, K_, N, K.operand(dc, N), K.operand(w, K_, layout=L.BF_LAY_XC), K.epilogue(out, K_, aux_mode=L.BF_AUX_ADD, aux=aux, ld_aux=K_))
    ref = dc.float() @ w.float() + aux.float()
    assert _rel(out.float(), ref) < TOL[dtype]
    K.gemm(dtype, M, K_, N, K.operand(dc, N), K.operand(w, K_, layout=L.BF_LAY_XC), K.epilogue(out, K_, aux_mode=L.BF_AUX_DGELU, aux=aux, ld_aux=K_))
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref = (dc.float() @ w.float()) * x.grad
    assert _rel(out.float(), ref) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("splitk", [1, 5])
def test_gemm_dW_layout_tn_splitk_prologue(K, dtype, splitk):
    """dW[n][k] = sum_m dC[m][n] * (x[m][k]*sc[f,k]+sh[f,k]); both operands outer-contiguous (transposing LDS reads)."""
    from bubbleformer_amd import _lib as L
    Mtok, N, K_, S = 6 * 52, 136, 72, 52
    g = torch.Generator(device="cuda").manual_seed(3)
    dc = torch.randn(Mtok, N, device="cuda", generator=g).to(dtype)
    x = torch.randn(Mtok, K_, device="cuda", generator=g).to(dtype)
    sc = torch.randn(6, K_, device="cuda", generator=g)
    sh = torch.randn(6, K_, device="cuda", generator=g)
    out = torch.zeros(N, K_, device="cuda", dtype=torch.float32)
    K.gemm(dtype, N, K_, Mtok, K.operand(dc, N, layout=L.BF_LAY_XC),
           K.operand(x, K_, layout=L.BF_LAY_XC, pro=L.BF_PRO_AFFINE, sc=sc, sh=sh, rows_per_frame=S, nch=K_),
           K.epilogue(out, K_, out_mode=L.BF_OUT_ATOMIC_F32), splitk=splitk)
    xn = (x.float().view(6, S, K_) * sc[:, None] + sh[:, None]).view(Mtok, K_)
    if dtype == torch.bfloat16:
        xn = xn.bfloat16().float()
    ref = dc.float().t() @ xn
    assert _rel(out, ref) < TOL[dtype]


@pytest.mark.parametrize("M,N,K_", [(256 * 3, 128 * 5, 384), (256 * 20, 1152, 384), (256 * 5, 384, 256), (256 * 9, 256, 320), (18432, 1536, 384),
                                    (256, 128, 128), (256 * 7, 384, 1536), (256 * 31, 768, 64 * 11), (256 * 18, 1536, 384)])
@pytest.mark.parametrize("variant", ["plain", "gelu2", "add_cs", "add", "dgelu"])
def test_gemm_stream_weight_stationary(K, M, N, K_, variant):
    """The persistent LDS-DMA streaming kernel (gemm_stream.hip, ping-pong form) takes these bf16 shapes: every epilogue variant against
    fp32 torch on the same bf16 operands; 2 .. 24 K-steps (ring wrap inside and across tiles), one and many tiles per workgroup, a
    single-tile launch; 18 row tiles x 12 column blocks is the case the column blocks are dealt to two groups of teams (team_split)."""
    from bubbleformer_amd import _lib as L
    if variant != "plain" and M == 18432:
        pytest.skip("full-size shape once")
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(M, K_, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, K_, device="cuda", generator=g) * 0.1).to(dt)
    bias = torch.randn(N, device="cuda", generator=g)
    aux = torch.randn(M, N, device="cuda", generator=g).to(dt)
    c = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
    ref = a.float() @ w.float().t() + bias
    import ctypes, json
    h = L.lib()
    h.bf_prof_enable(1)
    if variant == "plain":
        K.gemm(dt, M, N, K_, K.operand(a, K_), K.operand(w, K_), K.epilogue(c, N, bias=bias))
    elif variant == "gelu2":
        c2 = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
        K.gemm(dt, M, N, K_, K.operand(a, K_), K.operand(w, K_), K.epilogue(c, N, bias=bias, gelu_out=c2))
        assert _rel(c2.float(), torch.nn.functional.gelu(c.float())) < 6e-3
    elif variant == "add_cs":
        cs = torch.randn(N, device="cuda", generator=g); ch = torch.randn(N, device="cuda", generator=g)
        rpg = 48
        rs = torch.randn((M + rpg - 1) // rpg, device="cuda", generator=g)
        K.gemm(dt, M, N, K_, K.operand(a, K_), K.operand(w, K_),
               K.epilogue(c, N, bias=bias, colscale=cs, colshift=ch, aux_mode=L.BF_AUX_ADD, aux=aux, ld_aux=N, rowscale=rs, rows_per_group=rpg))
        ref = (ref * cs + ch) * rs.repeat_interleave(rpg)[:M, None] + aux.float()
    elif variant == "add":
        K.gemm(dt, M, N, K_, K.operand(a, K_), K.operand(w, K_), K.epilogue(c, N, bias=bias, aux_mode=L.BF_AUX_ADD, aux=aux, ld_aux=N))
        ref = ref + aux.float()
    else:
        K.gemm(dt, M, N, K_, K.operand(a, K_), K.operand(w, K_), K.epilogue(c, N, bias=bias, aux_mode=L.BF_AUX_DGELU, aux=aux, ld_aux=N))
        x = aux.float().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        ref = ref * x.grad
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    h.bf_prof_report(buf, len(buf))
    h.bf_prof_enable(0)
    assert any(k.startswith("stream_") for k in json.loads(buf.value.decode())), "the streaming kernel did not take this shape"
    assert torch.isfinite(c.float()).all()
    assert _rel(c.float(), ref) < 4e-3            # bf16 output rounding (2^-9 relative per element)
    # each output element against its own scale: a misplaced 8-column group or row shows up as O(1) errors somewhere
    # (the gelu' variant multiplies by the bf16 kernels' polynomial gelu', |error| <= 8e-5 inside |x| <= 4 and <= 5e-4 in the clamped tails)
    assert float(((c.float() - ref).abs() / (ref.abs() + 0.05 * ref.abs().mean())).max()) < (0.1 if variant == "dgelu" else 0.05)


@pytest.mark.parametrize("frames,N", [(16, 1152), (3, 1152), (12, 1024), (16, 384), (128, 1536)])
@pytest.mark.parametrize("variant", ["norm_bias", "norm_fold_resid", "plain_fold_resid", "gelu"])
def test_frame_linear_equals_the_separate_launches(K, frames, N, variant):
    """Whole-frame inference projection (frame_fwd.hip, K = 384 resident form; column blocks of 96 / 64 / 32): the InstanceNorm in front,
    bias, out-projection fold + residual and GELU against fp32 torch on the same bf16 operands, and BIT FOR BIT against the launches it
    replaces (bf_in_stats + bf_affine_apply, then bf_gemm with the same epilogue)."""
    from bubbleformer_amd import _lib as L
    if frames == 128 and variant != "gelu":
        pytest.skip("large batch once")
    dt = torch.bfloat16
    S, E = 144, 384
    M = frames * S
    g = torch.Generator(device="cuda").manual_seed(21)
    a = (torch.randn(M, E, device="cuda", generator=g) * 1.5 + 0.7 * torch.randn(1, E, device="cuda", generator=g)).to(dt)
    w = (torch.randn(N, E, device="cuda", generator=g) * 0.1).to(dt)
    nw = 1 + 0.3 * torch.randn(E, device="cuda", generator=g); nb = 0.3 * torch.randn(E, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    cs = torch.randn(N, device="cuda", generator=g); ch = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g).to(dt)
    norm = variant.startswith("norm")
    if norm:
        mean, rstd, sc, sh = K.in_stats(a, frames, S, E, nw, nb)
        xn = torch.empty_like(a)
        L.check(L.lib().bf_affine_apply(L.BF_DTYPE_BF16, a.data_ptr(), None, sc.data_ptr(), sh.data_ptr(), xn.data_ptr(), M, S, E, torch.cuda.current_stream().cuda_stream), "affine")
        af = a.float().view(frames, S, E)
        ref_in = ((af - af.mean(1, keepdim=True)) * torch.rsqrt(af.var(1, unbiased=False, keepdim=True) + 1e-5) * nw + nb).view(M, E)
    else:
        xn, ref_in = a, a.float()
    sep = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
    ref = ref_in @ w.float().t()
    if variant == "norm_bias":
        got = K.frame_linear(a, w, frames, S, norm=(nw, nb), bias=bias)
        K.gemm(dt, M, N, E, K.operand(xn, E), K.operand(w, E), K.epilogue(sep, N, bias=bias))
        ref = ref + bias
    elif variant in ("norm_fold_resid", "plain_fold_resid"):
        got = K.frame_linear(a, w, frames, S, norm=(nw, nb) if norm else None, colscale=cs, colshift=ch, resid=resid)
        K.gemm(dt, M, N, E, K.operand(xn, E), K.operand(w, E), K.epilogue(sep, N, colscale=cs, colshift=ch, aux_mode=L.BF_AUX_ADD, aux=resid, ld_aux=N))
        ref = ref * cs + ch + resid.float()
    else:
        got = K.frame_linear(a, w, frames, S, bias=bias, gelu=True)
        pre = torch.empty_like(sep)
        K.gemm(dt, M, N, E, K.operand(xn, E), K.operand(w, E), K.epilogue(pre, N, bias=bias, gelu_out=sep))
        ref = torch.nn.functional.gelu(ref + bias)
    assert got is not None, "bf_frame_linear refused a covered shape"
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert _rel(got.float(), ref) < 6e-3
    if "fold" not in variant:     # (the fold rescales and shifts the product: its element-wise check is the bit-equality below)
        assert float(((got.float() - ref).abs() / (ref.abs() + 0.05 * ref.abs().mean())).max()) < 0.3      # the bf16 rounding of the normalised operand included
    assert torch.equal(got, sep), "differs from the separate launches: %g of the elements" % float((got != sep).float().mean())
    if variant == "norm_fold_resid":       # the temporal out-projection also emits the next block's norm1 of its output
        xw = 1 + 0.3 * torch.randn(N, device="cuda", generator=g); xb = 0.3 * torch.randn(N, device="cuda", generator=g)
        got_o, got_n = K.frame_linear(a, w, frames, S, norm=(nw, nb), colscale=cs, colshift=ch, resid=resid, next_norm=(xw, xb))
        _, _, sc2, sh2 = K.in_stats(sep, frames, S, N, xw, xb)
        sep_n = torch.empty_like(sep)
        L.check(L.lib().bf_affine_apply(L.BF_DTYPE_BF16, sep.data_ptr(), None, sc2.data_ptr(), sh2.data_ptr(), sep_n.data_ptr(), M, S, N, torch.cuda.current_stream().cuda_stream), "affine")
        assert torch.equal(got_o, sep) and torch.equal(got_n, sep_n), float((got_n != sep_n).float().mean())


@pytest.mark.parametrize("frames,N", [(16, 384), (3, 384), (64, 384), (24, 512)])
def test_frame_linear_fc2_with_the_instance_norm_behind(K, frames, N):
    """Streamed form (K = 4E = 1536, ring of operand slots) with the InstanceNorm of its own output in the epilogue:
    out = resid + gamma * IN(hid @ W2^T + b2) against fp32 torch and bit for bit against bf_gemm -> bf_in_stats(g = gamma) -> bf_affine_apply."""
    from bubbleformer_amd import _lib as L
    dt = torch.bfloat16
    S, Kd = 144, 1536
    M = frames * S
    g = torch.Generator(device="cuda").manual_seed(22)
    hid = torch.randn(M, Kd, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, Kd, device="cuda", generator=g) * 0.05).to(dt)
    bias = torch.randn(N, device="cuda", generator=g)
    ew = 1 + 0.3 * torch.randn(N, device="cuda", generator=g); eb = 0.3 * torch.randn(N, device="cuda", generator=g)
    eg = 0.5 * torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g).to(dt)
    got = K.frame_linear(hid, w, frames, S, bias=bias, resid=resid, out_norm=(ew, eb, eg))
    assert got is not None
    z = torch.empty(M, N, device="cuda", dtype=dt)
    K.gemm(dt, M, N, Kd, K.operand(hid, Kd), K.operand(w, Kd), K.epilogue(z, N, bias=bias))
    mean, rstd, sc, sh = K.in_stats(z, frames, S, N, ew, eb, g=eg.view(1, N).contiguous(), gdiv=frames)
    sep = torch.empty_like(z)
    L.check(L.lib().bf_affine_apply(L.BF_DTYPE_BF16, z.data_ptr(), resid.data_ptr(), sc.data_ptr(), sh.data_ptr(), sep.data_ptr(), M, S, N, torch.cuda.current_stream().cuda_stream), "affine")
    zf = (hid.float() @ w.float().t() + bias).view(frames, S, N)
    ref = (resid.float().view(frames, S, N) + eg * ((zf - zf.mean(1, keepdim=True)) * torch.rsqrt(zf.var(1, unbiased=False, keepdim=True) + 1e-5) * ew + eb)).view(M, N)
    torch.cuda.synchronize()
    assert _rel(got.float(), ref) < 6e-3
    assert torch.equal(got, sep), "differs from the separate launches: %g of the elements" % float((got != sep).float().mean())
    # second output: the next layer's norm1 of `out`, as bf_in_stats + bf_affine_apply make it from the stored tensor
    xw = 1 + 0.3 * torch.randn(N, device="cuda", generator=g); xb = 0.3 * torch.randn(N, device="cuda", generator=g)
    got_o, got_n = K.frame_linear(hid, w, frames, S, bias=bias, resid=resid, out_norm=(ew, eb, eg), next_norm=(xw, xb))
    _, _, sc2, sh2 = K.in_stats(sep, frames, S, N, xw, xb)
    sep_n = torch.empty_like(sep)
    L.check(L.lib().bf_affine_apply(L.BF_DTYPE_BF16, sep.data_ptr(), None, sc2.data_ptr(), sh2.data_ptr(), sep_n.data_ptr(), M, S, N, torch.cuda.current_stream().cuda_stream), "affine")
    assert torch.equal(got_o, sep) and torch.equal(got_n, sep_n), float((got_n != sep_n).float().mean())
    # plain streamed form (no norm): bias + residual
    got2 = K.frame_linear(hid, w, frames, S, bias=bias, resid=resid)
    sep2 = torch.empty_like(z)
    K.gemm(dt, M, N, Kd, K.operand(hid, Kd), K.operand(w, Kd), K.epilogue(sep2, N, bias=bias, aux_mode=L.BF_AUX_ADD, aux=resid, ld_aux=N))
    assert torch.equal(got2, sep2)


@pytest.mark.parametrize("Nout,Kin,M,with_cs", [(128, 128, 64, True), (384, 128, 64 * 7, True), (256, 384, 64 * 13, False),
                                                 (1152, 384, 2304, True), (384, 192, 32 * 4, True), (384, 192, 32 * 7, True),
                                                 (768, 384, 32 * 50, False), (384, 1536, 32 * 61, True), (1536, 384, 18432, True), (192, 192, 32 * 9, True),
                                                 (576, 384, 32 * 33, True)])
def test_gemm_tokred_slabs_match_fp64_and_are_bit_reproducible(K, Nout, Kin, M, with_cs):
    """Weight-gradient GEMM (LDS-DMA ring, token slices, slab reduction): out (+)= dy^T x and colsum(dy) against fp64, through the
    384 x 192 ping-pong kernel (shapes that tile by it: one slice up to twelve, odd half-step counts, ragged last slice) and the
    128 x 128 kernel (the rest); two runs are bit-identical (no float atomics)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    dy = torch.randn(M, Nout, device="cuda", generator=g).bfloat16()
    x = torch.randn(M, Kin, device="cuda", generator=g).bfloat16()
    base = torch.randn(Nout, Kin, device="cuda", generator=g)
    cs0 = torch.randn(Nout, device="cuda", generator=g)
    outs = []
    for rep in range(2):
        out = base.clone()
        cs = cs0.clone() if with_cs else None
        assert K.gemm_tokred(dy, x, out, accumulate=True, colsum=cs)
        outs.append((out, cs))
    ref = base.double() + dy.double().t() @ x.double()
    assert _rel(outs[0][0], ref) < 2e-6
    assert torch.equal(outs[0][0], outs[1][0])
    if with_cs:
        assert _rel(outs[0][1], cs0.double() + dy.double().sum(0)) < 2e-6
        assert torch.equal(outs[0][1], outs[1][1])
    out = torch.full((Nout, Kin), 7.0, device="cuda")          # accumulate = 0 overwrites
    cs = torch.full((Nout,), 7.0, device="cuda")
    assert K.gemm_tokred(dy, x, out, accumulate=False, colsum=cs)
    assert _rel(out, dy.double().t() @ x.double()) < 2e-6 and _rel(cs, dy.double().sum(0)) < 2e-6
    # shapes it does not take are declined, not mangled
    assert not K.gemm_tokred(dy[:, :120].contiguous(), x, torch.zeros(120, Kin, device="cuda"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_patch_gather_and_scatter(K, dtype):
    """k2s2 conv as a patch-gather GEMM and k2s2 transposed conv as a scatter-store GEMM vs torch conv ops."""
    from bubbleformer_amd import _lib as L
    Fr, H, W, Ci, Co = 3, 8, 12, 16, 24
    g = torch.Generator(device="cuda").manual_seed(4)
    img = torch.randn(Fr, H, W, Ci, device="cuda", generator=g).to(dtype)           # channels-last
    wconv = torch.randn(Co, Ci, 2, 2, device="cuda", generator=g) * 0.2
    wprep = wconv.permute(0, 2, 3, 1).reshape(Co, 4 * Ci).contiguous().to(dtype)   # [Co][(ky,kx,ci)]
    gh, gw = H // 2, W // 2
    P = Fr * gh * gw
    out = torch.empty(P, Co, device="cuda", dtype=dtype)
    K.gemm(dtype, P, Co, 4 * Ci, K.operand(img, Ci, gw=gw, gh=gh, gc=Ci, seglen=2 * Ci, segstride=2 * gw * Ci), K.operand(wprep, 4 * Ci),
           K.epilogue(out, Co))
    ref = torch.nn.functional.conv2d(img.float().permute(0, 3, 1, 2), wprep.float().view(Co, 2, 2, Ci).permute(0, 3, 1, 2), stride=2)
    assert _rel(out.float().view(Fr, gh, gw, Co).permute(0, 3, 1, 2), ref) < TOL[dtype]
    # transposed conv: x [Fr, gh, gw, Co] -> [Fr, H, W, Ci] with wt [Co][Ci][2][2]
    wt = torch.randn(Co, Ci, 2, 2, device="cuda", generator=g) * 0.2
    wtp = wt.permute(2, 3, 1, 0).reshape(4 * Ci, Co).contiguous().to(dtype)        # [(ky,kx,ci)][Co]
    up = torch.empty(Fr, H, W, Ci, device="cuda", dtype=dtype)
    K.gemm(dtype, P, 4 * Ci, Co, K.operand(out, Co), K.operand(wtp, Co),
           K.epilogue(up, Ci, gw=gw, gh=gh, gc=Ci, seglen=2 * Ci, segstride=2 * gw * Ci))
    ref = torch.nn.functional.conv_transpose2d(out.float().view(Fr, gh, gw, Co).permute(0, 3, 1, 2),
                                               wtp.float().view(2, 2, Ci, Co).permute(3, 2, 0, 1), stride=2)
    assert _rel(up.float().permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Ci", [16, 96])      # 96: 4*Ci = 384 columns = three 128-column tiles whose channels wrap (column mod Ci)
def test_gemm_conv_weight_gradient_gathered_prologue(K, dtype, Ci):
    """dW of a k2s2 conv fed by GELU(InstanceNorm-affine(image)): token-reduction GEMM whose B operand is an outer-contiguous
    2x2 patch gather with an affine+GELU prologue.  The per-(frame, channel) table is indexed by patch COLUMN (ky, kx, ci), so
    tiles past the first need channel = column mod Ci (regression: embed.in_proj.{3,6}.weight gradients at E = 384)."""
    from bubbleformer_amd import _lib as L
    Fr, H, W, Co = 3, 16, 24, 40
    g = torch.Generator(device="cuda").manual_seed(6)
    img = torch.randn(Fr, H, W, Ci, device="cuda", generator=g).to(dtype)
    gh, gw = H // 2, W // 2
    P = Fr * gh * gw
    dy = torch.randn(P, Co, device="cuda", generator=g).to(dtype)
    sc = 1 + 0.3 * torch.randn(Fr, Ci, device="cuda", generator=g)
    sh = 0.3 * torch.randn(Fr, Ci, device="cuda", generator=g)
    out = torch.zeros(Co, 4 * Ci, device="cuda", dtype=torch.float32)
    for splitk in (1, 3):
        out.zero_()
        K.gemm(dtype, Co, 4 * Ci, P, K.operand(dy, Co, layout=L.BF_LAY_XC),
               K.operand(img, Ci, layout=L.BF_LAY_XC, gw=gw, gh=gh, gc=Ci, seglen=2 * Ci, segstride=2 * gw * Ci, pro=L.BF_PRO_AFFINE_GELU,
                         sc=sc, sh=sh, rows_per_frame=gh * gw, nch=Ci),
               K.epilogue(out, 4 * Ci, out_mode=L.BF_OUT_ATOMIC_F32), splitk=splitk)
        act = torch.nn.functional.gelu(img.float() * sc[:, None, None] + sh[:, None, None])
        if dtype == torch.bfloat16:
            act = act.bfloat16().float()
        patches = act.view(Fr, gh, 2, gw, 2, Ci).permute(0, 1, 3, 2, 4, 5).reshape(P, 4 * Ci)      # columns (ky, kx, ci)
        ref = dy.float().t() @ patches
        assert _rel(out, ref) < TOL[dtype], splitk


@pytest.mark.parametrize("S,Cc", [(37, 72), (1000, 72), (1000, 96)])     # 1000 tokens per frame: the sliced (frames x slices) path, ragged last
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])          # slice; 96 channels: the 96-channel x 192-thread geometry
def test_in_stats_two_pass(K, dtype, S, Cc):
    Fr = 5
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(Fr, S, Cc, device="cuda", generator=g) * 3 + 100.0).to(dtype)     # large mean: a one-pass variance would fail
    w = torch.randn(Cc, device="cuda", generator=g)
    b = torch.randn(Cc, device="cuda", generator=g)
    mean, rstd, sc, sh = K.in_stats(x, Fr, S, Cc, w, b)
    xf = x.float()
    mu = xf.mean(1)
    var = xf.var(1, unbiased=False)
    assert _rel(mean, mu) < 1e-6
    assert _rel(rstd, (var + 1e-5).rsqrt()) < 1e-5
    # the folded affine x*sc + sh loses ~eps*|mean|/std relative accuracy by construction; activations have |mean|/std = O(1)
    assert _rel(xf * sc[:, None] + sh[:, None], torch.nn.functional.instance_norm(xf.permute(0, 2, 1), weight=w, bias=b).permute(0, 2, 1)) < 2e-4
    x2 = (x.float() - 97.0).to(dtype)
    _, _, sc2, sh2 = K.in_stats(x2, Fr, S, Cc, w, b)
    x2f = x2.float()
    assert _rel(x2f * sc2[:, None] + sh2[:, None], torch.nn.functional.instance_norm(x2f.permute(0, 2, 1), weight=w, bias=b).permute(0, 2, 1)) < 5e-6


@pytest.mark.parametrize("gelu", [False, True])
@pytest.mark.parametrize("S,Cc", [(50, 40), (700, 40), (700, 192)])      # 700: sliced reduce / sum / apply path (ragged last slice), 50: one
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])       # workgroup per frame; 192 channels: the 96-channel slice geometry
def test_in_bwd_matches_autograd(K, dtype, S, Cc, gelu):
    """nn.InstanceNorm2d(affine) [+ GELU] backward, incl. the residual add and the parameter gradients."""
    Fr = 3
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.randn(Fr, S, Cc, device="cuda", generator=g) * 1.5 + 0.3).to(dtype)
    dy = torch.randn(Fr, S, Cc, device="cuda", generator=g).to(dtype)
    add = torch.randn(Fr, S, Cc, device="cuda", generator=g).to(dtype)
    w = torch.randn(Cc, device="cuda", generator=g)
    b = torch.randn(Cc, device="cuda", generator=g)
    mean, rstd, _, _ = K.in_stats(x, Fr, S, Cc, w, b)
    dx, dw, db = K.in_bwd(dy, x, Fr, S, Cc, mean, rstd, w, b, add=add, gelu=gelu)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = torch.nn.functional.instance_norm(xr.permute(0, 2, 1), weight=wr, bias=br, eps=1e-5).permute(0, 2, 1)
    if gelu:
        y = torch.nn.functional.gelu(y)
    (y * dy.double()).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert _rel(dx.double(), xr.grad + add.double()) < tol
    assert _rel(dw.double(), wr.grad) < tol
    assert _rel(db.double(), br.grad) < tol


@pytest.mark.parametrize("Kd,N,with_add", [(1152, 384, True), (384, 384, False), (64, 128, True), (1536, 256, False)])
def test_gemm_inbwd_frames_matches_gemm_then_in_bwd(K, Kd, N, with_add):
    """conv1x1 data gradient + InstanceNorm2d backward in one kernel (144-token frames) vs fp64 autograd and vs the two-kernel path."""
    _check_gemm_inbwd_frames(K, 5, Kd, N, with_add)


@pytest.mark.parametrize("Kd,N,with_add", [(1152, 384, True), (384, 384, False), (64, 128, True), (128, 256, True), (1536, 256, False)])
def test_gemm_inbwd_frame_pairs(K, Kd, N, with_add):
    """The same product on the frame-pair kernel (an even number of frames: two frames per workgroup, both operands by LDS-DMA into a
    3-slot ring, waves 4-7 one barrier behind waves 0-3, register epilogue): K of one, two, six, 18 and 24 ring steps."""
    _check_gemm_inbwd_frames(K, 6, Kd, N, with_add)


@pytest.mark.parametrize("M,N,Kd,with_add", [(576, 384, 1536, True), (288, 128, 128, False), (18432, 384, 1536, True), (2304, 256, 1152, False)])
def test_gemm_pair_plain_data_gradient(K, M, N, Kd, with_add):
    """bf_gemm with a K-contiguous A, an outer-contiguous weight and 288-row tiles (fc1's data gradient `dpre @ W1 + dout`,
    layers/linear_layers.py:18-25) runs on the frame-pair kernel: against fp64 on the same bf16-valued operands."""
    g = torch.Generator(device="cuda").manual_seed(33)
    A = (torch.randn(M, Kd, device="cuda", generator=g) * 0.5).bfloat16()
    W = (torch.randn(Kd, N, device="cuda", generator=g) / Kd ** 0.5).bfloat16()
    add = torch.randn(M, N, device="cuda", generator=g).bfloat16() if with_add else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    h = K.L.lib()
    h.bf_prof_enable(1)
    K.gemm(torch.bfloat16, M, N, Kd, K.operand(A, Kd, K.L.BF_LAY_KC), K.operand(W, N, K.L.BF_LAY_XC),
           K.epilogue(out, N, **(dict(aux_mode=K.L.BF_AUX_ADD, aux=add, ld_aux=N) if with_add else {})))
    torch.cuda.synchronize()
    import ctypes, json
    buf = ctypes.create_string_buffer(1 << 14)
    h.bf_prof_report(buf, len(buf))
    h.bf_prof_enable(0)
    assert any(k.startswith("gemm_pair") for k in json.loads(buf.value.decode())), "the frame-pair kernel did not take this shape"
    ref = A.double() @ W.double() + (add.double() if with_add else 0.0)
    assert _rel(out.double(), ref) < 4e-3                      # one bf16 rounding of the result


def test_gemm_inbwd_frames_at_the_bench_size(K):
    """BASELINE configs[1]: 128 frames (batch 8 x 16), E = 384, the QKV projection's data gradient (K = 1152) with the residual."""
    _check_gemm_inbwd_frames(K, 128, 1152, 384, True)


@pytest.mark.parametrize("Kd,N,with_add", [(1152, 384, True), (384, 128, False)])
def test_gemm_inbwd_whole_288_token_frames(K, Kd, N, with_add):
    """Frames of 288 tokens (24 x 12 grids: BASELINE configs[3]) are ONE tile of the frame-pair kernel: the two wave groups hold the
    frame's halves and exchange their column sums through LDS."""
    _check_gemm_inbwd_frames(K, 5, Kd, N, with_add, S=288)


def _check_gemm_inbwd_frames(K, Fr, Kd, N, with_add, S=144):
    M = Fr * S
    g = torch.Generator(device="cuda").manual_seed(21)
    A = (torch.randn(M, Kd, device="cuda", generator=g) * 0.5).bfloat16()
    W = (torch.randn(Kd, N, device="cuda", generator=g) / Kd ** 0.5).bfloat16()
    x = (torch.randn(Fr, S, N, device="cuda", generator=g) * 1.5 + 0.3).bfloat16()
    add = torch.randn(M, N, device="cuda", generator=g).bfloat16() if with_add else None
    w = torch.randn(N, device="cuda", generator=g)
    b = torch.randn(N, device="cuda", generator=g)
    mean, rstd, _, _ = K.in_stats(x, Fr, S, N, w, b)
    res = K.gemm_inbwd_frames(A, W, x.view(M, N), S, mean, rstd, w, add=add)
    assert res is not None
    dx, ws = res
    torch.cuda.synchronize()
    # fp64 autograd on the same (bf16-valued) operands
    dy = A.double() @ W.double()
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = torch.nn.functional.instance_norm(xr.permute(0, 2, 1), weight=wr, bias=br, eps=1e-5).permute(0, 2, 1)
    (y * dy.view(Fr, S, N)).sum().backward()
    ref = xr.grad.view(M, N) + (add.double() if with_add else 0.0)
    assert _rel(dx.double(), ref) < 6e-3                      # one bf16 rounding of the result
    s = ws.view(Fr, N, 2).double()
    assert _rel(s[..., 1].sum(0), wr.grad) < 2e-3 and _rel(s[..., 0].sum(0), br.grad) < 2e-3
    # the two-kernel path rounds dy to bf16 in between: same answer to bf16 accuracy
    dyb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    K.gemm(torch.bfloat16, M, N, Kd, K.operand(A, Kd, K.L.BF_LAY_KC), K.operand(W, N, K.L.BF_LAY_XC), K.epilogue(dyb, N))
    dx2, dw2, db2 = K.in_bwd(dyb.view(Fr, S, N), x, Fr, S, N, mean, rstd, w, b, add=add.view(Fr, S, N) if with_add else None)
    assert _rel(dx.double(), dx2.view(M, N).double()) < 1.5e-2
    # per-frame-group factor on dy (stochastic depth: the branch gradient of frame f is fscale[f / fdiv] * dy), applied to the accumulators
    fdiv = 2
    m = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25][:(Fr + fdiv - 1) // fdiv] * ((Fr + 9) // 10 + 1), device="cuda")[:(Fr + fdiv - 1) // fdiv].contiguous()
    dx3, ws3 = K.gemm_inbwd_frames(A, W, x.view(M, N), S, mean, rstd, w, add=add, fscale=m, fdiv=fdiv)
    mrow = m.repeat_interleave(fdiv)[:Fr].repeat_interleave(S)[:, None].double()
    xr2 = x.double().requires_grad_(True)
    y2 = torch.nn.functional.instance_norm(xr2.permute(0, 2, 1), weight=w.double(), bias=b.double(), eps=1e-5).permute(0, 2, 1)
    (y2 * (dy * mrow).view(Fr, S, N)).sum().backward()
    assert _rel(dx3.double(), xr2.grad.view(M, N) + (add.double() if with_add else 0.0)) < 6e-3
    # shapes outside the whole-frame form are refused, not mis-computed
    assert K.gemm_inbwd_frames(A[:144 * 2], W, x.view(M, N)[:144 * 2], 72, mean, rstd, w) is None
    assert K.gemm_inbwd_frames(A.float(), W.float(), x.view(M, N).float(), S, mean, rstd, w) is None


@pytest.mark.parametrize("Fr,Kd,N,with_g", [(4, 1152, 384, True), (6, 384, 256, False)])
def test_gemm_inbwd_frame_pairs_with_a_chained_second_norm(K, Fr, Kd, N, with_g):
    """bf_gemm_inbwd_frames_chain: the frame-pair data-gradient kernel also applies the backward of the InstanceNorm whose output gradient it
    has just produced -- against the unchained kernel followed by bf_in_bwd on its output (same bf16 rows in, so only the order of the
    frame sums differs)."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    S, M = 144, Fr * 144
    g = torch.Generator(device="cuda").manual_seed(61)
    A = (torch.randn(M, Kd, device="cuda", generator=g) * 0.5).bfloat16()
    W = (torch.randn(Kd, N, device="cuda", generator=g) / Kd ** 0.5).bfloat16()
    x = (torch.randn(Fr, S, N, device="cuda", generator=g) * 1.5 + 0.3).bfloat16()
    add = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    w, b = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    mean, rstd, _, _ = K.in_stats(x, Fr, S, N, w, b)
    z = (torch.randn(Fr, S, N, device="cuda", generator=g) * 0.7 - 0.2).bfloat16()
    w3, b3 = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    mean3, rstd3, _, _ = K.in_stats(z, Fr, S, N, w3, b3)
    gt = (0.5 + torch.rand(Fr, N, device="cuda", generator=g)) if with_g else None
    ref_dx, ref_ws = K.gemm_inbwd_frames(A, W, x.view(M, N), S, mean, rstd, w, add=add)
    ref_dz, ref_dw3, ref_db3 = K.in_bwd(ref_dx.view(Fr, S, N), z, Fr, S, N, mean3, rstd3, w3, b3, g=gt, gdiv=1)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    dz = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(Fr * N * 2, device="cuda")
    cws = torch.full((Fr * N * 2,), float("nan"), device="cuda")
    rc = lib.bf_gemm_inbwd_frames_chain(1, M, N, Kd, _p(A), A.stride(0), _p(W), W.stride(0), _p(x), _p(add), _p(out), S, _p(mean), _p(rstd), _p(w), _p(ws),
                                        None, 1, _p(z), _p(dz), _p(mean3), _p(rstd3), _p(w3), _p(gt), 1, _p(cws), _stream())
    L.check(rc, "bf_gemm_inbwd_frames_chain")
    assert torch.equal(out, ref_dx) and torch.equal(ws, ref_ws)
    assert torch.isfinite(dz.float()).all() and _rel(dz, ref_dz.view(M, N)) < 4e-3
    s = cws.view(Fr, N, 2)
    gsum = gt if with_g else torch.ones(Fr, N, device="cuda")
    assert _rel((gsum * s[..., 1]).sum(0), ref_dw3) < 1e-3 and _rel((gsum * s[..., 0]).sum(0), ref_db3) < 1e-3
    # 288-token frames (one per tile): declined
    assert lib.bf_gemm_inbwd_frames_chain(1, 288 * 2, N, Kd, _p(A), A.stride(0), _p(W), W.stride(0), _p(x), _p(add), _p(out), 288, _p(mean), _p(rstd), _p(w), _p(ws),
                                          None, 1, _p(z), _p(dz), _p(mean3), _p(rstd3), _p(w3), _p(gt), 1, _p(cws), _stream()) == 1


@pytest.mark.parametrize("Fr,Kd,N,kind", [(16, 384, 384, "outproj"), (16, 1536, 384, "fc2"), (16, 1536, 384, "fc2_chain"), (128, 384, 384, "outproj"),
                                          (128, 1536, 384, "fc2_chain"), (16, 128, 256, "fc2_chain")])
def test_gemm_fwd_frames_with_folded_norms_changes_no_bit(K, Fr, Kd, N, kind):
    """bf_gemm_fwd_frames (the forward twin of the frame-pair kernel): projection + linear epilogue, then the InstanceNorm(s) of the rows as
    stored folded into the same launch -- `outproj`: out = x + (alpha * (on @ W^T) + beta) * drop[f] and the NEXT stage's norm1 of it
    (layers/attention.py:121-123 + :208); `fc2`: z = hid @ W2^T + b2, out = x1 + g * IN(z) (:312-317), `fc2_chain`: plus the next stage's norm1
    of out (:77).  Against bf_gemm (streaming kernels) + bf_in_stats + bf_affine_apply on the same operands: every tensor bit for bit."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    import ctypes as C
    lib = L.lib()
    S, M = 144, Fr * 144
    g = torch.Generator(device="cuda").manual_seed(71)
    A = (torch.randn(M, Kd, device="cuda", generator=g) * 0.7).bfloat16()
    W = (torch.randn(N, Kd, device="cuda", generator=g) / Kd ** 0.5).bfloat16()          # [out][in], as nn.Linear / Conv2d 1x1 store it
    Wt = W.t().contiguous()
    resid = (torch.randn(M, N, device="cuda", generator=g) * 1.3 + 0.2).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    alpha, beta = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    drop = (torch.rand(Fr // 4, device="cuda", generator=g) > 0.3).float() / 0.7          # one factor per group of 4 frames
    w1, b1 = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    w2, b2 = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    gt = 0.5 + torch.rand(Fr, N, device="cuda", generator=g)                             # per-(frame, channel) post scale (layer scale x drop)

    def stats_apply(x, w, b, gpost=None, res=None):
        mean, rstd, sc, sh = K.in_stats(x.view(Fr, S, N), Fr, S, N, w, b, g=gpost, gdiv=1)
        out = torch.empty_like(x)
        L.check(lib.bf_affine_apply(1, _p(x), _p(res), _p(sc), _p(sh), _p(out), M, S, N, _stream()), "bf_affine_apply")
        return out, (mean, rstd, sc, sh)

    def new_stats():
        return [torch.full((Fr, N), float("nan"), device="cuda") for _ in range(4)]

    def norm_rec(w, b, gpost, st, res, out):
        return L.FrameNorm(_p(w), _p(b), _p(gpost), 1, _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _p(res), _p(out))

    nan = lambda: torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    if kind == "outproj":
        ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        K.gemm(torch.bfloat16, M, N, Kd, K.operand(A, Kd), K.operand(W, Kd),
               K.epilogue(ref, N, colscale=alpha, colshift=beta, aux_mode=L.BF_AUX_ADD, aux=resid, ld_aux=N, rowscale=drop, rows_per_group=4 * S))
        ref_xn, ref_st = stats_apply(ref, w2, b2)
        out, xn, st2 = nan(), nan(), new_stats()
        n2 = norm_rec(w2, b2, None, st2, None, xn)
        rc = lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, None, _p(alpha), _p(beta), _p(drop), 4, _p(resid), _p(out), S, None, C.byref(n2), _stream())
        L.check(rc, "bf_gemm_fwd_frames")
        assert torch.equal(out, ref) and torch.equal(xn, ref_xn)
        assert all(torch.equal(a, b) for a, b in zip(st2, ref_st))
    else:
        z_ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        K.gemm(torch.bfloat16, M, N, Kd, K.operand(A, Kd), K.operand(W, Kd), K.epilogue(z_ref, N, bias=bias))
        out_ref, st1_ref = stats_apply(z_ref, w1, b1, gpost=gt, res=resid)
        z, out, xn, st1, st2 = nan(), nan(), nan(), new_stats(), new_stats()
        n1 = norm_rec(w1, b1, gt, st1, resid, out)
        n2 = norm_rec(w2, b2, None, st2, None, xn)
        chain = kind == "fc2_chain"
        rc = lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, _p(bias), None, None, None, 1, None, _p(z), S, C.byref(n1), C.byref(n2) if chain else None,
                                    _stream())
        L.check(rc, "bf_gemm_fwd_frames")
        assert torch.equal(z, z_ref) and torch.equal(out, out_ref)
        assert all(torch.equal(a, b) for a, b in zip(st1, st1_ref))
        if chain:
            xn_ref, st2_ref = stats_apply(out_ref, w2, b2)
            assert torch.equal(xn, xn_ref) and all(torch.equal(a, b) for a, b in zip(st2, st2_ref))
    # an fp64 check of the statistics themselves (the references above share the kernels' arithmetic)
    last = (out if kind != "outproj" else out).double().view(Fr, S, N)
    if kind != "fc2":
        assert _rel(st2[0], last.mean(1)) < 1e-5 and _rel(st2[1], (last.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    # declined: fp32, an odd number of frames, 288-token frames
    assert lib.bf_gemm_fwd_frames(0, M, N, Kd, _p(A), Kd, _p(Wt), N, None, None, None, None, 1, None, _p(nan()), S, None, None, _stream()) == 1
    assert lib.bf_gemm_fwd_frames(1, M - 144, N, Kd, _p(A), Kd, _p(Wt), N, None, None, None, None, 1, None, _p(nan()), S, None, None, _stream()) == 1
    assert lib.bf_gemm_fwd_frames(1, M, N, Kd, _p(A), Kd, _p(Wt), N, None, None, None, None, 1, None, _p(nan()), 288, None, None, _stream()) == 1


def _attn_call(L_lib, qkv, dout, geo, heads, d, prm, generic):
    """Run attention fwd + bwd through the C ABI; returns (out, dqkv, param grads)."""
    import ctypes as C
    from bubbleformer_amd.ops import _dt, _p, _stream
    from bubbleformer_amd import _lib as L
    h = L.lib()
    h.bf_debug_force_generic_attn(1 if generic else 0)
    try:
        N = qkv.shape[0]
        E = heads * d
        out = torch.zeros(N, E, device="cuda", dtype=qkv.dtype)
        dqkv = torch.zeros_like(qkv)
        grads = [torch.zeros_like(t) for t in prm]
        nseq, Lq, inner, ostr, istr, tstr = geo
        L.check(h.bf_attn_fwd(_dt(qkv.dtype), _p(qkv), _p(out), nseq, Lq, inner, ostr, istr, tstr, heads, d, *[_p(t) for t in prm], 0.5, 0, _stream()), "fwd")
        L.check(h.bf_attn_bwd(_dt(qkv.dtype), _p(qkv), _p(dout), _p(dqkv), nseq, Lq, inner, ostr, istr, tstr, heads, d, *[_p(t) for t in prm],
                              *[_p(t) for t in grads], 0.5, 0, None, 0, _stream()), "bwd")
        torch.cuda.synchronize()
        return out, dqkv, grads
    finally:
        h.bf_debug_force_generic_attn(0)


@pytest.mark.parametrize("L,d,heads", [(12, 64, 6), (16, 64, 6), (24, 64, 2), (32, 64, 3), (7, 32, 2), (20, 128, 1)])
@pytest.mark.parametrize("axis", ["contig", "strided"])
def test_attention_mfma_matches_generic_and_fp32(L, d, heads, axis):
    """bf16 MFMA attention (fwd + bwd) vs (a) the generic fp32-VALU kernel on the same bf16 inputs, (b) the generic
    kernel on fp32 copies of the inputs (the parity-mode kernel, itself pinned by the golden model tests)."""
    g = torch.Generator(device="cuda").manual_seed(100 + L + d)
    E = heads * d
    inner_n = 5
    if axis == "contig":          # sequences of L contiguous tokens
        nseq = 14
        geo = (nseq, L, 1, L, 0, 1)
    else:                         # token = (s / inner) * L * inner + s % inner + l * inner
        nseq = 3 * inner_n
        geo = (nseq, L, inner_n, L * inner_n, 1, inner_n)
    N = nseq * L
    qkv = (torch.randn(N, 3 * E, device="cuda", generator=g) * 1.5).bfloat16()
    dout = torch.randn(N, E, device="cuda", generator=g).bfloat16()
    prm = [1 + 0.2 * torch.randn(d, device="cuda", generator=g), 0.2 * torch.randn(d, device="cuda", generator=g),
           1 + 0.2 * torch.randn(d, device="cuda", generator=g), 0.2 * torch.randn(d, device="cuda", generator=g),
           0.5 * torch.randn(32, heads, device="cuda", generator=g), 1 + 0.3 * torch.randn(heads, device="cuda", generator=g)]
    o_m, dq_m, g_m = _attn_call(None, qkv, dout, geo, heads, d, prm, generic=False)
    o_g, dq_g, g_g = _attn_call(None, qkv, dout, geo, heads, d, prm, generic=True)
    o_f, dq_f, g_f = _attn_call(None, qkv.float(), dout.float(), geo, heads, d, prm, generic=True)
    assert torch.isfinite(o_m.float()).all() and torch.isfinite(dq_m.float()).all()
    assert _rel(o_m.float(), o_f) < 1.5e-2 and _rel(o_g.float(), o_f) < 1.5e-2
    parts = lambda t: t.float().view(N, heads, 3, d)
    for pi, pn in enumerate("qkv"):
        e = _rel(parts(dq_m)[:, :, pi], parts(dq_f)[:, :, pi])
        assert e < 3e-2, (pn, e)
    for a, b, name in zip(g_m, g_f, ("dqw", "dqb", "dkw", "dkb", "demb", "dhscale")):
        if name == "dkb":     # structurally zero (softmax is shift invariant): bf16 rounding noise around 0, absolute bound
            assert float((a - b).norm()) < 1e-2 * float(g_f[0].norm()), (name, float((a - b).norm()), float(g_f[0].norm()))
        else:
            assert float((a - b).norm()) / float(b.norm()) < (1e-1 if name == "dhscale" else 5e-2), (name, float((a - b).norm()) / float(b.norm()))


@pytest.mark.parametrize("h,w,d,heads", [(12, 12, 64, 6), (6, 10, 64, 2), (16, 12, 32, 3), (24, 12, 64, 2)])
def test_axial_attention_backward_raw_pair_of_passes(h, w, d, heads):
    """bf_attn_bwd's raw-gradient modes (accumulate 2, then 5): the W and H passes of the axial block share one q / k LayerNorm
    (layers/attention.py:213-214), whose backward is linear in its incoming gradient -- the W pass leaves the gradients with respect to the
    LayerNorm outputs, the H pass adds its own and runs the LayerNorm backward and the parameter sums once.  Against the two plain passes
    (accumulate 0, then 1), which run them twice: data gradients and every parameter gradient equal to bf16 rounding of the intermediate."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    g = torch.Generator(device="cuda").manual_seed(7 + h + w)
    Fr, E = 5, heads * d
    N = Fr * h * w
    qkv = (torch.randn(N, 3 * E, device="cuda", generator=g) * 1.5).bfloat16()
    dout = torch.randn(N, E, device="cuda", generator=g).bfloat16()
    prm = [1 + 0.2 * torch.randn(d, device="cuda", generator=g), 0.2 * torch.randn(d, device="cuda", generator=g),
           1 + 0.2 * torch.randn(d, device="cuda", generator=g), 0.2 * torch.randn(d, device="cuda", generator=g),
           0.5 * torch.randn(32, heads, device="cuda", generator=g), 1 + 0.3 * torch.randn(heads, device="cuda", generator=g)]
    geoW = (Fr * h, w, 1, w, 0, 1)
    geoH = (Fr * w, h, w, h * w, 1, w)

    def run(a1, a2):
        dqkv = torch.full_like(qkv, float("nan"))
        grads = [torch.zeros_like(t) for t in prm]
        for geo, acc in ((geoW, a1), (geoH, a2)):
            L.check(lib.bf_attn_bwd(1, _p(qkv), _p(dout), _p(dqkv), *geo, heads, d, *[_p(t) for t in prm], *[_p(t) for t in grads], 0.5, acc, None, 0,
                                    _stream()), "bf_attn_bwd")
        torch.cuda.synchronize()
        return dqkv, grads

    ref, gref = run(0, 1)
    raw, graw = run(2, 5)
    assert torch.isfinite(raw.float()).all()
    parts = lambda t: t.float().view(N, heads, 3, d)
    for pi, pn in enumerate("qkv"):
        e = _rel(parts(raw)[:, :, pi], parts(ref)[:, :, pi])
        assert e < (1e-6 if pn == "v" else 8e-3), (pn, e)          # v: the same arithmetic; q, k: one bf16 rounding placed differently
    assert torch.equal(parts(raw)[:, :, 2], parts(ref)[:, :, 2])
    for a, b, name in zip(graw, gref, ("dqw", "dqb", "dkw", "dkb", "demb", "dhscale")):
        if name == "dkb":     # structurally zero: absolute bound
            assert float((a - b).norm()) < 1e-2 * float(gref[0].norm()), name
        else:
            assert float((a - b).norm()) / float(b.norm()) < 5e-3, (name, float((a - b).norm()) / float(b.norm()))
    # the raw modes exist on the bf16 MFMA path only
    lib.bf_debug_force_generic_attn(1)
    try:
        assert lib.bf_attn_bwd(1, _p(qkv), _p(dout), _p(raw), *geoW, heads, d, *[_p(t) for t in prm], *[_p(t) for t in graw], 0.5, 2, None, 0, _stream()) < 0
    finally:
        lib.bf_debug_force_generic_attn(0)


@pytest.mark.parametrize("path", ["mfma_bf16", "generic_f32"])
@pytest.mark.parametrize("L", [4, 6, 8, 12, 16, 24, 32])
def test_attention_t5_buckets_bit_exact_on_device(K, path, L):
    """The device copies of the T5 bucket function (csrc/attn_mfma.hip: t5b, csrc/attn.hip: t5_bucket) against the reference's integer
    tables (tests/golden/relpos_tables.npz), read back through the attention forward itself: with q = k = 0 the scores are the bias
    alone, with V = one-hot(key) the output row is the softmax row, and with emb[b] = log(1 + b) the ratio P[q][k] / P[q][q] is
    1 + bucket(q - k) -- an integer, recovered exactly."""
    import os
    import numpy as np
    from bubbleformer_amd import _lib as Lb
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "relpos_tables.npz"))
    d, heads, nseq = 64, 1, 3
    dt = torch.bfloat16 if path == "mfma_bf16" else torch.float32
    qkv = torch.zeros(nseq * L, 3 * d, device="cuda", dtype=dt)
    for s_ in range(nseq):
        for l_ in range(L):
            qkv[s_ * L + l_, 2 * d + l_] = 1.0                      # v[key l] = e_l
    out = torch.empty(nseq * L, d, device="cuda", dtype=dt)
    ones, zeros = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    emb = torch.log1p(torch.arange(32, device="cuda", dtype=torch.float32)).view(32, 1).contiguous()
    Lb.lib().bf_debug_force_generic_attn(0 if path == "mfma_bf16" else 1)
    try:
        K.attn_fwd(qkv, out, nseq, L, 1, L, 0, 1, heads, d, ones, zeros, ones, zeros, emb, None)
    finally:
        Lb.lib().bf_debug_force_generic_attn(0)
    P = out.float().view(nseq, L, d)[:, :, :L]
    got = torch.round(P / torch.diagonal(P, dim1=1, dim2=2).unsqueeze(-1) - 1.0).long().cpu().numpy()
    for s_ in range(nseq):
        assert np.array_equal(got[s_], z[f"bucket_{L}"]), (L, s_)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gelu_mlp_layer_standalone(dtype):
    """layers.GeluMLP used on its own (linear_layers.py:5-25): forward and every gradient vs torch in fp64."""
    from bubbleformer_amd.layers import GeluMLP
    torch.manual_seed(4)
    m = GeluMLP(96).cuda()
    x = torch.randn(3, 7, 11, 96, device="cuda").to(dtype).requires_grad_(True)
    y = m(x)
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().double().requires_grad_(True)
    w1, b1, w2, b2 = (p.detach().double().requires_grad_(True) for p in (m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias))
    if dtype == torch.bfloat16:                      # the GEMMs see bf16 weights
        w1, w2 = (w.detach().bfloat16().double().requires_grad_(True) for w in (w1, w2))
    yr = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xr, w1, b1)), w2, b2)
    yr.backward(g.double())
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert y.shape == x.shape and y.dtype == dtype
    assert _rel(y.detach().double(), yr.detach()) < tol and _rel(x.grad.double(), xr.grad) < tol
    for got, want in ((m.fc1.weight.grad, w1.grad), (m.fc1.bias.grad, b1.grad), (m.fc2.weight.grad, w2.grad), (m.fc2.bias.grad, b2.grad)):
        assert _rel(got.double(), want) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_film_layer_standalone(dtype):
    """layers.FiLMMLP used on its own (linear_layers.py:49-77): gamma * x + beta and every gradient vs torch in fp64."""
    from bubbleformer_amd.layers import FiLMMLP
    torch.manual_seed(6)
    B, T, Cc, h, w, P = 3, 4, 64, 5, 6, 9
    m = FiLMMLP(P, Cc).cuda()
    x = torch.randn(B, T, Cc, h, w, device="cuda").to(dtype).requires_grad_(True)
    cond = torch.randn(B, P, device="cuda")
    y = m(x, cond)
    g = torch.randn_like(y)
    y.backward(g)
    ref = torch.nn.Sequential(torch.nn.LayerNorm(P), torch.nn.Linear(P, 2 * Cc)).cuda().double()
    ref.load_state_dict({k: v.double() for k, v in m.film_net.state_dict().items()})
    xr = x.detach().double().requires_grad_(True)
    gamma, beta = ref(cond.double()).chunk(2, dim=1)
    yr = gamma.view(-1, 1, Cc, 1, 1) * xr + beta.view(-1, 1, Cc, 1, 1)
    yr.backward(g.double())
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert y.shape == x.shape and _rel(y.detach().double(), yr.detach()) < tol and _rel(x.grad.double(), xr.grad) < tol
    for a, b in zip(m.film_net.parameters(), ref.parameters()):
        assert _rel(a.grad.double(), b.grad) < tol


@pytest.mark.parametrize("h,w,heads,d", [(12, 12, 6, 64), (5, 9, 3, 32), (16, 3, 2, 128), (1, 7, 2, 96)])
def test_axial_attention_forward_one_launch_equals_two_passes(h, w, heads, d):
    """bf_attn_axial_fwd (W then H in one launch, intermediate in LDS) vs the two bf_attn_fwd passes it replaces: bit-identical,
    and both agree with the fp32 generic kernels."""
    import ctypes as C
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    Fr, E = 7, heads * d
    N = Fr * h * w
    g = torch.Generator(device="cuda").manual_seed(17)
    qkv = torch.randn(N, 3 * E, device="cuda", generator=g).bfloat16()
    prm = [1 + 0.1 * torch.randn(d, device="cuda", generator=g), 0.1 * torch.randn(d, device="cuda", generator=g),
           1 + 0.1 * torch.randn(d, device="cuda", generator=g), 0.1 * torch.randn(d, device="cuda", generator=g),
           0.3 * torch.randn(32, heads, device="cuda", generator=g)]
    hx = 1 + 0.3 * torch.randn(heads, device="cuda", generator=g)
    hy = 1 + 0.3 * torch.randn(heads, device="cuda", generator=g)
    one = torch.zeros(N, E, device="cuda", dtype=torch.bfloat16)
    L.check(lib.bf_attn_axial_fwd(1, _p(qkv), _p(one), Fr, h, w, heads, d, *[_p(t) for t in prm], _p(hx), _p(hy), _stream()), "axial")
    two = torch.zeros_like(one)
    L.check(lib.bf_attn_fwd(1, _p(qkv), _p(two), Fr * h, w, 1, w, 0, 1, heads, d, *[_p(t) for t in prm], _p(hx), 0.5, 0, _stream()), "w")
    L.check(lib.bf_attn_fwd(1, _p(qkv), _p(two), Fr * w, h, w, h * w, 1, w, heads, d, *[_p(t) for t in prm], _p(hy), 0.5, 1, _stream()), "h")
    assert torch.equal(one, two)
    ref = torch.zeros(N, E, device="cuda")                        # fp32 generic kernels through the same entry point
    L.check(lib.bf_attn_axial_fwd(0, _p(qkv.float()), _p(ref), Fr, h, w, heads, d, *[_p(t) for t in prm], _p(hx), _p(hy), _stream()), "axial f32")
    assert _rel(one.double(), ref.double()) < 2e-2


@pytest.mark.parametrize("h,w,heads,d", [(12, 12, 6, 64), (5, 9, 3, 32), (16, 3, 2, 128), (4, 7, 2, 96)])
def test_axial_attention_forward_with_instance_norm(K, h, w, heads, d):
    """bf_attn_axial_norm_fwd: the attention output is bit-identical to bf_attn_axial_fwd, and out_n / mean / rstd / sc / sh equal what
    bf_in_stats + the affine apply give on that output (fp32 summation order aside)."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    Fr, E = 5, heads * d
    S = h * w
    N = Fr * S
    g = torch.Generator(device="cuda").manual_seed(23)
    qkv = torch.randn(N, 3 * E, device="cuda", generator=g).bfloat16()
    prm = [1 + 0.1 * torch.randn(d, device="cuda", generator=g), 0.1 * torch.randn(d, device="cuda", generator=g),
           1 + 0.1 * torch.randn(d, device="cuda", generator=g), 0.1 * torch.randn(d, device="cuda", generator=g),
           0.3 * torch.randn(32, heads, device="cuda", generator=g)]
    hx = 1 + 0.3 * torch.randn(heads, device="cuda", generator=g)
    hy = 1 + 0.3 * torch.randn(heads, device="cuda", generator=g)
    nw = 1 + 0.2 * torch.randn(E, device="cuda", generator=g)
    nb = 0.2 * torch.randn(E, device="cuda", generator=g)
    o = torch.zeros(N, E, device="cuda", dtype=torch.bfloat16)
    on = torch.zeros_like(o)
    mean, rstd, sc, sh = (torch.zeros(Fr, E, device="cuda") for _ in range(4))
    rc = lib.bf_attn_axial_norm_fwd(1, _p(qkv), _p(o), _p(on), Fr, h, w, heads, d, *[_p(t) for t in prm], _p(hx), _p(hy), _p(nw), _p(nb),
                                    _p(mean), _p(rstd), _p(sc), _p(sh), _stream())
    assert rc == 0
    ref = torch.zeros_like(o)
    L.check(lib.bf_attn_axial_fwd(1, _p(qkv), _p(ref), Fr, h, w, heads, d, *[_p(t) for t in prm], _p(hx), _p(hy), _stream()), "axial")
    assert torch.equal(o, ref)
    m2, r2, sc2, sh2 = K.in_stats(ref.view(Fr, S, E), Fr, S, E, nw, nb)
    assert _rel(mean, m2) < 1e-5 and _rel(rstd, r2) < 1e-5 and _rel(sc, sc2) < 1e-5 and torch.allclose(sh, sh2, rtol=1e-4, atol=1e-5)
    want = (ref.view(Fr, S, E).float() * sc2[:, None] + sh2[:, None]).view(N, E)
    assert _rel(on.float(), want) < 6e-3
    # a shape the one-launch form does not cover is refused, the caller then takes the two-step path
    assert lib.bf_attn_axial_norm_fwd(1, _p(qkv), _p(o), _p(on), 1, 20, 2, heads, d, *[_p(t) for t in prm], _p(hx), _p(hy), _p(nw), _p(nb),
                                      _p(mean), _p(rstd), _p(sc), _p(sh), _stream()) == 1


@pytest.mark.parametrize("Ci,Co,h,w", [(96, 4, 96, 96), (32, 3, 6, 16), (64, 1, 5, 32), (128, 4, 3, 48)])
def test_debed_last_stage_one_pass(Ci, Co, h, w):
    """bf_debed_last (InstanceNorm affine + GELU + ConvTranspose2d(k=2, s=2) + NCHW store + relative-L2 partial sums in one streaming
    kernel) against torch on the same bf16-rounded operands (layers/patching.py:92-104 last stage; utils/losses.py:79-89 sums), and the
    shapes it declines (return 1: the caller keeps GEMM + bf_pm2nchw)."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    Fr = 3
    g = torch.Generator(device="cuda").manual_seed(23)
    act = torch.randn(Fr * h * w, Ci, device="cuda", generator=g).bfloat16()
    sc = 1 + 0.2 * torch.randn(Fr, Ci, device="cuda", generator=g)
    sh = 0.3 * torch.randn(Fr, Ci, device="cuda", generator=g)
    Wt = (torch.randn(Ci, Co, 2, 2, device="cuda", generator=g) / Ci ** 0.5)
    wc = torch.zeros(Ci, 16, device="cuda")
    wc[:, :4 * Co] = Wt.reshape(Ci, 4 * Co)
    wc = wc.bfloat16()
    y = torch.randn(Fr, Co, 2 * h, 2 * w, device="cuda", generator=g)
    pred = torch.full((Fr, Co, 2 * h, 2 * w), float("nan"), device="cuda")
    lossbuf = torch.zeros(Fr, Co, 2, L.BF_LOSS_LIMBS, device="cuda", dtype=torch.int64)      # integer limbs: base-2^48 digits in units of 2^-112
    L.check(lib.bf_debed_last(1, _p(act), _p(sc), _p(sh), _p(wc), _p(pred), _p(y), _p(lossbuf), Fr, Ci, Co, h, w, 16, _stream()), "debed_last")
    a = torch.nn.functional.gelu(act.float().view(Fr, h * w, Ci) * sc[:, None] + sh[:, None]).bfloat16().float()
    a = a.view(Fr, h, w, Ci).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv_transpose2d(a.double(), wc[:, :4 * Co].double().view(Ci, Co, 2, 2), stride=2)
    assert torch.isfinite(pred).all()
    assert _rel(pred, ref) < 2e-3
    num = (pred.double() - y.double()).pow(2).sum(dim=(-1, -2))
    den = y.double().pow(2).sum(dim=(-1, -2))
    lossbuf = _loss_limbs_value(lossbuf)
    assert _rel(lossbuf[..., 0], num) < 1e-5 and _rel(lossbuf[..., 1], den) < 1e-5
    # without a target: prediction only
    pred2 = torch.zeros_like(pred)
    L.check(lib.bf_debed_last(1, _p(act), _p(sc), _p(sh), _p(wc), _p(pred2), None, None, Fr, Ci, Co, h, w, 16, _stream()), "debed_last")
    assert torch.equal(pred2, pred)
    # declined shapes
    assert lib.bf_debed_last(0, _p(act), _p(sc), _p(sh), _p(wc), _p(pred), None, None, Fr, Ci, Co, h, w, 16, _stream()) == 1      # fp32
    assert lib.bf_debed_last(1, _p(act), _p(sc), _p(sh), _p(wc), _p(pred), None, None, Fr, Ci, Co, h, w, 24, _stream()) == 1      # Np != 16
    assert lib.bf_debed_last(1, _p(act), _p(sc), _p(sh), _p(wc), _p(pred), None, None, Fr, Ci, Co, h * 2, w // 2 + 4, 16, _stream()) == 1      # w % 16


@pytest.mark.parametrize("Ci,Co,h,w", [(96, 4, 96, 96), (32, 3, 6, 16), (64, 1, 5, 32), (128, 4, 3, 48)])
@pytest.mark.parametrize("fused_loss", [False, True])
def test_debed_last_stage_backward_one_pass(Ci, Co, h, w, fused_loss):
    """bf_debed_last_bwd: patch-major loss gradient (given, or coef * gscale * (pred - y)) and the transposed convolution's data gradient
    in one kernel, against bf_nchw2pm's definition and a plain matmul on the same bf16 operands."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    Fr = 3
    g = torch.Generator(device="cuda").manual_seed(29)
    wc = torch.zeros(Ci, 16, device="cuda")
    wc[:, :4 * Co] = torch.randn(Ci, 4 * Co, device="cuda", generator=g) / 4
    wc = wc.bfloat16()
    pred = torch.randn(Fr, Co, 2 * h, 2 * w, device="cuda", generator=g)
    y = torch.randn(Fr, Co, 2 * h, 2 * w, device="cuda", generator=g)
    coef = 0.5 + torch.rand(Fr, Co, device="cuda", generator=g)
    gs = torch.tensor([0.7], device="cuda")
    dpred = coef[:, :, None, None] * gs * (pred - y)
    P = Fr * h * w
    dpm = torch.full((P, 16), float("nan"), device="cuda", dtype=torch.bfloat16)
    dact = torch.full((P, Ci), float("nan"), device="cuda", dtype=torch.bfloat16)
    if fused_loss:
        rc = lib.bf_debed_last_bwd(1, None, _p(pred), _p(y), _p(coef), _p(gs), _p(wc), _p(dpm), _p(dact), Fr, Ci, Co, h, w, 16, _stream())
    else:
        rc = lib.bf_debed_last_bwd(1, _p(dpred), None, None, None, None, _p(wc), _p(dpm), _p(dact), Fr, Ci, Co, h, w, 16, _stream())
    L.check(rc, "debed_last_bwd")
    ref_pm = torch.zeros(P, 16, device="cuda")
    ref_pm[:, :4 * Co] = dpred.view(Fr, Co, h, 2, w, 2).permute(0, 2, 4, 1, 3, 5).reshape(P, 4 * Co)       # n = co*4 + ky*2 + kx
    ref_pm = ref_pm.bfloat16()
    assert torch.equal(dpm, ref_pm)
    ref = ref_pm.double() @ wc.double().t()
    assert torch.isfinite(dact.float()).all()
    assert _rel(dact, ref) < 4e-3                      # one bf16 rounding of the result
    assert lib.bf_debed_last_bwd(0, _p(dpred), None, None, None, None, _p(wc), _p(dpm), _p(dact), Fr, Ci, Co, h, w, 16, _stream()) == 1


@pytest.mark.parametrize("Ci,Co,h,w", [(96, 4, 96, 96), (32, 3, 6, 16), (64, 1, 5, 32), (128, 4, 3, 48)])
@pytest.mark.parametrize("fused_loss", [False, True])
def test_debed_last_stage_backward_with_the_norm_in_front(Ci, Co, h, w, fused_loss):
    """bf_debed_last_bwd_norm: the last stage's data gradient, GELU' and the InstanceNorm backward of the map in front of it in two passes that
    never store the rank-16 gradient map, against autograd in fp64 through  InstanceNorm -> GELU -> @ wc  on the same bf16 operands
    (layers/patching.py:92-104); dpm bit-identical to bf_debed_last_bwd's.  Whole and ragged last slices (h*w / 16 groups of 16 per wave)."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    Fr = 3
    g = torch.Generator(device="cuda").manual_seed(41)
    wc = torch.zeros(Ci, 16, device="cuda")
    wc[:, :4 * Co] = torch.randn(Ci, 4 * Co, device="cuda", generator=g) / 4
    wc = wc.bfloat16()
    pred = torch.randn(Fr, Co, 2 * h, 2 * w, device="cuda", generator=g)
    y = torch.randn(Fr, Co, 2 * h, 2 * w, device="cuda", generator=g)
    coef = 0.5 + torch.rand(Fr, Co, device="cuda", generator=g)
    gs = torch.tensor([0.7], device="cuda")
    dpred = coef[:, :, None, None] * gs * (pred - y)
    S, P = h * w, Fr * h * w
    ymap = (torch.randn(P, Ci, device="cuda", generator=g) * 1.5 + 0.2).bfloat16()
    in_w, in_b = 1 + 0.2 * torch.randn(Ci, device="cuda", generator=g), 0.3 * torch.randn(Ci, device="cuda", generator=g)
    yf = ymap.float().view(Fr, S, Ci)
    mean, rstd = yf.mean(1).contiguous(), (yf.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dpm = torch.full((P, 16), float("nan"), device="cuda", dtype=torch.bfloat16)
    dx = torch.full((P, Ci), float("nan"), device="cuda", dtype=torch.bfloat16)
    dw, db = torch.full((Ci,), 2.0, device="cuda"), torch.full((Ci,), -1.0, device="cuda")
    nws = 2 * Fr * Ci * (1 + (S // 16 + 15) // 16)          # totals + one row of partials per 256-row slice (bf_in_ws_floats covers it for long frames)
    ws = torch.full((nws,), float("nan"), device="cuda")
    src = (None, _p(pred), _p(y), _p(coef), _p(gs)) if fused_loss else (_p(dpred), None, None, None, None)
    rc = lib.bf_debed_last_bwd_norm(1, *src, _p(wc), _p(dpm), _p(ymap), _p(mean), _p(rstd), _p(in_w), _p(in_b), _p(dx), _p(dw), _p(db),
                                    Fr, Ci, Co, h, w, 16, _p(ws), nws, _stream())
    L.check(rc, "debed_last_bwd_norm")
    ref_pm = torch.zeros(P, 16, device="cuda")
    ref_pm[:, :4 * Co] = dpred.view(Fr, Co, h, 2, w, 2).permute(0, 2, 4, 1, 3, 5).reshape(P, 4 * Co)       # n = co*4 + ky*2 + kx
    ref_pm = ref_pm.bfloat16()
    assert torch.equal(dpm, ref_pm)
    yr = ymap.double().view(Fr, S, Ci).requires_grad_(True)
    wr, br = in_w.double().requires_grad_(True), in_b.double().requires_grad_(True)
    xh = (yr - yr.mean(1, keepdim=True)) / (yr.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
    out = torch.nn.functional.gelu(xh * wr + br).view(P, Ci) @ wc.double()
    (out * ref_pm.double()).sum().backward()
    assert torch.isfinite(dx.float()).all()
    assert _rel(dx, yr.grad.view(P, Ci)) < 6e-3          # bf16 output rounding + the polynomial gelu'
    assert _rel(dw - 2.0, wr.grad) < 2e-3 and _rel(db + 1.0, br.grad) < 2e-3
    # declined: fp32, and a workspace that does not hold the slices
    assert lib.bf_debed_last_bwd_norm(0, *src, _p(wc), _p(dpm), _p(ymap), _p(mean), _p(rstd), _p(in_w), _p(in_b), _p(dx), None, None,
                                      Fr, Ci, Co, h, w, 16, _p(ws), nws, _stream()) == 1
    assert lib.bf_debed_last_bwd_norm(1, *src, _p(wc), _p(dpm), _p(ymap), _p(mean), _p(rstd), _p(in_w), _p(in_b), _p(dx), None, None,
                                      Fr, Ci, Co, h, w, 16, _p(ws), 2 * Fr * Ci, _stream()) == 1
    # ... and a refusal says why (a caller that treats code 1 as an error has a text to show)
    assert b"workspace" in lib.bf_last_error()
    with pytest.raises(L.BubbleformerHipError, match="declined"):
        L.check(1, "debed_last_bwd_norm")


@pytest.mark.parametrize("scale,err", [(1.0, 1e-1), (3.0e5, 1e-1), (1.0e-7, 1e-1), (1.0, 1.0e-7), (40.0, 1.0e-6), (1.0e15, 1e-2)])
def test_lploss_sums_cover_the_float_range(scale, err):
    """The fused relative-L2 sums (utils/losses.py:79-89: ||pred - y||_2 / ||y||_2 per (frame, channel), float sums in the reference) on
    un-normalised large-magnitude fields (norm="none" is the dataset default: sums of 1e15 and more), on tiny ones, and with a nearly
    converged numerator (pred - y ~ 1e-7 * y): the integer limbs must neither overflow into NaN nor flush to zero.  Against fp64; the
    result must also be bit-identical run to run (integer adds)."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    Fr, Co, h, w = 3, 4, 48, 40
    g = torch.Generator(device="cuda").manual_seed(5)
    y = (torch.randn(Fr, Co, 2 * h, 2 * w, device="cuda", generator=g) + 0.3) * scale
    pred_ref = y * (1.0 + err * torch.randn(y.shape, device="cuda", generator=g))
    # patch-major prediction rows [P][16]: n = co * 4 + ky * 2 + kx
    pm = pred_ref.view(Fr, Co, h, 2, w, 2).permute(0, 2, 4, 1, 3, 5).reshape(Fr * h * w, Co * 4).contiguous()

    def run():
        pred = torch.empty_like(y)
        lossbuf = torch.zeros(Fr, Co, 2, L.BF_LOSS_LIMBS, device="cuda", dtype=torch.int64)
        loss, coef = torch.zeros(1, device="cuda"), torch.zeros(Fr, Co, device="cuda")
        L.check(lib.bf_pm2nchw(_p(pm), _p(pred), _p(y), _p(lossbuf), Fr, Co, h, w, 16, _stream()), "bf_pm2nchw")
        L.check(lib.bf_lploss_finalize(_p(lossbuf), Fr, Co, _p(loss), _p(coef), _stream()), "bf_lploss_finalize")
        return pred, lossbuf, loss, coef

    pred, lossbuf, loss, coef = run()
    assert torch.equal(pred, pred_ref)
    num = ((pred_ref.double() - y.double()) ** 2).sum((2, 3))
    den = (y.double() ** 2).sum((2, 3))
    v = _loss_limbs_value(lossbuf)
    assert _rel(v[..., 0], num) < 1e-5 and _rel(v[..., 1], den) < 1e-5           # (the kernel squares and pre-sums in fp32)
    ref_loss = (num.sqrt() / den.sqrt()).sum() / Fr
    assert torch.isfinite(loss).all() and abs(float(loss) - float(ref_loss)) < 1e-5 * float(ref_loss)
    ref_coef = 1.0 / (Fr * num.sqrt() * den.sqrt())
    assert torch.isfinite(coef).all() and _rel(coef, ref_coef) < 1e-5
    _, lossbuf2, loss2, coef2 = run()
    assert torch.equal(lossbuf, lossbuf2) and torch.equal(loss, loss2) and torch.equal(coef, coef2)
    # a non-finite prediction poisons the sum instead of wrapping around
    pm_bad = pm.clone(); pm_bad[7, 3] = float("inf")
    lb = torch.zeros(Fr, Co, 2, L.BF_LOSS_LIMBS, device="cuda", dtype=torch.int64)
    L.check(lib.bf_pm2nchw(_p(pm_bad), _p(torch.empty_like(y)), _p(y), _p(lb), Fr, Co, h, w, 16, _stream()), "bf_pm2nchw")
    L.check(lib.bf_lploss_finalize(_p(lb), Fr, Co, _p(loss), _p(coef), _stream()), "bf_lploss_finalize")
    assert not torch.isfinite(loss).all()


@pytest.mark.parametrize("cin,h2,w2", [(4, 96, 96), (3, 40, 32), (4, 33, 16)])
def test_embed_first_stage_one_pass_with_statistics(cin, h2, w2):
    """bf_embed_first: 2x2 patches of the fp32 NCHW clip x the [C0][16] convolution weight (layers/patching.py:30-48, first stage) in one
    kernel, the patch rows bf_im2col_nchw would write, and -- optionally -- the InstanceNorm slice partials of its output, which
    bf_in_stats_merge_slices turns into the same statistics bf_in_stats computes from the stored rows (ragged last slice included)."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    Fr, C0 = 3, 96
    g = torch.Generator(device="cuda").manual_seed(31)
    x = torch.randn(Fr, cin, 2 * h2, 2 * w2, device="cuda", generator=g) + 0.5
    Wc = torch.zeros(C0, 16, device="cuda")
    Wc[:, :4 * cin] = torch.randn(C0, 4 * cin, device="cuda", generator=g) / 2
    Wc = Wc.bfloat16()
    P, S = Fr * h2 * w2, h2 * w2
    patches = torch.full((P, 16), float("nan"), device="cuda", dtype=torch.bfloat16)
    y0 = torch.full((P, C0), float("nan"), device="cuda", dtype=torch.bfloat16)
    nws = lib.bf_in_ws_floats(1, Fr, S, C0)
    ws = torch.zeros(nws, device="cuda")
    part = ws[2 * Fr * C0:]
    assert nws >= 2 * Fr * C0 * (1 + (S + 255) // 256)
    L.check(lib.bf_embed_first(1, _p(x), _p(Wc), _p(patches), _p(y0), Fr, C0, cin, h2, w2, 16, _p(part), _stream()), "embed_first")
    ref_p = torch.zeros(P, 16, device="cuda")
    ref_p[:, :4 * cin] = x.view(Fr, cin, h2, 2, w2, 2).permute(0, 2, 4, 1, 3, 5).reshape(P, 4 * cin)      # k = c*4 + ky*2 + kx
    ref_p = ref_p.bfloat16()
    assert torch.equal(patches, ref_p)
    assert _rel(y0, ref_p.double() @ Wc.double().t()) < 4e-3
    w, b = 1 + 0.1 * torch.randn(C0, device="cuda", generator=g), 0.1 * torch.randn(C0, device="cuda", generator=g)
    got = [torch.empty(Fr, C0, device="cuda") for _ in range(4)]
    L.check(lib.bf_in_stats_merge_slices(1, Fr, S, C0, 256, _p(w), _p(b), None, 1, None, *[_p(t) for t in got], _p(ws), _stream()), "merge")
    ref = [torch.empty(Fr, C0, device="cuda") for _ in range(4)]
    ws2 = torch.zeros(nws, device="cuda")
    L.check(lib.bf_in_stats(1, _p(y0), Fr, S, C0, _p(w), _p(b), None, 1, None, *[_p(t) for t in ref], _p(ws2), _stream()), "in_stats")
    yf = y0.float().view(Fr, S, C0)
    assert _rel(got[0], yf.mean(1)) < 1e-5 and _rel(got[1], (yf.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    for a, r in zip(got, ref):
        assert _rel(a, r) < 1e-5


@pytest.mark.parametrize("Fr,gh,gw", [(3, 2, 16), (2, 12, 32), (7, 48, 48), (3, 24, 24), (2, 4, 40)])
@pytest.mark.parametrize("variant", ["gelu_nk", "plain_kn"])
def test_gather_gemm_2x2_stage(Fr, gh, gw, variant):
    """bf_gather_gemm (gather_gemm.hip): rows of 2x2 / stride-2 patches of a 96-channel map times a [384][96] weight, with GELU(x * sc + sh)
    folded into the operand (the HMLPEmbed stages, weight stored [n][k]) or plain (the HMLPDebed data gradients, weight stored [k][n]),
    against fp64 torch on the same bf16 operands.  One tile in all, runs that cross frames (7 frames x 72 tiles over 512 x 4 waves), 16-row
    blocks in the middle of an image row."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    C0, N = 96, 96
    g = torch.Generator(device="cuda").manual_seed(43)
    fine = torch.randn(Fr, 2 * gh, 2 * gw, C0, device="cuda", generator=g).bfloat16()
    W = (torch.randn(4 * C0, N, device="cuda", generator=g) / 16).bfloat16()              # [k = (2 ky + kx) * C0 + c][n]
    sc, sh = 0.5 + torch.rand(Fr, C0, device="cuda", generator=g), 0.3 * torch.randn(Fr, C0, device="cuda", generator=g)
    out = torch.full((Fr * gh * gw, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    x = fine.double()
    if variant == "gelu_nk":
        x = torch.nn.functional.gelu(x * sc.double()[:, None, None] + sh.double()[:, None, None]).bfloat16().double()      # the operand as the MFMA sees it
        wdev = W.t().contiguous()
        rc = lib.bf_gather_gemm(1, _p(fine), _p(wdev), 0, _p(sc), _p(sh), _p(out), Fr, gh, gw, C0, N, _stream())
    else:
        rc = lib.bf_gather_gemm(1, _p(fine), _p(W), 1, None, None, _p(out), Fr, gh, gw, C0, N, _stream())
    L.check(rc, "gather_gemm")
    A = x.view(Fr, gh, 2, gw, 2, C0).permute(0, 1, 3, 2, 4, 5).reshape(Fr * gh * gw, 4 * C0)
    ref = A @ W.double()
    assert torch.isfinite(out.float()).all()
    assert _rel(out, ref) < 4e-3
    # no misplaced row / 8-column group (the gelu form: polynomial vs exact GELU flips the bf16 rounding of a few operands)
    assert float(((out.double() - ref).abs() / (ref.abs() + 0.05 * ref.abs().mean())).max()) < (0.2 if variant == "gelu_nk" else 0.06)
    assert lib.bf_gather_gemm(0, _p(fine), _p(W), 1, None, None, _p(out), Fr, gh, gw, C0, N, _stream()) == 1          # fp32
    assert lib.bf_gather_gemm(1, _p(fine), _p(W), 1, None, None, _p(out), Fr, 3, 5, C0, N, _stream()) == 1      # rows % 32


@pytest.mark.parametrize("Fr,gh,gw", [(3, 2, 16), (2, 12, 32), (7, 48, 48), (3, 24, 24), (2, 4, 40)])
@pytest.mark.parametrize("pro", [True, False])
def test_scatter_gemm_2x2_stage_with_statistics(Fr, gh, gw, pro):
    """bf_scatter_gemm (gather_gemm.hip): the transposed 2x2 / stride-2 stage at 96 channels -- rows of a coarse map (GELU(x * sc + sh) folded
    in, or plain) times a [4 x 96][96] weight, scattered to the four positions of the fine map -- against fp64 torch on the same bf16
    operands; and the slice partials it leaves, merged by bf_in_stats_merge_slices, against bf_in_stats on the stored map."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    K, C0 = 96, 96
    g = torch.Generator(device="cuda").manual_seed(47)
    P, S4 = Fr * gh * gw, 4 * gh * gw
    a = torch.randn(P, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(4 * C0, K, device="cuda", generator=g) / 8).bfloat16()              # [n = (2 ky + kx) * C0 + c][k]
    sc, sh = 0.5 + torch.rand(Fr, K, device="cuda", generator=g), 0.3 * torch.randn(Fr, K, device="cuda", generator=g)
    fine = torch.full((Fr, 2 * gh, 2 * gw, C0), float("nan"), device="cuda", dtype=torch.bfloat16)
    nsl = gh * gw // 32
    ws = torch.zeros(2 * Fr * C0 * (1 + max(nsl, (S4 + 95) // 96)), device="cuda")
    part = ws[2 * Fr * C0:]
    x = a.double()
    if pro:
        x = torch.nn.functional.gelu(x.view(Fr, gh * gw, K) * sc.double()[:, None] + sh.double()[:, None]).view(P, K).bfloat16().double()
    # the folded form takes the weight as stored by the debed stages ([n][k]), the plain form as stored by the embed stages ([k][n])
    wdev = W if pro else W.t().contiguous()
    rc = lib.bf_scatter_gemm(1, _p(a), _p(wdev), 0 if pro else 1, _p(sc) if pro else None, _p(sh) if pro else None, _p(fine), _p(part), Fr, gh, gw, K, C0, _stream())
    L.check(rc, "scatter_gemm")
    ref = (x @ W.double().t()).view(Fr, gh, gw, 2, 2, C0).permute(0, 1, 3, 2, 4, 5).reshape(Fr, 2 * gh, 2 * gw, C0)
    assert torch.isfinite(fine.float()).all()
    assert _rel(fine, ref) < 4e-3
    assert float(((fine.double() - ref).abs() / (ref.abs() + 0.05 * ref.abs().mean())).max()) < (0.2 if pro else 0.06)
    # statistics of the map AS STORED
    w_, b_ = 1 + 0.1 * torch.randn(C0, device="cuda", generator=g), 0.1 * torch.randn(C0, device="cuda", generator=g)
    yf = fine.float().view(Fr, S4, C0)
    if lib.bf_in_ws_floats(1, Fr, S4, C0) >= 2 * Fr * C0 * (1 + nsl):          # long frames: the sliced workspace holds the 128-pixel slices
        got = [torch.empty(Fr, C0, device="cuda") for _ in range(4)]
        L.check(lib.bf_in_stats_merge_slices(1, Fr, S4, C0, 128, _p(w_), _p(b_), None, 1, None, *[_p(t) for t in got], _p(ws), _stream()), "merge")
        assert _rel(got[0], yf.mean(1)) < 1e-5 and _rel(got[1], (yf.var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-5
    pm = part[:Fr * nsl * C0 * 2].view(Fr, nsl, C0, 2)
    # (a slice is 32 coarse rows x 4 positions: equal counts, so the frame mean is the mean of the slice means)
    tot_mean = pm[..., 0].mean(1)
    assert _rel(tot_mean, yf.mean(1)) < 1e-5
    assert lib.bf_scatter_gemm(0, _p(a), _p(W), 0, None, None, _p(fine), None, Fr, gh, gw, K, C0, _stream()) == 1          # fp32
    assert lib.bf_scatter_gemm(1, _p(a), _p(W), 0, None, None, _p(fine), None, Fr, 3, 5, K, C0, _stream()) == 1      # rows % 32


@pytest.mark.parametrize("Fr,gh,gw", [(3, 2, 16), (2, 12, 32), (7, 48, 48), (3, 24, 24), (2, 4, 40)])
@pytest.mark.parametrize("variant", ["fine_gelu_T", "coarse_gelu", "plain"])
def test_gather_wgrad_2x2_stage(Fr, gh, gw, variant):
    """bf_gather_wgrad (gather_gemm.hip): dW[(q, c)][k] = sum over coarse rows of the gathered fine row times the coarse row, one side
    optionally through GELU(x * sc + sh), against fp64 torch on the same bf16 operands; both output orientations; bit-reproducible."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    C0 = 96
    g = torch.Generator(device="cuda").manual_seed(53)
    P = Fr * gh * gw
    fine = torch.randn(Fr, 2 * gh, 2 * gw, C0, device="cuda", generator=g).bfloat16()
    coarse = torch.randn(P, C0, device="cuda", generator=g).bfloat16()
    sc, sh = 0.5 + torch.rand(Fr, C0, device="cuda", generator=g), 0.3 * torch.randn(Fr, C0, device="cuda", generator=g)
    xf, xc = fine.double(), coarse.double()
    fs = cs = (None, None)
    if variant == "fine_gelu_T":
        xf = torch.nn.functional.gelu(xf * sc.double()[:, None, None] + sh.double()[:, None, None]).bfloat16().double()
        fs = (_p(sc), _p(sh))
    elif variant == "coarse_gelu":
        xc = torch.nn.functional.gelu(xc.view(Fr, gh * gw, C0) * sc.double()[:, None] + sh.double()[:, None]).view(P, C0).bfloat16().double()
        cs = (_p(sc), _p(sh))
    tr = 1 if variant == "fine_gelu_T" else 0
    nws = lib.bf_gather_wgrad_ws_floats(Fr, gh, gw)
    assert nws > 0
    ws = torch.full((nws,), float("nan"), device="cuda")
    out = torch.full((C0, 4 * C0) if tr else (4 * C0, C0), float("nan"), device="cuda")
    L.check(lib.bf_gather_wgrad(1, _p(fine), _p(coarse), *fs, *cs, _p(out), tr, Fr, gh, gw, C0, C0, _p(ws), nws, _stream()), "gather_wgrad")
    A = xf.view(Fr, gh, 2, gw, 2, C0).permute(0, 1, 3, 2, 4, 5).reshape(P, 4 * C0)
    ref = A.t() @ xc
    got = out.t() if tr else out
    assert torch.isfinite(out).all()
    assert _rel(got, ref) < 3e-3
    first = out.clone()
    L.check(lib.bf_gather_wgrad(1, _p(fine), _p(coarse), *fs, *cs, _p(out), tr, Fr, gh, gw, C0, C0, _p(ws), nws, _stream()), "gather_wgrad")
    assert torch.equal(first, out)
    assert lib.bf_gather_wgrad(0, _p(fine), _p(coarse), *fs, *cs, _p(out), tr, Fr, gh, gw, C0, C0, _p(ws), nws, _stream()) == 1            # fp32
    assert lib.bf_gather_wgrad(1, _p(fine), _p(coarse), *fs, *cs, _p(out), tr, Fr, gh, gw, C0, C0, _p(ws), Fr * 36864 - 1, _stream()) == 1        # workspace: not one slab per frame
    L.check(lib.bf_gather_wgrad(1, _p(fine), _p(coarse), *fs, *cs, _p(out), tr, Fr, gh, gw, C0, C0, _p(ws), Fr * 36864, _stream()), "one run per frame")
    assert _rel(out.t() if tr else out, ref) < 3e-3
    assert lib.bf_gather_wgrad(1, _p(fine), _p(coarse), _p(sc), _p(sh), _p(sc), _p(sh), _p(out), tr, Fr, gh, gw, C0, C0, _p(ws), nws, _stream()) == 1


@pytest.mark.parametrize("Fr,gh,gw", [(3, 2, 16), (2, 12, 32), (5, 48, 48), (3, 24, 24)])
def test_stage_kernels_with_the_map_given_as_its_factors(Fr, gh, gw):
    """bf_gather_gemm_rebuilt / bf_gather_wgrad_rebuilt: the fine map is W0 . patch (the first HMLPEmbed stage) and is rebuilt per tile from
    the 16-wide patch rows instead of read -- against the same kernels on the stored map bf16(patches @ W0^T): the rows are rounded the
    same way, so only the order of the fp32 accumulation differs."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    C0, N = 96, 96
    g = torch.Generator(device="cuda").manual_seed(59)
    P = Fr * gh * gw
    patches = (torch.randn(4 * P, 16, device="cuda", generator=g) + 0.2).bfloat16()
    W0 = (torch.randn(C0, 16, device="cuda", generator=g) / 3).bfloat16()
    fine = (patches.float() @ W0.float().t()).bfloat16().view(Fr, 2 * gh, 2 * gw, C0)
    W = (torch.randn(N, 4 * C0, device="cuda", generator=g) / 16).bfloat16()              # [n][k]
    sc, sh = 0.5 + torch.rand(Fr, C0, device="cuda", generator=g), 0.3 * torch.randn(Fr, C0, device="cuda", generator=g)
    a = torch.full((P, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    b = torch.full((P, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    L.check(lib.bf_gather_gemm(1, _p(fine), _p(W), 0, _p(sc), _p(sh), _p(a), Fr, gh, gw, C0, N, _stream()), "gather_gemm")
    L.check(lib.bf_gather_gemm_rebuilt(1, _p(patches), _p(W0), _p(W), 0, _p(sc), _p(sh), _p(b), Fr, gh, gw, C0, N, _stream()), "gather_gemm_rebuilt")
    assert torch.isfinite(b.float()).all() and _rel(b, a) < 2e-3          # bf16 outputs of two fp32 summation orders
    coarse = torch.randn(P, C0, device="cuda", generator=g).bfloat16()
    nws = lib.bf_gather_wgrad_ws_floats(Fr, gh, gw)
    ws = torch.full((nws,), float("nan"), device="cuda")
    o1 = torch.full((C0, 4 * C0), float("nan"), device="cuda")
    o2 = torch.full((C0, 4 * C0), float("nan"), device="cuda")
    L.check(lib.bf_gather_wgrad(1, _p(fine), _p(coarse), _p(sc), _p(sh), None, None, _p(o1), 1, Fr, gh, gw, C0, C0, _p(ws), nws, _stream()), "gather_wgrad")
    L.check(lib.bf_gather_wgrad_rebuilt(1, _p(patches), _p(W0), _p(coarse), _p(sc), _p(sh), _p(o2), 1, Fr, gh, gw, C0, C0, _p(ws), nws, _stream()), "gather_wgrad_rebuilt")
    assert torch.isfinite(o2).all() and _rel(o2, o1) < 1e-5


@pytest.mark.parametrize("Fr,gh1,gw1,C1", [(3, 8, 16, 96), (2, 12, 32, 192), (5, 48, 48, 96)])
def test_embed_backward_tail_one_pass(Fr, gh1, gw1, C1):
    """bf_embed_tail_bwd (embed_tail.hip): the stage-1 data gradient, GELU', the stage-0 InstanceNorm backward and the stage-0 weight
    gradient without the stage-0 gradient map, against autograd through  patches @ W0^T -> InstanceNorm -> GELU -> <. , dact>  in fp32
    (layers/patching.py:24-56); dact is the plain product dy1 @ W1 scattered to the 2x2 positions.  One / two / six tiles per wave,
    one and several runs per frame, 16-row blocks in the middle of an image row (gw1 = 32, 48)."""
    from bubbleformer_amd import _lib as L
    from bubbleformer_amd.ops import _p, _stream
    lib = L.lib()
    C0, Kp = 96, 16
    g = torch.Generator(device="cuda").manual_seed(37)
    S1, S0 = gh1 * gw1, 4 * gh1 * gw1
    patches = (torch.randn(Fr * S0, Kp, device="cuda", generator=g) + 0.3).bfloat16()
    W0 = (torch.randn(C0, Kp, device="cuda", generator=g) / 3).bfloat16()
    W1 = (torch.randn(C1, 4 * C0, device="cuda", generator=g) / 8).bfloat16()
    dy1 = torch.randn(Fr * S1, C1, device="cuda", generator=g).bfloat16()
    in_w, in_b = 1 + 0.2 * torch.randn(C0, device="cuda", generator=g), 0.3 * torch.randn(C0, device="cuda", generator=g)
    y0 = (patches.float() @ W0.float().t()).bfloat16()                       # what bf_embed_first stores
    yf = y0.float().view(Fr, S0, C0)
    mean, rstd = yf.mean(1).contiguous(), (yf.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    sc = (rstd * in_w).contiguous()
    sh = (in_b - mean * sc).contiguous()
    # reference: upstream gradient on the stage-0 grid, then autograd
    dact = (dy1.double() @ W1.double()).view(Fr, gh1, gw1, 2, 2, C0).permute(0, 1, 3, 2, 4, 5).reshape(Fr, S0, C0)      # pixel (2y + ky, 2x + kx)
    W0r = W0.double().requires_grad_(True)
    wr, br = in_w.double().requires_grad_(True), in_b.double().requires_grad_(True)
    y = (patches.double() @ W0r.t()).view(Fr, S0, C0)
    xh = (y - y.mean(1, keepdim=True)) / (y.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
    (torch.nn.functional.gelu(xh * wr + br) * dact).sum().backward()
    nws = lib.bf_embed_tail_ws_floats(Fr, gh1, gw1, C0, Kp)
    assert nws > 0
    ws = torch.full((nws,), float("nan"), device="cuda")
    dwprep = torch.full((C0, Kp), float("nan"), device="cuda")
    dw, db = torch.full((C0,), 2.0, device="cuda"), torch.full((C0,), -1.0, device="cuda")       # accumulated into
    args = [_p(t) for t in (dy1, W1, y0, patches, W0, sc, sh, mean, rstd, in_w, dwprep, dw, db)]
    L.check(lib.bf_embed_tail_bwd(1, *args, Fr, gh1, gw1, C1, C0, Kp, _p(ws), nws, _stream()), "embed_tail_bwd")
    assert torch.isfinite(dwprep).all()
    # bf16 operands of the pixel contraction (dd, 2^-9 each), the polynomial gelu' and the bf16 rounding of the stored y0
    assert _rel(dwprep, W0r.grad) < 1e-2
    assert _rel(dw - 2.0, wr.grad) < 1e-2 and _rel(db + 1.0, br.grad) < 1e-2
    first = dwprep.clone()
    L.check(lib.bf_embed_tail_bwd(1, *args, Fr, gh1, gw1, C1, C0, Kp, _p(ws), nws, _stream()), "embed_tail_bwd")
    assert torch.equal(first, dwprep)                      # fixed summation order: bit-reproducible
    # without the stored stage-0 map: its rows are rebuilt from the patch rows (y0 = W0 . patch, rounded like the stored ones)
    dw2, db2 = torch.full((C0,), 2.0, device="cuda"), torch.full((C0,), -1.0, device="cuda")
    args2 = [_p(t) if t is not None else None for t in (dy1, W1, None, patches, W0, sc, sh, mean, rstd, in_w, dwprep, dw2, db2)]
    dwprep.fill_(float("nan"))
    L.check(lib.bf_embed_tail_bwd(1, *args2, Fr, gh1, gw1, C1, C0, Kp, _p(ws), nws, _stream()), "embed_tail_bwd (rebuilt rows)")
    assert _rel(dwprep, first) < 1e-5 and _rel(dw2 - 2.0, wr.grad) < 1e-2 and _rel(db2 + 1.0, br.grad) < 1e-2
    # declined shapes: nothing launched
    assert lib.bf_embed_tail_bwd(0, *args, Fr, gh1, gw1, C1, C0, Kp, _p(ws), nws, _stream()) == 1        # fp32
    assert lib.bf_embed_tail_bwd(1, *args, Fr, gh1, gw1 + 8, C1, C0, Kp, _p(ws), nws, _stream()) == 1    # gw1 % 16
    assert lib.bf_embed_tail_bwd(1, *args, Fr, gh1, gw1, C1, C0, Kp, _p(ws), nws - 1, _stream()) == 1    # workspace too small
