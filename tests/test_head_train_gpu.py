"""Lifting head training step on the MI355X (src/train.py:137-176): ``train.TrainableHead`` through the C ABI
against two optimizer steps of the REFERENCE module (tests/golden/train_head_golden.pt) and against the autograd oracle."""
import pytest
import torch

from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib_built():
    from implementation_phd_lab_vision_amd import _lib
    _lib.build_library()
    return _lib.load_library()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _head(c_or_dims, seed, precision="fp16"):
    from implementation_phd_lab_vision_amd import train
    from oracle import lifting_oracle as lo
    d, nb = c_or_dims
    sd = lo.synthetic_head_state_dict(d, nb, seed)
    m = train.TrainableHead(d, 17, nb, precision=precision)
    m.load_state_dict(sd); m.to(DEV)
    return m, sd


# ---------------------------------------------------------------- kernels ----------------------------------------------------
def test_transpose16_pads_and_transposes(lib_built):
    from implementation_phd_lab_vision_amd import _lib
    for rows, cols in ((15, 64), (130, 192), (64, 51), (1, 1)):
        x = torch.randn(rows, cols).half().to(DEV)
        ld = (rows + 63) // 64 * 64
        out = torch.full((cols, ld), 7.0, dtype=torch.float16, device=DEV)
        _lib.check(lib_built.r50_op_transpose16(x.data_ptr(), rows, cols, out.data_ptr(), ld, _stream()), None, "t")
        assert torch.equal(out[:, :rows], x.t())
        assert bool((out[:, rows:] == 7.0).all())                 # padding columns are not written


def test_colsum_relu_bwd_mask_scale_grad_accum(lib_built):
    from implementation_phd_lab_vision_amd import _lib
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 128, generator=g).half().to(DEV)
    out = torch.ones(100, device=DEV)
    _lib.check(lib_built.r50_op_colsum(x.data_ptr(), 37, 100, 128, 0.5, out.data_ptr(), 1, 1, _stream()), None, "colsum")
    torch.testing.assert_close(out.cpu(), 1 + 0.5 * x.float()[:, :100].sum(0).cpu(), rtol=1e-5, atol=1e-5)
    xf = torch.randn(5, 70, generator=g).to(DEV)
    out = torch.empty(70, device=DEV)
    _lib.check(lib_built.r50_op_colsum_f32(xf.data_ptr(), 5, 70, 2.0, out.data_ptr(), 0, _stream()), None, "colsum32")
    torch.testing.assert_close(out, 2 * xf.sum(0), rtol=1e-6, atol=1e-6)
    act = torch.randn(1000, generator=g).half().to(DEV)
    dy = torch.randn(1000, generator=g).half().to(DEV)
    want = (dy.float() * 2.0 * (act.float() > 0)).half()
    _lib.check(lib_built.r50_op_relu_bwd(dy.data_ptr(), act.data_ptr(), 2.0, 1000, 1, _stream()), None, "relu_bwd")
    assert torch.equal(dy, want)
    h = torch.randn(1000, generator=g).half().to(DEV)
    mask = (torch.rand(1000, generator=g) > 0.5).to(torch.uint8).to(DEV)
    want = (h.float() * 2.0 * mask).half()
    _lib.check(lib_built.r50_op_mask_scale(h.data_ptr(), mask.data_ptr(), 2.0, 1000, 1, _stream()), None, "mask_scale")
    assert torch.equal(h, want)
    dst = torch.ones(1000, device=DEV)
    _lib.check(lib_built.r50_op_grad_accum(want.data_ptr(), 0.25, dst.data_ptr(), 1000, 1, 1, _stream()), None, "grad_accum")
    torch.testing.assert_close(dst, 1 + 0.25 * want.float())


def test_mse_loss_grad_and_check_finite(lib_built):
    from implementation_phd_lab_vision_amd import _lib
    g = torch.Generator().manual_seed(4)
    y, gt = torch.randn(6, 17, 3, generator=g).to(DEV), torch.randn(6, 17, 3, generator=g).to(DEV)
    dy, loss2 = torch.empty_like(y), torch.empty(2, device=DEV)
    _lib.check(lib_built.r50_op_mse_loss_grad(y.data_ptr(), gt.data_ptr(), y.numel(), 8.0, dy.data_ptr(), loss2.data_ptr(), _stream()), None, "mse")
    torch.testing.assert_close(loss2[0], (y - gt).pow(2).mean(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss2[1], torch.norm(y - gt, dim=-1).mean(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dy, 8.0 * 2 * (y - gt) / y.numel(), rtol=1e-5, atol=1e-7)
    found = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib_built.r50_op_check_finite(dy.data_ptr(), dy.numel(), found.data_ptr(), _stream()), None, "finite")
    assert int(found) == 0
    dy.view(-1)[100] = float("inf")
    _lib.check(lib_built.r50_op_check_finite(dy.data_ptr(), dy.numel(), found.data_ptr(), _stream()), None, "finite")
    assert int(found) == 1
    dy.view(-1)[100] = float("nan"); found.zero_()
    _lib.check(lib_built.r50_op_check_finite(dy.data_ptr(), dy.numel(), found.data_ptr(), _stream()), None, "finite")
    assert int(found) == 1


def test_gn_relu_causal3_bwd_against_autograd(lib_built):
    import torch.nn.functional as F
    from implementation_phd_lab_vision_amd import _lib
    b, t, c = 3, 6, 128
    g = torch.Generator().manual_seed(6)
    x = torch.randn(b, t, c, generator=g).half()
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    dr = torch.randn(b * t, 3 * c, generator=g).half()
    add = torch.randn(b * t, c, generator=g).half()
    xr, gr, br = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = F.relu(F.group_norm(xr.permute(0, 2, 1), 32, gr, br, 1e-5)).permute(0, 2, 1)
    idx = (torch.arange(t).view(t, 1) + torch.arange(-2, 1).view(1, 3)).clamp_min(0)
    rows = y[:, idx, :].reshape(b * t, 3 * c)
    rows.backward(dr.double())
    dx = torch.empty(b * t, c, dtype=torch.float16, device=DEV)
    part = torch.empty(2, b, c, device=DEV)
    args = [v.to(DEV) for v in (dr, x, gamma, beta, add)]
    _lib.check(lib_built.r50_op_gn_relu_causal3_bwd(args[0].data_ptr(), args[1].data_ptr(), b, t, c, 32, args[2].data_ptr(), args[3].data_ptr(),
                                                    1e-5, args[4].data_ptr(), dx.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), 1, _stream()),
               None, "gn_bwd")
    assert _rel(dx.cpu().float(), xr.grad.reshape(b * t, c) + add.double()) < 2e-3
    assert _rel(part[0].sum(0).cpu(), gr.grad) < 1e-4
    assert _rel(part[1].sum(0).cpu(), br.grad) < 1e-4


def test_adamw_kernel_equals_torch_adamw(lib_built):
    from implementation_phd_lab_vision_amd import _lib
    g = torch.Generator().manual_seed(8)
    n = 5000
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-2)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p16 = torch.empty(n, dtype=torch.float16, device=DEV)
    found = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        ref.grad = grad.clone(); opt.step()
        gd = grad.to(DEV)
        _lib.check(lib_built.r50_op_adamw(p.data_ptr(), m.data_ptr(), v.data_ptr(), gd.data_ptr(), p16.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8,
                                          1e-2, step, found.data_ptr(), 1, _stream()), None, "adamw")
        torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        assert torch.equal(p16, p.half())
    found.fill_(1)
    before = p.clone()
    _lib.check(lib_built.r50_op_adamw(p.data_ptr(), m.data_ptr(), v.data_ptr(), gd.data_ptr(), p16.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 4,
                                      found.data_ptr(), 1, _stream()), None, "adamw")
    assert torch.equal(p, before)                                  # skipped on the device when the inf flag is up


# ---------------------------------------------------------------- the step ---------------------------------------------------
GRAD_TOL = {"fp16": 1.5e-2, "bf16": 8e-2}


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_train_steps_equal_reference_module(lib_built, precision):
    from implementation_phd_lab_vision_amd import train
    from tests.golden.make_golden_train_head import batches_for
    tol = GRAD_TOL[precision]
    for c in torch.load(GOLDEN / "train_head_golden.pt", map_location="cpu", weights_only=True):
        m, sd = _head((c["latent_dim"], c["number_blocks"]), c["seed"], precision)
        m.eval()                                                   # the fixture's steps ran with dropout = identity
        optim = train.AdamW(m, lr=c["lr"], weight_decay=1e-2)
        scaler = train.GradScaler(init_scale=1024.0)
        for s, (feats, gt) in enumerate(batches_for(c["seed"], c["b"], c["t"])):
            loss, mpjpe, skipped = m.train_step(feats.to(DEV), gt.to(DEV), optim, scaler)
            assert not skipped
            assert loss == pytest.approx(c["losses"][s], rel=3 * tol), (s, loss)
            if s == 0:
                grads = m.named_gradients()
                assert sorted(grads) == sorted(c["grads"])
                for k, g in c["grads"].items():
                    assert float(grads[k].norm()) == pytest.approx(g["norm"], rel=tol), k
                    assert _rel(grads[k].reshape(-1)[:64], g["head"]) < 2 * tol, (k, _rel(grads[k].reshape(-1)[:64], g["head"]))
        final = m.state_dict()
        for k, p in c["params"].items():
            delta_want = p["head"] - sd[k].reshape(-1)[:64]
            delta_got = final[k].reshape(-1)[:64] - sd[k].reshape(-1)[:64]
            assert float(delta_want.abs().max()) > 0
            # two Adam steps are ~ -lr * (sign-like) updates: elements whose two gradients nearly cancel are ill-conditioned, so the
            # bulk is held tightly (median) and the whole slice loosely (norm); every element moved by at most ~2 lr either way
            err = (delta_got - delta_want).abs()
            loose = 1.0 if precision == "fp16" else 3.0
            assert float(err.median()) < 0.05 * loose * c["lr"], (k, float(err.median()))
            assert _rel(delta_got, delta_want) < 0.3 * loose and float(err.max()) < 2.5 * loose * c["lr"], (k, _rel(delta_got, delta_want))
        for k in sd:
            if k.startswith("f_AR.") or k == "f_3D.y0":
                assert torch.equal(final[k], sd[k])
        assert optim.step_count == 2


def test_train_step_with_dropout_masks_against_oracle(lib_built):
    """train.py's configuration (D=1024, 2 blocks), B x T = 4 x 40, explicit keep-masks shared with the oracle."""
    from implementation_phd_lab_vision_amd import train
    from oracle import lifting_oracle as lo
    m, sd = _head((1024, 2), 21)
    m.train()
    g = torch.Generator().manual_seed(210)
    feats = torch.randn(4, 40, 2048, generator=g).abs()
    gt = torch.randn(4, 40, 17, 3, generator=g) * 0.5
    gen = torch.Generator(device=DEV).manual_seed(5)
    masks = m.make_dropout_masks(4, 40, gen)
    frac = float(torch.cat([v.float().view(-1) for v in masks.values()]).mean())
    assert 0.49 < frac < 0.51
    pred, loss2 = m.forward_backward(feats.to(DEV), gt.to(DEV), loss_scale=256.0, masks=masks)
    losses, mpjpes, grads, _ = lo.train_steps_reference(sd, [(feats, gt)], [{k: v.cpu() for k, v in masks.items()}], dtype=torch.float64)
    assert float(loss2[0]) == pytest.approx(losses[0], rel=2e-2)
    assert float(loss2[1]) == pytest.approx(mpjpes[0], rel=2e-2)
    got = m.named_gradients()
    for k, gr in grads.items():
        assert _rel(got[k], gr) < 3e-2, (k, _rel(got[k], gr))


def test_overflow_skips_the_step_and_halves_the_scale(lib_built):
    from implementation_phd_lab_vision_amd import train
    m, sd = _head((64, 2), 31)
    m.eval()
    optim, scaler = train.AdamW(m, lr=1e-4), train.GradScaler(init_scale=2.0 ** 40)       # far beyond fp16's range
    g = torch.Generator().manual_seed(310)
    feats, gt = torch.randn(2, 5, 2048, generator=g).abs().to(DEV), torch.randn(2, 5, 17, 3, generator=g).to(DEV)
    before = m.flat_master.clone()
    _, _, skipped = m.train_step(feats, gt, optim, scaler)
    assert skipped and scaler.get_scale() == 2.0 ** 39 and optim.step_count == 0
    assert torch.equal(m.flat_master, before)
    scaler = train.GradScaler(init_scale=256.0)
    _, _, skipped = m.train_step(feats, gt, optim, scaler)
    assert not skipped and optim.step_count == 1 and not torch.equal(m.flat_master, before)
    # the eval forward reads the refreshed 16-bit weights
    assert torch.equal(m.flat_w16, m.flat_master.half())


def test_loss_decreases_over_steps(lib_built):
    from implementation_phd_lab_vision_amd import train
    m, _ = _head((128, 2), 41)
    m.train()
    optim, scaler = train.AdamW(m, lr=1e-3), train.GradScaler(init_scale=1024.0)
    g = torch.Generator().manual_seed(410)
    feats, gt = torch.randn(8, 10, 2048, generator=g).abs().to(DEV), (torch.randn(8, 10, 17, 3, generator=g) * 0.5).to(DEV)
    first = last = None
    for _ in range(30):
        loss, _, _ = m.train_step(feats, gt, optim, scaler)
        first = loss if first is None else first
        last = loss
    assert last < 0.5 * first, (first, last)


def test_graphed_step_equals_eager_step(lib_built):
    """forward + loss + backward replayed as one HIP graph: same gradients and parameters as the eager launches, bit for bit."""
    from implementation_phd_lab_vision_amd import train
    g = torch.Generator().manual_seed(510)
    batches = [(torch.randn(4, 10, 2048, generator=g).abs().to(DEV), (torch.randn(4, 10, 17, 3, generator=g) * 0.5).to(DEV)) for _ in range(3)]
    finals = []
    for graphed in (False, True):
        m, _ = _head((128, 2), 51)
        m.eval().enable_graphs(graphed)
        optim, scaler = train.AdamW(m, lr=1e-3), train.GradScaler(init_scale=512.0)
        losses = [m.train_step(f, y, optim, scaler)[0] for f, y in batches]
        finals.append((losses, m.flat_grad.clone(), m.flat_master.clone()))
        if graphed:
            assert len(m._graphs) == 1
    assert finals[0][0] == finals[1][0]
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])
    # train mode under a graph: fresh dropout masks on every replay
    m, _ = _head((128, 2), 51)
    m.train().enable_graphs(True)
    optim, scaler = train.AdamW(m, lr=1e-3), train.GradScaler(init_scale=512.0)
    f, y = batches[0]
    g1 = (m.train_step(f, y, optim, scaler), m.flat_grad.clone())[1]
    g2 = (m.train_step(f, y, optim, scaler), m.flat_grad.clone())[1]
    assert not torch.equal(g1, g2) and bool(torch.isfinite(g2).all())


def test_data_parallel_overflow_on_one_rank_skips_on_every_rank(lib_built, tmp_path):
    """World size 2 (both ranks on this GPU, gloo collectives): only rank 1's gradients overflow fp16.  The found-overflow flag is
    MAX-reduced before the decision, so both ranks skip the step together and stay bit-identical replicas."""
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "dp"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(root / "tests" / "dist_train_gpu_worker.py"), str(out)], env=env, cwd=str(root),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, f"rank {rank} failed:\n{o.decode(errors='replace')[-3000:]}"
    r0, r1 = (torch.load(f"{out}.rank{r}", weights_only=True) for r in range(2))
    assert r0["local_found"] == 0 and r1["local_found"] == 1          # the scenario: exactly one rank overflows locally
    assert r0["skipped"] == r1["skipped"] and r0["skipped"][0] is True
    assert r0["scales"] == r1["scales"] and r0["scales"][0] == 2.0 ** 11
    assert r0["steps"] == r1["steps"]
    assert torch.equal(r0["flat_master"], r1["flat_master"])
