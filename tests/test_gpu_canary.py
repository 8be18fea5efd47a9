"""Out-of-bounds guard: every output and the workspace sit between sentinel regions that must come back untouched
(masked vocabularies, unaligned rows, bfloat16, strided views, label tensors wider than the labels, Hessian / HVP)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PAD = 4096  # bytes of sentinel on either side


def _guarded(nbytes, dev):
    buf = torch.full((nbytes + 2 * PAD,), 0xA5, dtype=torch.uint8, device=dev)
    return buf, buf[PAD:PAD + nbytes]


def _intact(buf, nbytes):
    return bool((buf[:PAD] == 0xA5).all().item()) and bool((buf[PAD + nbytes:] == 0xA5).all().item())


@pytest.mark.parametrize("selector", ["any_pipeline", "logits_call"])
@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("B,T,V,U,dtype", [(5, 37, 29, 11, "f32"), (3, 50, 256, 128, "f32"), (4, 41, 300, 40, "f32"),
                                             (2, 30, 1021, 17, "f32"), (3, 26, 64, 200, "f32"), (4, 33, 32, 9, "bf16"),
                                             (300, 13, 8, 3, "f32"), (2, 40, 2048, 12, "f32"),
                                             # long labels (eight per lane, 3-frame blocks), wide vocabulary with a half-full last pass
                                             (2, 620, 256, 300, "f32"), (2, 70, 512, 512, "f32"), (2, 23, 4100, 9, "f32"),
                                             (3, 31, 2560, 20, "f32")])
def test_loss_grad_stays_inside_its_buffers(kind, B, T, V, U, dtype, selector):
    from tf_seq2seq_losses_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(B * 131 + V)
    esz = 4 if dtype == "f32" else 2
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.from_numpy(rng.standard_normal((B, T, V)).astype(np.float32)).to(dev).to(tdt)
    labels = torch.from_numpy(rng.integers(1, V, (B, U)).astype(np.int32)).to(dev)
    ll = torch.from_numpy(rng.integers(0, U + 1, B).astype(np.int32)).to(dev)
    tl = torch.from_numpy(rng.integers(0, T + 1, B).astype(np.int32)).to(dev)
    gbuf, gview = _guarded(B * T * V * esz, dev)
    lbuf, lview = _guarded(B * 4, dev)
    # the conservative workspace and the pipeline's own (checkpoint rows only on the fused tiers: 10x smaller)
    nws = _lib.workspace_bytes(_lib.WS_LOSS_GRAD if selector == "any_pipeline" else _lib.WS_LOSS_GRAD_LOGITS, kind, B, T, V, U)
    wbuf, wview = _guarded(nws, dev)
    dt = _lib.F32 if dtype == "f32" else _lib.BF16
    rc = lib.ctc_amd_loss_grad_ex(kind, 0, x.data_ptr(), dt, T * V, V, labels.data_ptr(), U, ll.data_ptr(), tl.data_ptr(), 0,
                                  B, T, V, U, lview.data_ptr(), gview.data_ptr(), dt, T * V, V, None, wview.data_ptr(), nws,
                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.ctc_amd_last_error()
    torch.cuda.synchronize()
    assert _intact(gbuf, B * T * V * esz), "gradient written out of bounds"
    assert _intact(lbuf, B * 4), "loss written out of bounds"
    assert _intact(wbuf, nws), "workspace overrun"
    g = gview.view(tdt).reshape(B, T, V).float()
    assert torch.isfinite(g).all()


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("B,T,V,U", [(2, 21, 7, 5), (3, 16, 64, 32), (2, 12, 10, 40)])
def test_hessian_and_hvp_stay_inside_their_buffers(kind, B, T, V, U):
    from tf_seq2seq_losses_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7 * B + V)
    x = torch.from_numpy(rng.standard_normal((B, T, V)).astype(np.float32)).to(dev)
    labels = torch.from_numpy(rng.integers(1, V, (B, U)).astype(np.int32)).to(dev)
    ll = torch.from_numpy(rng.integers(0, min(U, T // 2) + 1, B).astype(np.int32)).to(dev)
    tl = torch.from_numpy(rng.integers(T // 2, T + 1, B).astype(np.int32)).to(dev)
    common = (kind, 0, x.data_ptr(), labels.data_ptr(), U, ll.data_ptr(), tl.data_ptr(), 0, B, T, V, U)
    st = torch.cuda.current_stream().cuda_stream
    # dense Hessian
    hb, hv = _guarded(B * (T * V) ** 2 * 4, dev)
    lb, lv = _guarded(B * 4, dev)
    n = _lib.workspace_bytes(_lib.WS_HESSIAN, kind, B, T, V, U)
    wb, wv = _guarded(n, dev)
    assert lib.ctc_amd_hessian(*common, lv.data_ptr(), None, hv.data_ptr(), wv.data_ptr(), n, st) == 0
    torch.cuda.synchronize()
    assert _intact(hb, B * (T * V) ** 2 * 4) and _intact(lb, B * 4) and _intact(wb, n)
    # Hessian-vector product
    vec = torch.randn((B, T, V), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    ob, ov = _guarded(B * T * V * 4, dev)
    n = _lib.workspace_bytes(_lib.WS_HVP, kind, B, T, V, U)
    wb, wv = _guarded(n, dev)
    assert lib.ctc_amd_hvp(*common, vec.data_ptr(), lv.data_ptr(), None, ov.data_ptr(), wv.data_ptr(), n, st) == 0
    torch.cuda.synchronize()
    assert _intact(ob, B * T * V * 4) and _intact(wb, n)
