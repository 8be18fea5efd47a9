"""The C ABI is asynchronous on the caller's stream, allocates nothing and never synchronises, so a call can be captured
in a HIP graph and replayed on new data (launch-bound inner loops: DESIGN.md section 1)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["classic", "simplified"])
@pytest.mark.parametrize("V,U", [(256, 40), (20, 9)])  # fused kernel / three-kernel pipeline
def test_loss_grad_in_a_hip_graph(kind, V, U):
    from tf_seq2seq_losses_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, T = 6, 50
    k = ops.KINDS[kind]
    rng = np.random.default_rng(0)
    x = torch.zeros((B, T, V), device=dev)
    labels = torch.zeros((B, U), dtype=torch.int32, device=dev)
    ll = torch.zeros(B, dtype=torch.int32, device=dev)
    tl = torch.zeros(B, dtype=torch.int32, device=dev)
    loss = torch.zeros(B, device=dev)
    grad = torch.zeros((B, T, V), device=dev)
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_LOSS_GRAD, k, B, T, V, U), dtype=torch.uint8, device=dev)

    def call():
        rc = lib.ctc_amd_loss_grad(k, _lib.WRT_LOGITS, x.data_ptr(), labels.data_ptr(), U, ll.data_ptr(), tl.data_ptr(), 0,
                                   B, T, V, U, loss.data_ptr(), grad.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                   torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.ctc_amd_last_error()

    def fill(seed):
        r = np.random.default_rng(seed)
        h = dict(x=r.standard_normal((B, T, V)).astype(np.float32), labels=r.integers(1, V, (B, U)).astype(np.int32),
                 ll=r.integers(0, U + 1, B).astype(np.int32), tl=r.integers(T // 2, T + 1, B).astype(np.int32))
        x.copy_(torch.from_numpy(h["x"])); labels.copy_(torch.from_numpy(h["labels"]))
        ll.copy_(torch.from_numpy(h["ll"])); tl.copy_(torch.from_numpy(h["tl"]))
        return h

    fill(1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()  # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call()
    for seed in (2, 3):
        h = fill(seed)
        loss.zero_(); grad.zero_()
        g.replay()
        torch.cuda.synchronize()
        rl, rg = C.loss_grad(kind, h["labels"], h["x"], h["ll"], h["tl"], 0)
        fin = np.isfinite(rl)
        assert np.array_equal(np.isfinite(loss.cpu().numpy()), fin)
        assert np.abs(loss.cpu().numpy()[fin] - rl[fin]).max() < 1e-4 * max(1.0, np.abs(rl[fin]).max())
        assert np.abs(grad.cpu().numpy() - rg).max() < 1e-4


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_concurrent_calls_on_two_streams(kind):
    """The library keeps no state: calls with distinct (stream, workspace) pairs may overlap."""
    from tf_seq2seq_losses_amd import _lib, ops
    dev = torch.device("cuda:0")
    k = ops.KINDS[kind]
    B, T, V, U = 40, 300, 256, 60
    rng = np.random.default_rng(5)
    jobs = []
    for i in range(2):
        x = torch.from_numpy(rng.standard_normal((B, T, V)).astype(np.float32)).to(dev)
        labels = torch.from_numpy(rng.integers(1, V, (B, U)).astype(np.int32)).to(dev)
        ll = torch.from_numpy(rng.integers(0, U + 1, B).astype(np.int32)).to(dev)
        tl = torch.from_numpy(rng.integers(T // 2, T + 1, B).astype(np.int32)).to(dev)
        jobs.append(ops.Prepared(labels, x, ll, tl, 0))
    ref = [ops.loss_grad(k, _lib.WRT_LOGITS, p, True) for p in jobs]       # one after the other
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for rep in range(3):                                                      # overlapped
        for i, (p, s) in enumerate(zip(jobs, streams)):
            with torch.cuda.stream(s):
                outs[i] = ops.loss_grad(k, _lib.WRT_LOGITS, p, True)
    torch.cuda.synchronize()
    for (l0, g0), (l1, g1) in zip(ref, outs):
        assert torch.equal(l0, l1) and torch.equal(g0, g1)


@pytest.mark.parametrize("kind", ["classic", "simplified"])
def test_fused_hvp_in_a_hip_graph(kind):
    """ctc_amd_hvp on the fused tier is ONE launch: captured once, replayed on new logits / vectors (one of them sharp enough to
    be redone in the log domain inside that launch), equal to the eager call."""
    from tf_seq2seq_losses_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, T, V, U = 5, 120, 64, 30
    k = ops.KINDS[kind]
    x = torch.zeros((B, T, V), device=dev)
    v = torch.zeros((B, T, V), device=dev)
    labels = torch.zeros((B, U), dtype=torch.int32, device=dev)
    ll = torch.zeros(B, dtype=torch.int32, device=dev)
    tl = torch.zeros(B, dtype=torch.int32, device=dev)
    loss = torch.zeros(B, device=dev)
    out = torch.zeros((B, T, V), device=dev)
    ws = torch.zeros(_lib.workspace_bytes(_lib.WS_HVP, k, B, T, V, U), dtype=torch.uint8, device=dev)

    def call():
        rc = lib.ctc_amd_hvp(k, _lib.WRT_LOGITS, x.data_ptr(), labels.data_ptr(), U, ll.data_ptr(), tl.data_ptr(), 0, B, T, V, U,
                             v.data_ptr(), loss.data_ptr(), None, out.data_ptr(), ws.data_ptr(), ws.numel(),
                             torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.ctc_amd_last_error()

    def fill(seed):
        r = np.random.default_rng(seed)
        scale = np.array([1, 1, 8, 1, 1], np.float32)[:, None, None]   # utterance 2: beyond the linear-domain format
        x.copy_(torch.from_numpy(r.standard_normal((B, T, V)).astype(np.float32) * scale))
        v.copy_(torch.from_numpy(r.standard_normal((B, T, V)).astype(np.float32)))
        labels.copy_(torch.from_numpy(r.integers(1, V, (B, U)).astype(np.int32)))
        ll.copy_(torch.from_numpy(r.integers(U // 2, U + 1, B).astype(np.int32)))
        tl.copy_(torch.from_numpy(r.integers(T // 2, T + 1, B).astype(np.int32)))

    fill(1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call()
    for seed in (2, 3):
        fill(seed)
        loss.zero_(); out.zero_()
        g.replay()
        torch.cuda.synchronize()
        got_l, got_o = loss.clone(), out.clone()
        loss.zero_(); out.zero_()
        call()
        torch.cuda.synchronize()
        assert torch.equal(got_l, loss) and torch.equal(got_o, out)
        assert torch.isfinite(out).all() and out.abs().max().item() > 0
