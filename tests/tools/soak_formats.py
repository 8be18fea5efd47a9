"""Soak run of the producer formats (test infrastructure): random shapes, each in a random storage format -- float32 / bfloat16 /
float16, batch-major or time-major storage, a view whose base pointer is not 16-byte aligned, a packed ragged batch with gaps and a
row stride wider than V -- through ctc_amd_loss_grad_ex / ctc_amd_loss_grad_packed, compared with the plain contiguous float32 call
on the SAME (rounded) values (loss to 1e-5 relative, gradient to one rounding of the output type plus the 6e-5 two tiers may differ by) and, on the first utterances,
with the float64 C oracle.  usage: python tests/tools/soak_formats.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import c_oracle as C
from tf_seq2seq_losses_amd import _lib, ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2468)
dev = torch.device("cuda:0")
RT = {torch.float32: 2e-5, torch.bfloat16: 1.0 / 128, torch.float16: 1.0 / 1024}
t0 = time.time(); n = 0; last = t0; worst = {}; count = {}
while time.time() - t0 < budget:
    B = int(rng.integers(1, 40)); T = int(rng.integers(1, 200)); V = int(rng.choice([3, 8, 29, 64, 100, 256, 300, 512, 1000, 1028, 2048]))
    U = int(rng.choice([1, 7, 40, 64, 100, 128, 200, 256, 300]))
    kind = int(rng.integers(0, 2))
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    layout = ["batch-major", "time-major", "offset-view", "packed"][int(rng.integers(0, 4))]
    x32 = torch.from_numpy(rng.standard_normal((B, T, V)).astype(np.float32) * float(rng.choice([0.3, 1.0, 3.0]))).to(dev)
    xq = x32.to(dtype)                      # the values every variant sees
    xref = xq.to(torch.float32).contiguous()
    labels = torch.from_numpy(rng.integers(1, max(V, 2), (B, U)).astype(np.int32) % V).to(dev)
    ll = torch.from_numpy(rng.integers(0, U + 1, B).astype(np.int32)).to(dev)
    tl = torch.from_numpy(rng.integers(0, T + 1, B).astype(np.int32)).to(dev)
    d_loss = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(dev) if rng.integers(0, 2) else None
    pref = ops.Prepared(labels, xref, ll, tl, 0, U=U)
    loss_ref, grad_ref = ops.loss_grad(kind, _lib.WRT_LOGITS, pref, True, d_loss=d_loss)
    tag = f"{str(dtype).split('.')[-1]} {layout}"
    if layout == "packed":
        tln = tl.cpu().numpy()
        off = np.zeros(B, np.int64); total = 0
        for b in range(B):
            off[b] = total + int(rng.integers(0, 3)); total = int(off[b]) + int(tln[b])
        stride = V + int(rng.choice([0, 4, 5]))
        store = torch.full((total + 2, stride), 3.0, dtype=dtype, device=dev)
        packed = store[:, :V]
        for b in range(B):
            packed[off[b]:off[b] + tln[b]] = xq[b, :tln[b]]
        loss, gp = ops.loss_grad_packed(kind, _lib.WRT_LOGITS, labels, packed, torch.from_numpy(off).to(dev), ll, tl, 0, T, U=U, d_loss=d_loss)
        grad = torch.zeros((B, T, V), dtype=dtype, device=dev)
        for b in range(B):
            grad[b, :tln[b]] = gp[off[b]:off[b] + tln[b]]
    else:
        if layout == "time-major":
            xv = xq.transpose(0, 1).contiguous().transpose(0, 1)
        elif layout == "offset-view":  # base pointer 4 (2) bytes past a 16-byte boundary, rows padded
            buf = torch.zeros(B * T * (V + 3) + 8, dtype=dtype, device=dev)
            xv = buf[1:1 + B * T * (V + 3)].view(B, T, V + 3)[:, :, :V]
            xv.copy_(xq)
        else:
            xv = xq
        p = ops.Prepared(labels, xv, ll, tl, 0, U=U, keep_format=True)
        loss, grad = ops.loss_grad(kind, _lib.WRT_LOGITS, p, True, d_loss=d_loss)
        assert grad.dtype == dtype and (not p.native or grad.stride() == xv.stride()), (tag, B, T, V, U)
    ctx = (tag, B, T, V, U, kind)
    fin = torch.isfinite(loss_ref)
    assert torch.equal(torch.isfinite(loss), fin), ctx
    assert torch.allclose(loss[fin], loss_ref[fin], rtol=1e-5, atol=1e-5), (ctx, (loss[fin] - loss_ref[fin]).abs().max().item())
    scale = 1.0 if d_loss is None else max(1.0, float(d_loss.abs().max()))
    err = ((grad.float() - grad_ref).abs() - RT[dtype] * grad_ref.abs()).max().item() if grad.numel() else 0.0
    if not err < 6e-5 * scale:  # which of the two is off?  (float64 oracle on the whole batch)
        rl_, rg_ = C.loss_grad("classic" if kind == 0 else "simplified", labels.cpu().numpy(), xref.cpu().numpy(), ll.cpu().numpy(), tl.cpu().numpy(), 0)
        if d_loss is not None:
            rg_ = rg_ * d_loss.cpu().numpy()[:, None, None]
        ea = np.abs(grad.float().cpu().numpy() - rg_).max(axis=(1, 2)); eb = np.abs(grad_ref.cpu().numpy() - rg_).max(axis=(1, 2))
        os.makedirs("gpurun_out", exist_ok=True)
        for bb in np.nonzero((ea > 1e-4 * scale) | (eb > 1e-4 * scale))[0][:4]:  # the offending utterances alone (tests/tools/replay_case.py)
            np.savez(os.path.join("gpurun_out", f"soak_formats_fail_{n}_b{bb}.npz"), x=xref[bb:bb + 1].cpu().numpy(), labels=labels[bb:bb + 1].cpu().numpy(),
                     ll=ll[bb:bb + 1].cpu().numpy(), tl=tl[bb:bb + 1].cpu().numpy(), kind=kind, U=U, why=str(ctx))
        print("FAILED", ctx, "format call vs float64 per utterance:", np.round(ea, 6).tolist(), "\nplain call vs float64:", np.round(eb, 6).tolist(),
              "\nll", ll.cpu().numpy().tolist(), "tl", tl.cpu().numpy().tolist(), "pipeline of the plain call:", ops.pipeline_of(kind, 0, pref), flush=True)
    assert err < 6e-5 * scale, (ctx, err)  # (the two calls may run different tiers: each within 1e-4 of the float64 result, typically 1e-5 apart)
    m = min(B, 3)
    rl, rg = C.loss_grad("classic" if kind == 0 else "simplified", labels[:m].cpu().numpy(), xref[:m].cpu().numpy(), ll[:m].cpu().numpy(), tl[:m].cpu().numpy(), 0)
    if d_loss is not None:
        rg = rg * d_loss[:m].cpu().numpy()[:, None, None]
    e64 = float(np.abs(grad[:m].float().cpu().numpy() - rg).max()) if rg.size else 0.0
    assert e64 < (1e-4 if dtype == torch.float32 else 1e-2 if dtype == torch.bfloat16 else 2e-3) * scale, (ctx, e64)
    worst[tag] = max(worst.get(tag, 0.0), e64 / scale); count[tag] = count.get(tag, 0) + 1
    n += 1
    if time.time() - last > 10:
        print(f"{n} cases", flush=True); last = time.time()
print(f"formats soak ok: {n} random cases in {time.time() - t0:.0f} s; worst gradient error vs float64 per format (cases): " +
      ", ".join(f"{k} {worst[k]:.1e} ({count[k]})" for k in sorted(worst)))
