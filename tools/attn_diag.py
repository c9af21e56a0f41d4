import sys, os, torch, math
sys.path.insert(0, os.getcwd())
import sarssl_boot  # noqa
from sar_ssl_amd import hip
dev = torch.device("cuda:0")
def relerr(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for (B, H, T, dh, adt, p_drop) in [(4, 4, 256, 64, torch.float16, 0.0), (4, 4, 256, 128, torch.float16, 0.0), (4, 4, 256, 64, torch.bfloat16, 0.1), (2, 4, 200, 64, torch.float16, 0.0)]:
    d = H * dh; scale = 1 / math.sqrt(d)
    g = torch.Generator().manual_seed(5)
    mk = lambda *s, sc=1.0, dt_=adt: (torch.randn(s, generator=g) * sc).to(dt_).to(dev)
    qu, qv, k, v, pos = mk(B*T, d), mk(B*T, d), mk(B*T, d), mk(B*T, d), mk(T, d, sc=2.0)
    dctx = mk(B*T, d, dt_=torch.bfloat16)
    ctx, aux, bias = hip.relpos_attn_fwd_pos(qu, qv, k, v, pos, B, H, T, dh, scale, p_drop, 7)
    # unfused backward
    dqu, dk, dv = (torch.empty((B*T, d), dtype=torch.bfloat16, device=dev) for _ in range(3))
    dbias = hip.relpos_attn_bwd(qu, k, v, bias, aux, dctx, dqu, dk, dv, B, H, T, dh, scale, p_drop, 7)
    dps = hip.relshift_bwd(dbias)
    dqv_u = torch.empty((B*T, d), dtype=torch.bfloat16, device=dev)
    hip.gemm(dps, pos, a_kc=True, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=B*H, batch_inner=H, sA=(H*T*T, T*T), sB=(0, dh), out=dqv_u, ldc=d, sC=(T*d, dh))
    dposb = torch.empty((B, T, d), dtype=torch.bfloat16, device=dev)
    hip.gemm(dps, qv, a_kc=False, b_kc=False, M=T, N=dh, K=T, lda=T, ldb=d, nbatch=B*H, batch_inner=H, sA=(H*T*T, T*T), sB=(T*d, dh), out=dposb, ldc=d, sC=(T*d, dh))
    dpos_u = hip.colsum_store(dposb.view(B, T*d)).view(T, d)
    # fused
    dqu2, dk2, dv2, dqv_f = (torch.empty((B*T, d), dtype=torch.bfloat16, device=dev) for _ in range(4))
    part = hip.relpos_attn_bwd_pos(qu, qv, k, v, pos, bias, aux, dctx, dqu2, dqv_f, dk2, dv2, B, H, T, dh, scale, p_drop, 7)
    dpos_f = hip.colsum_store(part.view(part.shape[0], T*d)).view(T, d)
    # f64 from the exact dps (= unshifted d(bias) the unfused kernels produced, bf16) -> isolates the two products
    dps64 = dps.double()
    want_dqv = torch.einsum("bhrm,mhc->brhc", dps64, pos.view(T, H, dh).double()).reshape(B*T, d)
    want_dpos = torch.einsum("bhrm,brhc->mhc", dps64, qv.view(B, T, H, dh).double()).reshape(T, d)
    print((B, H, T, dh, str(adt), p_drop), "dqv: unfused %.2e fused %.2e | dpos: unfused %.2e fused %.2e | fused-vs-unfused dqv %.2e dpos %.2e | dqu equal %s"
          % (relerr(dqv_u, want_dqv), relerr(dqv_f, want_dqv), relerr(dpos_u, want_dpos), relerr(dpos_f, want_dpos), relerr(dqv_f, dqv_u), relerr(dpos_f, dpos_u), torch.equal(dqu, dqu2)))
    rows = (dqv_f.double() - want_dqv).view(B, T, d).norm(dim=-1) / want_dqv.view(B, T, d).norm(dim=-1)
    print("   worst dqv rows:", [(int(i) % T, "%.1e" % rows.view(-1)[i]) for i in rows.view(-1).topk(4).indices])
    cols = (dpos_f.double() - want_dpos).norm(dim=-1) / want_dpos.norm(dim=-1)
    print("   worst dpos positions:", [(int(i), "%.1e" % cols[i]) for i in cols.topk(4).indices])
