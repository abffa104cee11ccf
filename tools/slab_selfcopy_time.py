import torch, time
nf, g, nz, ny, nx = 5, 2, 512, 512, 512
buf = torch.zeros((nf, nz + 2 * g, ny, nx), dtype=torch.float64, device="cuda")
send_lo, send_hi = buf[:, g:2 * g], buf[:, nz:nz + g]
recv_lo, recv_hi = buf[:, 0:g], buf[:, nz + g:nz + 2 * g]
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def strided():
    recv_lo.copy_(send_hi); recv_hi.copy_(send_lo)
def perfield():
    for f in range(nf):
        recv_lo[f].copy_(send_hi[f]); recv_hi[f].copy_(send_lo[f])
print("strided 2 copies: %.1f us; per-field 10 copies: %.1f us" % (t(strided), t(perfield)))
