import sys, os, torch
sys.path[:0] = ["/root/repo", "/root/repo/uav-wrf-les-ppo-lstm_amd", "/root/repo/tests"]
from uavppo import ops
DEV = torch.device("cuda:0")
H = I = 256
def run(T, N, seed=0):
    torch.manual_seed(seed)
    w_ih = torch.randn(4 * H, I) * 0.2; w_hh = torch.randn(4 * H, H) * 0.2
    b = torch.randn(4 * H) * 0.1
    x = torch.randn(N, T, I); h0 = torch.randn(N, H); c0 = torch.randn(N, H)
    keep = (torch.rand(N, T) > 0.25).float()
    dy, dhn, dcn = torch.randn(N, T, H), torch.randn(N, H), torch.randn(N, H)
    d = lambda t: t.to(DEV).contiguous()
    xg, kg = d(x), d(keep)
    yg, hng, cng, stash = ops.lstm_fwd(xg, kg, d(h0), d(c0), d(w_ih), d(w_hh), d(b), d(b))
    g = ops.lstm_bwd(xg, kg, stash, d(w_ih), d(w_hh), yg, d(h0), dy=d(dy), dhn=d(dhn), dcn=d(dcn), need_dx=True)
    with ops.lstm_arith("f32_mfma"):
        g2 = ops.lstm_bwd(xg, kg, stash, d(w_ih), d(w_hh), yg, d(h0), dy=d(dy), dhn=d(dhn), dcn=d(dcn), need_dx=False)
    e = (g["dgates"] - g2["dgates"]).abs().cpu()
    print(f"T={T} N={N}: dgates err by t:", [round(v, 3) for v in e.amax(dim=(0, 2)).tolist()], "dh0", round((g["dh0"] - g2["dh0"]).abs().max().item(), 3),
          "dc0", round((g["dc0"] - g2["dc0"]).abs().max().item(), 3))
    for t in reversed(range(T)):
        bad = [n for n in range(N) if e[n, t].max() > 1e-3]
        if bad:
            print(f"   t={t}: bad envs {bad}; envs with keep[t+1]=0: {[n for n in range(N) if t + 1 < T and keep[n, t + 1] == 0]}")
            break
    e0 = (g["dh0"] - g2["dh0"]).abs().cpu().amax(dim=1)
    print("   dh0 bad envs", [n for n in range(N) if e0[n] > 1e-3], "keep[0]=0 envs", [n for n in range(N) if keep[n, 0] == 0])
for T, N in ((2, 33), (3, 33), (2, 64), (3, 16), (3, 17), (3, 130)):
    run(T, N)
