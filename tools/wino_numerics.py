"""CPU experiment behind DESIGN.md section 3: is the Winograd F(2x2,3x3) form in fp32 as accurate as the
direct fp32 convolution?  Runs 10 residual blocks (20 conv layers, 64 channels, He-scaled random
weights) both ways in fp32 and compares each with the fp64 result.   python tools/wino_numerics.py"""
import torch
import torch.nn.functional as F

Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino(x, w, dt):
    Bt_, G_, At_ = Bt.to(dt), G.to(dt), At.to(dt)
    n, c, h, wd = x.shape
    hp, wp = (h + 1) // 2 * 2, (wd + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + wp - wd, 1, 1 + hp - h))
    U = torch.einsum('ij,ocjk,lk->iloc', G_, w, G_)
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                     # n, c, th, tw, 4, 4
    V = torch.einsum('ij,nctwjk,lk->ilnctw', Bt_, t, Bt_)
    M = torch.einsum('iloc,ilnctw->ilnotw', U, V)
    Y = torch.einsum('pi,ilnotw,ql->notpwq', At_, M, At_)
    n_, o_, th, p, tw, q = Y.shape
    return Y.reshape(n_, o_, th * 2, tw * 2)[:, :, :h, :wd]


def main():
    torch.manual_seed(0)
    nf, layers = 64, 20
    x = torch.rand(1, nf, 66, 80)
    ws = [torch.randn(nf, nf, 3, 3) * (2.0 / (nf * 9)) ** 0.5 for _ in range(layers)]

    def run(conv, dt):
        a = x.to(dt)
        for i in range(0, layers, 2):
            b = torch.relu(conv(a, ws[i].to(dt)))
            a = a + conv(b, ws[i + 1].to(dt))
        return a
    direct = lambda a, w: F.conv2d(a, w, padding=1)   # noqa: E731
    r64 = run(direct, torch.float64)
    r32 = run(direct, torch.float32)
    w32 = run(lambda a, w: wino(a, w, torch.float32), torch.float32)
    print('activation scale: mean |x| %.1f  max %.1f' % (r64.abs().mean(), r64.abs().max()))
    print('direct fp32   vs fp64: max %.3e  mean %.3e' % ((r32 - r64).abs().max(), (r32 - r64).abs().mean()))
    print('Winograd fp32 vs fp64: max %.3e  mean %.3e' % ((w32 - r64).abs().max(), (w32 - r64).abs().mean()))


if __name__ == '__main__':
    main()
