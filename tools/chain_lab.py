"""Where does a layer of the chained training body spend its time?  Lab library only
(TECOGAN_HIP_LIB=tools/_lab_libs/libtecogan_lab.so, built with -DTG_LAB=1): workgroup 37 stamps
s_memtime (100 MHz) at 7 points of every layer: 0 top (weights requested) 1 flags seen 2 patch in LDS
3 MFMAs done 4 stores issued 5 stores acknowledged 6 barrier passed (flag store follows)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tecogan_pytorch_amd import _lib as L, ops
hw = int(os.environ.get('HW', '32')); n, nb, nf = 2, 10, 64
nl = 1 + 2 * nb
lib = L.lib()
ws = [torch.randn(64, 51 if i == 0 else 64, 3, 3, device='cuda') * 0.03 for i in range(nl)]
parts = lib.tg_conv3x3_chain_supported(n, hw, hw, 64)
layout = 16 if parts == 4 else 64
pk = [ops.pack_conv3x3_m16(w) if layout == 16 else ops.pack_conv3x3(w, ocb=64)[0] for w in ws]
bs = [torch.zeros(64, device='cuda') for _ in range(nl)]
fw = (L.PackedLayer * nl)()
for i in range(nl):
    fw[i].w, fw[i].b = pk[i].data_ptr(), bs[i].data_ptr()
lr, tran = torch.rand(n, 3, hw, hw, device='cuda'), torch.rand(n, 48, hw, hw, device='cuda')
acts = torch.empty(nl, n, nf, hw, hw, device='cuda')
nfl = lib.tg_conv3x3_chain_flag_ints(24, n, hw, hw)
flags = torch.zeros(nfl + 2 * 24 * 8 + 64, dtype=torch.int32, device='cuda')
err = torch.zeros(16, dtype=torch.int32).pin_memory()
st = torch.cuda.current_stream().cuda_stream
for ep in range(1, 6):
    L.check(lib.tg_srnet_body_fwd(fw, layout, nb, lr.data_ptr(), 3, tran.data_ptr(), 48, acts.data_ptr(), n, nf, hw, hw,
                                  flags.data_ptr(), err.data_ptr(), ep, 1 << 21, st), 'fwd')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for ep in range(6, 26):
    L.check(lib.tg_srnet_body_fwd(fw, layout, nb, lr.data_ptr(), 3, tran.data_ptr(), 48, acts.data_ptr(), n, nf, hw, hw,
                                  flags.data_ptr(), err.data_ptr(), ep, 1 << 21, st), 'fwd')
e1.record(); torch.cuda.synchronize()
print(f'hw={hw} parts={parts}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 21-layer launch, faults {int(err[0])}')
dbg = flags[nfl:nfl + 2 * 24 * 8].view(torch.int64).cpu().view(24, 8)[:nl].double() / 2.3     # ns (the counter runs at the ~2.3 GHz shader clock here)
d = dbg[1:-1]                      # steady-state layers
names = ['wait flags', 'stage patch', 'MFMA', 'reduce+epilogue', 'store ack', 'barrier', 'flag->next top']
seg = [(d[:, k + 1] - d[:, k]).mean().item() for k in range(6)] + [(dbg[2:, 0] - dbg[1:-1, 6]).mean().item()]
print('  '.join(f'{nm} {v / 1e3:.2f}us' for nm, v in zip(names, seg)), ' | layer', (dbg[2:, 0] - dbg[1:-1, 0]).mean().item() / 1e3, 'us')
