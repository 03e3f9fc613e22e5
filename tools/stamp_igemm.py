"""Per-phase cycle counts of the igemm K loop from a -DDVG_STAMP diagnostic build (s_memtime stamps; see conv_tile.h):
build every csrc file with -DDVG_STAMP into scratch/stamp/libdvg_stamp.so, then run this on an MI355X."""
import ctypes, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
L = ctypes.CDLL("/root/repo/scratch/stamp/libdvg_stamp.so")
L.dvg_dev_conv_igemm.restype = ctypes.c_int
L.dvg_dev_conv_igemm.argtypes = [ctypes.c_void_p]*2 + [ctypes.c_int] + [ctypes.c_void_p]*4 + [ctypes.c_int64] + [ctypes.c_int]*7 + [ctypes.c_void_p]*2
B, n, R = 256, 128, 8; N = B*R
shapes = [("enc2 fwd", B*64, 64, 128, 3, 9, 0, 0, 0), ("dec0 fwd", N*4, 128, 128, 1, 9, 0, 0, 2), ("enc1 fwd", B*256, 32, 64, 4, 9, 0, 0, 0), ("enc1 dgrad", B*256, 64, 32, 4, 9, 0, 0, 1)]
for name, M, Cin, Cout, Lg, nt, ups, ps, mode in shapes:
    x = torch.randn(M // 4 if ups else M, Cin, device="cuda")
    w = torch.randn(nt * Cin * Cout, device="cuda"); wp = torch.empty_like(w)
    out = torch.empty(M, Cout, device="cuda")
    dbg = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
    for rep in range(3):
        dbg.zero_()
        rc = L.dvg_dev_conv_igemm(x.data_ptr(), w.data_ptr(), mode, wp.data_ptr(), None, out.data_ptr(), dbg.data_ptr(), M, Cin, Cout, Lg, nt, ups, ps, 1, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 5] > 0]
    it = d[0, 5] + 1
    ph = d[:, :5].mean(0) / it
    print(f"{name}: blocks={len(d)} iters={it}  cycles/iter: load-issue {ph[0]:.0f}  mfma {ph[1]:.0f}  barrier1 {ph[2]:.0f}  lds-store(+vmcnt) {ph[3]:.0f}  barrier2 {ph[4]:.0f}  total {ph.sum():.0f}")
