"""GPU-box probe (debug build: make EXTRA=-DPO_PROF): shader cycles of the pose-only BA kernel's phases for frame 0, thread 0."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "orb-slam3-mac_amd", "python"))
import numpy as np, torch, orbhip, synth_ba
ctx = orbhip.Context(0)
M = 1000
pr = synth_ba.make_pose_problem(3, n=M)
n = len(pr["inv_sigma2"])
dx = torch.from_numpy(np.ascontiguousarray(pr["Xw"], np.float64)).cuda(); do = torch.from_numpy(np.ascontiguousarray(pr["obs"], np.float64)).cuda()
dw = torch.from_numpy(np.ascontiguousarray(pr["inv_sigma2"], np.float64)).cuda(); dn = torch.tensor([n], dtype=torch.int32).cuda()
dout = torch.zeros(M, dtype=torch.uint8).cuda(); dni = torch.zeros(1, dtype=torch.int32).cuda()
buf = (C.c_longlong * 8)()
import time
for it in range(3):
    dp = torch.from_numpy(np.ascontiguousarray(pr["pose0"], np.float64).reshape(1, 7).copy()).cuda()
    orbhip.lib.orbhip_debug_po_prof(buf, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orbhip.pose_optimization_device(ctx, dx.data_ptr(), do.data_ptr(), dw.data_ptr(), dn.data_ptr(), 1, n, pr["cam"], dp.data_ptr(), dout.data_ptr(), dni.data_ptr())
    ctx.synchronize(); dt = time.perf_counter() - t0
    orbhip.lib.orbhip_debug_po_prof(buf, 0)
    v = list(buf)
    print("%.3f ms | build walk %d  sum28 %d  solve+oplus %d  trial walk %d  sum1 %d  reclass %d  setup(R) %d | trials %d | total %d" % (dt * 1e3, v[0], v[1], v[2], v[3], v[4], v[5], v[7], v[6], sum(v[:6]) + v[7]))
