import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as g
pkg = g.load_package()
import bench
lf = pkg.LensFlare(0)
lf.set_frame(64, 48)
ptr, nbytes = lf.device_buffer(pkg.SAMPLE_BUFFER)
t = torch.as_tensor(bench.DevView(ptr, nbytes // 8), device="cuda:0")
print("tensor", t.shape, t.dtype, t.device, t.data_ptr() == ptr)
t[:10] = 3.0
torch.cuda.synchronize()
print(lf.read_tile(0, 0, 0, 4, 1).reshape(-1)[:10])
