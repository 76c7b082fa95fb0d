import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from unet_amd import synthetic as syn
from unet_amd.nested_unet import NestedUNet
m = NestedUNet(3, precision="exact", max_batch=16, max_hw=(512, 512)).to("cuda:0")
m.load_state_dict(syn.make_state_dict(3, 3, True, 2))
x = torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(16, 512, 512, "smooth", 1234))).cuda()
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.segment(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(5): m.segment(x)
print("no events  ms/step", run(20), run(20))
m.profile(True)
print("with events ms/step", run(20)); m.profile_read(); m.profile(True); print("with events", run(20))
m.profile(False)
# hipGraph capture of the forward through torch
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): out = m.segment(x)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = m.segment(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print("graph replay ms/step", (time.perf_counter() - t0) / 20 * 1e3)
