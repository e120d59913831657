for lab in 8 4 16 32 64 108 116 132 208 216 232 308 316 332; do VBNN_CALIB_COPY=$lab python3 - "$lab" <<'PY' 2>&1 | grep -v amdgpu.ids
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from vbnn_amd import _lib as L
from vbnn_amd.nn import Context
ctx = Context.get(torch.device("cuda", 0))
r = []
for _ in range(3):
    info = L.BoxInfo(); L.check(L.lib().vbnn_box_calibrate(ctx.h, C.byref(info))); r.append(round(info.hbm_TBps, 3))
print("lab", sys.argv[1], r)
PY
done
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, time
a = torch.empty(512 << 20, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): b.copy_(a)
e1.record(); torch.cuda.synchronize()
print("torch copy_ TB/s", 2 * (512 << 20) * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
PY
