#!/usr/bin/env python3
"""GPU-box probe: sustained shader clock and board power while one kernel runs back to back for a few seconds.

Samples the amdgpu hwmon files (power1_average / power1_input, freq1_input) from a thread while the main thread keeps the
queue full; prints the median over the steady part.  WORK = conv shape name (scripts/bench_conv_shapes.py), "gemm_bf16"
(torch / hipBLASLt 8192^3) or "idle"."""
import glob, os, sys, time, threading, math, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r2dm_amd import _lib
from bench_conv_shapes import SHAPES

def hwmon():
    """hwmon directory of the card torch's device 0 sits on (matched by PCI address)."""
    pr = torch.cuda.get_device_properties(0)
    want = None
    if hasattr(pr, "pci_bus_id"):
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if not os.path.exists(d + "/freq1_input"): continue
        slot = ""
        try:
            for ln in open(os.path.dirname(os.path.dirname(d)) + "/uevent"):
                if ln.startswith("PCI_SLOT_NAME="): slot = ln.strip().split("=")[1]
        except Exception: pass
        cands.append((d, slot))
    for d, slot in cands:
        if want and slot.startswith(want): return d
    print("no PCI match for", want, "among", cands, file=sys.stderr)
    return cands[0][0] if cands else None

H = hwmon()
def rd(name):
    try:
        with open(f"{H}/{name}") as f: return int(f.read())
    except Exception: return None

samples, stop = [], False
def sampler():
    while not stop:
        samples.append((time.time(), rd("freq1_input"), rd("power1_average") or rd("power1_input")))
        time.sleep(0.02)

work = os.environ.get("WORK", "L1_64_64"); secs = float(os.environ.get("SECS", "3")); B = int(os.environ.get("B", "8"))
dev = "cuda"; L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
if work == "gemm_bf16":
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    call = lambda: torch.matmul(a, b); flop = 2 * 8192 ** 3
elif work == "idle":
    call = lambda: time.sleep(0.001); flop = 0
else:
    cin, cout, h, w, k, pro, res = SHAPES[work]
    x = torch.randn(B, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, device=dev); aff = torch.rand(B, cin, 2, device=dev) + 0.5 if pro else None
    r = torch.randn(B, cout, h, w, device=dev) if res else None; sc = torch.tensor([0.7071], device=dev) if res else None
    packed = torch.empty(L.r2dm_conv_packed_elems(cout, cin, k, B, h, w), device=dev); y = torch.empty(B, cout, h, w, device=dev)
    call = lambda: _lib.check(L.r2dm_conv2d_ring(x.data_ptr(), wt.data_ptr(), bias.data_ptr(), packed.data_ptr(), _lib.ptr(aff), pro, _lib.ptr(r), _lib.ptr(sc), y.data_ptr(), B, cin, cout, h, w, k, st))
    flop = 2 * B * cout * cin * k * k * h * w
if os.environ.get("PIECES"):
    import ctypes
    L.r2dm_set_conv_pieces.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    _lib.check(L.r2dm_set_conv_pieces(None, int(os.environ["PIECES"])))
for _ in range(5): call()
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(20): call()
    n += 20
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1)
steady = [s for s in samples if s[0] - t0 > secs * 0.4]
f = [s[1] for s in steady if s[1]]; p = [s[2] for s in steady if s[2]]
print(f"{work:12s} algo={os.environ.get('R2DM_CONV_ALGO','x3')} spec={os.environ.get('R2DM_SPEC','1')} pieces={os.environ.get('PIECES','3')}: {ms / max(n,1) * 1e3:8.1f} us/call  {flop * n / ms / 1e9:7.1f} TF/s   "
      f"sclk median {statistics.median(f) / 1e6 if f else float('nan'):6.0f} MHz (min {min(f) / 1e6 if f else 0:.0f}, max {max(f) / 1e6 if f else 0:.0f})   "
      f"power median {statistics.median(p) / 1e6 if p else float('nan'):6.0f} W  [{len(steady)} samples, hwmon={H}]")
