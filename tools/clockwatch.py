"""Samples the shader clock (rocm-smi) while the scorer GEMM runs back to back for a few seconds:
python tools/clockwatch.py  -> sclk samples under fp32-MFMA load vs idle."""
import os, subprocess, sys, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'a-recsys_amd'))
import torch
from arx import ops

def sclk():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:
        return str(e)
    m = re.findall(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
    return m[0] if m else out.strip().replace('\n', ' | ')[:200]

dev = torch.device('cuda:0')
ws = ops.Workspace(dev)
B, S, d = 16384, 1024, 128
U = torch.randn(B, d, device=dev); I = torch.randn(S, d, device=dev); L = torch.empty(B, S, device=dev)
print('idle sclk:', sclk())
samples, stop = [], False
def watch():
    while not stop:
        samples.append(sclk())
        time.sleep(0.15)
th = threading.Thread(target=watch); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 4.0:
    for _ in range(200):
        ops.gemm(U, I, L, ws, transB=True)
    n += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
us = e0.elapsed_time(e1) * 1e3 / n
print('launches', n, 'us/launch %.1f' % us, 'TF %.1f' % (2.0 * B * S * d / us / 1e6))
print('sclk under load:', samples)
