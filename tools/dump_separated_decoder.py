import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_hip_bf16_separated import _trained_decoder
cfg, sd_e, sd_d, ev, lab, hist = _trained_decoder(1, 5, 2, 480, 640, 11, 700, 3e-3)
torch.save({k: v.half() if False else v for k, v in sd_d.items()}, '/root/repo/gpurun_out/sep_sd_d.pt')
print(hist[0], hist[-1])
