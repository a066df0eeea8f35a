"""The recurrent encoder's 5x5 head (conv_bf16_head.hip) alone, mixed configuration, B = 8 / 2 x 480 x 640, 200 back-to-back calls timed with
events.  usage: python tools/head_probe.py <label>   (round 6: A/B of library variants built with ablation switches)"""
import sys, torch
sys.path.insert(0, '.')
from ess_amd import hip
from ess_amd.e2vid.model.submodules import ConvLayer
hip.set_compute('mixed')
torch.manual_seed(0)
head = ConvLayer(2, 32, kernel_size=5, stride=1, padding=2).cuda().eval()
x = torch.randn(8, 2, 480, 640, device='cuda')
x[torch.rand_like(x) < 0.7] = 0
with torch.no_grad():
    for _ in range(5): y = head.forward_mixed(x, want_fp32=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): y = head.forward_mixed(x, want_fp32=False)
    e1.record(); torch.cuda.synchronize()
print(sys.argv[1], 'head us per call', round(e0.elapsed_time(e1) / 200 * 1e3, 2))
