"""fp16x3 get_outputs on 133 120 points: with a -DNEAT_X3_TIMING=1 build workgroup 0 prints the cycle counts of its first batches."""
import torch, sys
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from test_gpu_parity import build_model
dev = torch.device('cuda:0')
m = build_model(dev, "rough", precision="fp16x3")
x = (torch.rand(133120, 3) * 4 - 2).to(dev)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        o = m.implicit_network.get_outputs(x)
torch.cuda.synchronize()
