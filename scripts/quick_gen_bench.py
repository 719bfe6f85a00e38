import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ste_gan_amd.models.generator import EMGGeneratorGanTTS
torch.manual_seed(0)
g = EMGGeneratorGanTTS("SPEECH_UNITS", 256, 17, 8).cuda()
B = 64
su = torch.randn(B, 12, 256, device="cuda"); sess = torch.randint(0, 17, (B,), device="cuda"); spk = torch.zeros(B, dtype=torch.long, device="cuda")
for _ in range(3): g.generate(su, sess, spk)
torch.cuda.synchronize()
t0 = time.time(); n = 20
for _ in range(n): g.generate(su, sess, spk)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"gen fwd B=64: {dt*1e3:.3f} ms  {B*192/dt:.0f} frames/s  {53.02*2/dt/1e3:.1f} TFLOP/s nominal")
