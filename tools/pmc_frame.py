import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crispy_amd import synthetic_weights, synth_audio
from crispy_amd.denoise import DenoiseState
B, T = 4096, 25
ds = DenoiseState(synthetic_weights(0), B, 0)
x = synth_audio.batch_torch(B, T, torch.device("cuda:0")); y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(2):
    ds.process_device(x.data_ptr(), y.data_ptr(), T); ds.synchronize()
