import torch, time, sys
sys.path.insert(0, '.')
from vocoderproject_amd import BatchVocoderProcessor
from vocoderproject_amd.synth import make_streams
for mode in ("pitch","both"):
    p = BatchVocoderProcessor(vocBool=int(mode=="both")); p.prepareToPlay(48000.0, 1024, 256); p.set_iir_mode("fast"); p.set_yin_mode("xcorr")
    x = make_streams(256, 1024*4, fs=48000.0, device="cuda").view(256,3,4,1024).permute(2,0,1,3).contiguous()
    y = torch.empty((256,2,1024), device="cuda")
    for i in range(20): p.process_device(x[i%4], y)
    torch.cuda.synchronize(); t=time.perf_counter()
    for i in range(200): p.process_device(x[i%4], y)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/200
    print(mode, "%.1f us/block  %.2f M frames(278)/s"%(dt*1e6, 256*1024/278/dt/1e6), p.pitch_kernel_name(), p.yin_certified_counts())
