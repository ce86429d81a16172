"""8 blocks per call against one block per call, pitch corrector only, 256 streams (and the wave-specialised kernels against the phase kernels)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from vocoderproject_amd import BatchVocoderProcessor
from vocoderproject_amd.synth import make_streams
S, N, MB, U = 256, 1024, 8, 16
dev = torch.device('cuda', 0)
x = make_streams(S, N * U, device=dev).view(S, 3, U, N).permute(2, 0, 1, 3).contiguous()
xm = x[:, :, 0, :].contiguous()
y = torch.empty((MB, S, 2, N), dtype=torch.float32, device=dev)
for ws in (True, False):
    for iir in ('fast', 'exact'):
        p = BatchVocoderProcessor(vocBool=0); p.prepareToPlay(44100.0, N, S); p.set_iir_mode(iir); p.set_yin_mode('xcorr'); p.set_wave_specialised(ws)
        for mb in (1, MB):
            for i in range(4): p.process_blocks_mono_device(xm[(i * mb) % U:(i * mb) % U + mb], y[:mb])
            torch.cuda.synchronize(); t0 = time.perf_counter(); K = 320 // mb
            for i in range(K): p.process_blocks_mono_device(xm[(i * mb) % U:(i * mb) % U + mb], y[:mb])
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"wave_specialised={ws} iir={iir} blocks_per_call={mb}: {S*N*mb//256*K/dt/1e6:.2f} M frames/s, {dt/K/mb*1e6:.1f} us per block", flush=True)
        t = p.debug_stamps(reset=False)
        print("   timeouts", [round(t[i] * 100) for i in (59, 60, 61)])
