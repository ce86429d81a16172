import sys, time; sys.path.insert(0,'.')
import numpy as np
from vocoderproject_amd import BatchVocoderProcessor
from vocoderproject_amd.synth import make_streams
S,N=256,1024
x=np.ascontiguousarray(make_streams(S,N*8).numpy())
p=BatchVocoderProcessor(vocBool=0); p.prepareToPlay(44100.0,N,S); p.set_iir_mode("fast")
blks=[np.ascontiguousarray(x[:,:,b*N:(b+1)*N]) for b in range(8)]
for b in range(8): p.process(blks[b])
t=time.perf_counter()
for i in range(100): p.process(blks[i%8])
dt=(time.perf_counter()-t)/100
print("host-pointer vp_process_block: %.3f ms per call, %.2f M frames/s" % (dt*1e3, S*N/256/dt/1e6))
