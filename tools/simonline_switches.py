"""simonline under the switches of its banded similarity (f16-split / fp32 kernels, LDS-DMA or register-staged K loop, look-back or
diagonal band layout): every form must give the default's output (the similarity only decides the lists). Run on the GPU box."""
import sys, os, subprocess, numpy as np
root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code=("import sys, numpy as np; sys.path[:0]=[%r,%r]; import repet; from repet_synth import synth; x=synth(21,22050,2,5); y=repet.simonline(x,22050); np.save(sys.argv[1], y)") % (os.path.join(root,'repet-python_amd'), root)
outs={}
for name,env in [('default',{}),('gram_f32',{'REPET_GRAM':'f32'}),('dma0',{'REPET_GRAM_DMA':'0'}),('lookback0',{'REPET_BAND_LOOKBACK':'0'}),('band_f32',{'REPET_GRAM_BAND':'f32'})]:
    out='/tmp/so_%s.npy'%name
    subprocess.check_call([sys.executable,'-c',code,out], env=dict(os.environ, **env))
    outs[name]=np.load(out)
d=outs['default']
for k,v in outs.items():
    print(k, 'equal' if np.array_equal(d,v,equal_nan=True) else 'rms %.3e' % np.sqrt(np.nanmean((d-v)**2)))
