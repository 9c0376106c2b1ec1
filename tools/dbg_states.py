import sys, os, ctypes as C, subprocess, json
sys.path.insert(0,'.')
import numpy as np
if len(sys.argv) > 1:
    from jpeg_gpu_amd import lib, synth
    lib.L.jga_huff_debug_states.argtypes=[C.c_void_p,C.c_void_p,C.c_longlong]; lib.L.jga_huff_debug_states.restype=C.c_longlong
    data=[synth.synthetic_jpeg(640,360,"420",quality=90,seed=5)]
    hb=lib.HuffBatch(1,len(data[0])+4096); g=hb.prepare(data)
    stride=(g.coef_shorts*2+255)//256*128; d=lib.DeviceBuffer(stride*2)
    try:
        rounds=hb.decode(d.ptr,stride); err=""
    except Exception as e:
        rounds=-1; err=str(e)
    n=lib.L.jga_huff_debug_states(hb.ptr,None,0); S=np.zeros(n,np.uint64); lib.L.jga_huff_debug_states(hb.ptr,S.ctypes.data,n)
    np.save(sys.argv[1],S); print("rounds",rounds,err)
else:
    for tag,cfg in (("a","256,256,3"),("b","3,2,6"),("c","1,1,4")):
        r=subprocess.run([sys.executable,__file__,"/tmp/S_%s.npy"%tag],env=dict(os.environ,JGA_HUFF_ITERS=cfg),capture_output=True,text=True); print(tag,cfg,r.stdout.strip(),r.stderr[-300:])
    a=np.load("/tmp/S_a.npy")
    for tag in "bc":
        b=np.load("/tmp/S_%s.npy"%tag); bad=np.nonzero(a!=b)[0]
        print(tag,"differing states",len(bad),"of",len(a),"first",bad[:10], [(hex(int(a[i])),hex(int(b[i]))) for i in bad[:4]])
