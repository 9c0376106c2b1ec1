import sys, struct
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from jpeg_gpu_amd import synth
def parse(data):
    i=2; dht={}; comps=[]; scan_comps=[]
    while True:
        assert data[i]==0xFF; m=data[i+1]; L=(data[i+2]<<8)|data[i+3]; seg=data[i+4:i+2+L]
        if m==0xC4:
            p=0
            while p<len(seg):
                tc_th=seg[p]; counts=seg[p+1:p+17]; n=sum(counts); vals=seg[p+17:p+17+n]; dht[tc_th]=(counts,vals); p+=17+n
        elif m==0xC0:
            nf=seg[5]; comps=[(seg[6+3*c],seg[7+3*c]>>4,seg[7+3*c]&15) for c in range(nf)]
            H=(seg[1]<<8)|seg[2]; W=(seg[3]<<8)|seg[4]
        elif m==0xDA:
            ns=seg[0]; scan_comps=[(seg[1+2*c],seg[2+2*c]>>4,seg[2+2*c]&15) for c in range(ns)]
            return dht,comps,scan_comps,W,H,i+2+L
        i+=2+L
def build(counts,vals):
    code=0;k=0;tab={}
    for ln in range(1,17):
        for _ in range(counts[ln-1]):
            tab[(ln,code)]=vals[k];k+=1;code+=1
        code<<=1
    return tab
def run(w,h,samp,q=90):
    data=synth.synthetic_jpeg(w,h,samp,quality=q,seed=1234)
    dht,comps,sc,W,Hh,pos=parse(data)
    # unstuff
    raw=bytearray(); i=pos
    while i<len(data):
        b=data[i]
        if b==0xFF:
            if data[i+1]==0: raw.append(0xFF); i+=2; continue
            else: break
        raw.append(b); i+=1
    bits=int.from_bytes(bytes(raw)+b'\xff'*8,'big'); nbits=(len(raw)+8)*8
    tabs={k:build(*v) for k,v in dht.items()}
    hmax=max(c[1] for c in comps); vmax=max(c[2] for c in comps)
    mcux=(W+8*hmax-1)//(8*hmax); mcuy=(Hh+8*vmax-1)//(8*vmax)
    slots=[]
    for ci,(cid,hs,vs) in enumerate(comps):
        td,ta=[(d,a) for (c,d,a) in sc if c==cid][0]
        slots += [(td,0x10|ta)]*(hs*vs)
    p=0
    def peek(n): return (bits>>(nbits-p-n))&((1<<n)-1)
    syms=[] # (is_dc, total_bits, is_eob)
    for m in range(mcux*mcuy):
        for td,ta in slots:
            # DC
            t=tabs[td]
            for ln in range(1,17):
                c=peek(ln)
                if (ln,c) in t: s=t[(ln,c)]; break
            p+=ln+s; syms.append((1,ln+s,0,0))
            k=1; t=tabs[ta]
            while k<64:
                for ln in range(1,17):
                    c=peek(ln)
                    if (ln,c) in t: rs=t[(ln,c)]; break
                r=rs>>4; s=rs&15
                p+=ln+s
                if rs==0: syms.append((0,ln,1,64)); break
                syms.append((0,ln+s,0,r+1)); k+=r+1
    print(w,h,samp,"scan bytes",len(raw),"symbols",len(syms),"bits/symbol %.2f"%(8*len(raw)/len(syms)))
    for Wd,maxn in ((9,3),(10,3),(11,3),(12,3),(12,4),(13,4),(14,4),(16,5)):
        steps=0;i=0;n=len(syms)
        while i<n:
            isdc,tb,eob,adv=syms[i]
            if isdc or tb>Wd: steps+=1;i+=1;continue
            # try pack
            used=0;cnt=0;j=i;pref=0
            while j<n and cnt<maxn:
                d2,tb2,eob2,adv2=syms[j]
                if d2 or used+tb2>Wd or pref>31: break
                used+=tb2;cnt+=1;j+=1
                if eob2: break
                pref+=adv2
            if cnt>=2: steps+=1;i=j
            else: steps+=1;i+=1
        print("   window %2d bits, up to %d symbols: %.3f symbols per step, steps per 128 B: %.1f"%(Wd,maxn,len(syms)/steps, steps/(len(raw)/128)))
run(640,360,"420")
run(640,360,"444")
