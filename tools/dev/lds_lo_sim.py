"""LDS bank-conflict model (MI355X_MICROARCH.md section LDS) of the accesses to the token-major lo tiles of fp16c (gemm16_common.h RSL / lo_pos):
the MFMA reader (ds_read_b128), the LayerNorm writer and the y staging (ds_write_b32), with and without the row swizzle tried first.   python tools/dev/lds_lo_sim.py"""
RSL=272
def lo_pos(k): return (k & ~63) + 32*((k>>3)&1) + 8*((k>>4)&3) + (k&7)
def cyc(addr_bytes, groups, mod):
    tot=extra=0
    for g in groups:
        banks={}
        for l in g:
            a=addr_bytes[l]
            if a is None: continue
            banks.setdefault((a//4)%mod,set()).add(a//4)
        c=max(len(v) for v in banks.values()); tot+=c; extra+=c-1
    return tot,extra
def b128(addr, groups):
    tot=extra=0
    for g in groups:
        slots={}
        for l in g:
            for d in range(4):
                slots.setdefault((addr[l]//4+d)%64,set()).add(addr[l]//4+d)
        c=max(len(v) for v in slots.values()); tot+=c; extra+=c-1
    return tot,extra
G32=[list(range(32)),list(range(32,64))]
G16n=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G16n=G16n+[[x+32 for x in g] for g in G16n]
for name,swz in (("no swizzle",lambda row:0),("current swizzle",lambda row:((row>>4)&7)<<5)):
    # R1: alo reads
    t=e=0
    for part in range(4):
        for mt in range(4):
            for half16 in (0,16):
                ad=[]
                for lane in range(64):
                    lrow=lane&31; lhalf=lane>>5; row=mt*32+lrow
                    ad.append(row*RSL + ((64*part+32*lhalf) ^ swz(row)) + half16)
                a,b=b128(ad,G16n); t+=a; e+=b
    print(name,"R1 alo ds_read_b128: cycles",t,"extra",e)
    # W1: LN lo writes (per wave)
    t=e=0
    for wave in range(8):
        for mt in range(4):
            for q in range(4):
                ad=[]
                for lane in range(64):
                    lrow=lane&31; lhalf=lane>>5; row=mt*32+lrow
                    ad.append(row*RSL + ((lo_pos(wave*32+8*q)+4*lhalf) ^ swz(row)))
                a,b=cyc(ad,G32,32); t+=a; e+=b
    print(name,"W1 LN lo ds_write_b32: cycles",t,"extra",e, "(per 8 waves)")
    # W2: ylo staging writes
    for mapping in ("cur","cg=lane"):
        t=e=0
        for wave in range(8):
            for a4 in range(4):
                for el in range(4):
                    ad=[]
                    for lane in range(64):
                        if mapping=="cur": cg=wave*8+(lane&7); tk=((lane>>3)&7)*16
                        else: cg=lane; tk=wave*16
                        row=tk+4*a4+el
                        ad.append(row*RSL + (lo_pos(4*cg) ^ swz(tk)))
                    a,b=cyc(ad,G32,32); t+=a; e+=b
        print(name,"W2 ylo staging ds_write_b32 mapping",mapping,": cycles",t,"extra",e)
