"""tools/lds_bank_model.py — offline bank-conflict model of the front end's FFT-buffer accesses (no GPU needed).
Prints the LDS cycles of one 960-point FFT (digit-reversal scatter, five radix stages, spectrum read-out) for four
streams of a wave under candidate index swizzles / slice strides / butterfly-to-lane assignments (S strided, B blocked).
The model reproduces the measured conflict count of the current layout (4856 modelled vs 5190 measured extra cycles per
wave-FFT, profiles/r02c_frontend_phase_pmc.txt).
Bank-conflict model of the front end's FFT-buffer accesses (ds_write_b64 / ds_read_b64 of float2 elements).
Per MI355X_MICROARCH.md: ds_read_b64: two 32-lane groups, 64 banks of 4 B (an 8-byte access covers 2 banks);
ds_write_b64: four contiguous 16-lane groups, 32 banks.  Cost of a group = max over banks of distinct 8-byte slots."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
L=16
def bitrev_table():
    from oracle.oracle import Oracle
    o=Oracle(None); return o.tables()[1].astype(int)
BR=bitrev_table()
def cost(addrs_bytes, kind):
    # addrs_bytes: array[64] byte address per lane (or -1 inactive)
    a=np.asarray(addrs_bytes)
    if kind=='r64': groups=[range(0,32),range(32,64)]; nb=64
    else: groups=[range(0,16),range(16,32),range(32,48),range(48,64)]; nb=32
    tot=0
    for g in groups:
        per={}
        for l in g:
            if a[l]<0: continue
            slot=a[l]//8
            for bank in ((a[l]//4)%nb, (a[l]//4+1)%nb):
                per.setdefault(bank,set()).add(slot)
        tot+= max((len(v) for v in per.values()), default=0)
    return tot
def run(P, slice_bytes):
    """P: index map element->physical element; slice stride in bytes. returns (ideal, actual) LDS cycles for one FFT incl. scatter"""
    ideal=act=0
    def acc(elem_of_lane, kind):
        nonlocal ideal, act
        # elem_of_lane: function (l, sub)->element index or None ; build 64-lane addresses
        a=np.full(64,-1)
        for sub in range(4):
            for l in range(L):
                e=elem_of_lane(l)
                if e is not None: a[sub*L+l]=sub*slice_bytes+8*P(e)
        n_groups=2 if kind=='r64' else 4
        ideal+=n_groups; act+=cost(a,kind)
    # scatter: lane l, it: samples i=4*(l+16*it)+c -> write F[bitrev[i]]
    for it in range(15):
        for c in range(4):
            acc(lambda l: BR[4*(l+L*it)+c], 'w64')
    # radix-4 m=1
    for b0 in range(0,240,L):
        for k in range(4): acc(lambda l: 4*(b0+l)+k, 'r64')
        for k in range(4): acc(lambda l: 4*(b0+l)+k, 'w64')
    for m,mm in ((4,16),(16,64)):
        for b0 in range(0,240,L):
            for k in range(4): acc(lambda l: ((b0+l)//m)*mm+(b0+l)%m+k*m, 'r64')
            for k in range(4): acc(lambda l: ((b0+l)//m)*mm+(b0+l)%m+k*m, 'w64')
    for b0 in range(0,320,L):
        for k in range(3): acc(lambda l: ((b0+l)>>6)*192+((b0+l)&63)+64*k, 'r64')
        for k in range(3): acc(lambda l: ((b0+l)>>6)*192+((b0+l)&63)+64*k, 'w64')
    for u0 in range(0,192,L):
        for k in range(5): acc(lambda l: u0+l+192*k, 'r64')
        for k in range(5): acc(lambda l: u0+l+192*k, 'w64')
    return ideal, act
def _simple_table():
    for name,P,sl in [("current", lambda i:i, 8256),
                      ("stride+16B", lambda i:i, 8256+16),
                      ("pad i+(i>>5)", lambda i:i+(i>>5), 8256+256),
                      ("pad i+(i>>4)", lambda i:i+(i>>4), 8256+512),
                      ("pad i+(i>>3)", lambda i:i+(i>>3), 8256+1024),
                      ("xor (i>>5)&31", lambda i: i ^ ((i>>5)&31), 8256),
                      ("xor ((i>>4)&15)", lambda i: i ^ ((i>>4)&15), 8256),
                      ("xor ((i>>2)&3)<<? ", lambda i: i ^ (((i>>6)&3)<<2) ^ ((i>>4)&3), 8256),
                      ]:
        for extra in (0,16,32,64,128):
            i,a=run(P,sl+extra)
            print(f"{name:20s} slice={sl+extra:6d}  ideal {i}  actual {a}  ratio {a/i:.2f}")


# ---- search over layouts ------------------------------------------------------------------------------------
def stage_cost(P, slice_bytes, stage, blocked):
    ideal=act=0
    def acc(fn, kind):
        nonlocal ideal, act
        a=np.full(64,-1)
        for sub in range(4):
            for l in range(L):
                e=fn(l)
                if e is not None: a[sub*L+l]=sub*slice_bytes+8*P(e)
        ideal+= 2 if kind=='r64' else 4; act+=cost(a,kind)
    def lanes(nb, it):   # butterfly index of lane l at iteration it
        n_it=(nb+L-1)//L
        if blocked: return lambda l: (l*n_it+it) if (l*n_it+it)<nb else None
        return lambda l: (l+L*it) if (l+L*it)<nb else None
    if stage=='scatter':
        for it in range(15):
            for c in range(4): acc(lambda l: BR[4*(l+L*it)+c], 'w64')
    elif stage=='m1':
        for it in range(15):
            f=lanes(240,it)
            for k in range(4): acc(lambda l: None if f(l) is None else 4*f(l)+k,'r64')
            for k in range(4): acc(lambda l: None if f(l) is None else 4*f(l)+k,'w64')
    elif stage in ('m4','m16'):
        m,mm=(4,16) if stage=='m4' else (16,64)
        for it in range(15):
            f=lanes(240,it)
            for k in range(4): acc(lambda l: None if f(l) is None else (f(l)//m)*mm+f(l)%m+k*m,'r64')
            for k in range(4): acc(lambda l: None if f(l) is None else (f(l)//m)*mm+f(l)%m+k*m,'w64')
    elif stage=='r3':
        for it in range(20):
            f=lanes(320,it)
            for k in range(3): acc(lambda l: None if f(l) is None else (f(l)>>6)*192+(f(l)&63)+64*k,'r64')
            for k in range(3): acc(lambda l: None if f(l) is None else (f(l)>>6)*192+(f(l)&63)+64*k,'w64')
    elif stage=='r5':
        for it in range(12):
            f=lanes(192,it)
            for k in range(5): acc(lambda l: None if f(l) is None else f(l)+192*k,'r64')
            for k in range(5): acc(lambda l: None if f(l) is None else f(l)+192*k,'w64')
    elif stage=='spec':   # spectrum read k = l + 16*it (Y store), 400 bins
        for it in range(25): acc(lambda l: l+L*it,'r64')
    return ideal,act
stages=['scatter','m1','m4','m16','r3','r5','spec']
cands={"ident":lambda i:i}
for k in (3,4,5,6):
    for p in (1,2,3):
        cands[f"i+{p}*(i>>{k})"]=(lambda k,p: lambda i:i+p*(i>>k))(k,p)
for sh,msk in ((5,31),(4,15),(4,31),(3,31),(6,15),(5,7)):
    cands[f"xor((i>>{sh})&{msk})"]=(lambda sh,msk: lambda i:i^((i>>sh)&msk))(sh,msk)
cands["xor(i>>5&31)^(i>>2&... )"]=lambda i: i ^ ((i>>5)&31) ^ (((i>>10)&1)<<4)
best=[]
for name,P in cands.items():
    mx=max(P(i) for i in range(960))+1
    for extra in (0,16,32,64,128,272):
        sl=((mx*8+576+15)//16)*16+extra
        tot_i=tot_a=0; detail=[]
        for st in stages:
            r=[stage_cost(P,sl,st,b) for b in ((False,True) if st not in ('scatter','spec') else (False,))]
            i,a=min(r,key=lambda x:x[1]); tot_i+=i; tot_a+=a; detail.append((st,a, 'B' if len(r)>1 and r[1][1]<r[0][1] else 'S'))
        best.append((tot_a,name,sl,tot_i,detail))
best.sort()
for b in best[:8]: print(b[0],b[1],b[2],b[3],b[4])
print("current:", [x for x in best if x[1]=="ident"][0])
