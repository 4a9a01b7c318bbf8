"""tools/masked_cascade.py <kind> <MiB> <lgwin> <perturb at MiB> [size hint] -- how far does ONE flipped input byte reach in the
SEQUENTIAL parse of the reference (the oracle, quality 5), measured in commands that differ afterwards?  Without a size hint and
fed in 1 MiB writes the hasher is H5, whose StoreRangeOptBatch entries are masked past the first ring-buffer revolution
(mod.rs:1163-1232); with a hint above 4 MiB it is H6 (no masked entries).  The answer -- a heavy-tailed, near-critical cascade under
H5 -- is why the masked regime is parsed by one chain per stream (numbers in profiles/r04_masked_h5_cascade.txt).
Test infrastructure (uses the oracle).  kinds: markov mixed silesia enwik.  TRIALS=<n> sets the number of perturbations."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ['ORC_FAST']='1'
import numpy as np
import orc, synth
L=orc.lib()
cmd_dt=np.dtype([('ins','<u4'),('copy','<u4'),('dx','<u4'),('cp','<u2'),('dp','<u2')])
def parse(data, params):
    L.orc_encoder_create.restype = ctypes.c_void_p
    L.orc_encoder_destroy.argtypes = [ctypes.c_void_p]
    L.orc_encoder_set_parameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
    L.orc_encoder_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_encoder_compress_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t),
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    s=L.orc_encoder_create()
    out=[]
    def cb(opaque, kind, start, nbytes, cmds, n, dc):
        a=np.ctypeslib.as_array(ctypes.cast(cmds, ctypes.POINTER(ctypes.c_uint8)), shape=(n*16,)).view(cmd_dt).copy() if n else np.zeros(0,cmd_dt)
        out.append((kind,start,nbytes,a))
    cbo=orc.TRACE_CB(cb)
    for k,v in params: L.orc_encoder_set_parameter(s,k,v)
    L.orc_encoder_set_trace(s, ctypes.cast(cbo, ctypes.c_void_p), None)
    cap=L.orc_max_compressed_size(len(data))+64
    ob=ctypes.create_string_buffer(cap)
    ib=ctypes.create_string_buffer(data,len(data))
    ao=ctypes.c_size_t(cap); no=ctypes.c_void_p(ctypes.addressof(ob)); tot=ctypes.c_size_t(0)
    CH=1<<20
    for o in range(0,len(data),CH):
        m=min(CH,len(data)-o)
        ai=ctypes.c_size_t(m); ni=ctypes.c_void_p(ctypes.addressof(ib)+o)
        op=2 if o+m==len(data) else 0
        while True:
            ok=L.orc_encoder_compress_stream(s,op,ctypes.byref(ai),ctypes.byref(ni),ctypes.byref(ao),ctypes.byref(no),ctypes.byref(tot))
            assert ok
            if ai.value==0: break
    L.orc_encoder_destroy(s)
    assert ok
    # command start positions
    res=[]
    for kind,start,nbytes,a in out:
        if kind!=0: continue
        ins=a['ins'].astype(np.int64); cl=(a['copy']&0x1ffffff).astype(np.int64)
        tot=ins+cl
        pos=start+np.concatenate([[0],np.cumsum(tot)[:-1]])
        res.append(np.stack([pos,ins,cl,a['dx'].astype(np.int64),a['dp'].astype(np.int64)],1))
    return np.concatenate(res), cap-ao.value
kind=sys.argv[1]; n=int(sys.argv[2])<<20; w=int(sys.argv[3]); x=int(sys.argv[4])<<20
hint=int(sys.argv[5]) if len(sys.argv)>5 else 0
d={'markov':synth.markov_text,'mixed':synth.mixed,'silesia':lambda n: synth.silesia_like(n,min_segment=1<<18,max_segment=2<<20),'enwik':synth.enwik_like}[kind](n)
params=[(1,5),(2,w)]+([(5,hint)] if hint else [])
t=time.time(); A,sa=parse(d,params); print('parse',time.time()-t,'s cmds',len(A),'size',sa)
for trial in range(int(os.environ.get('TRIALS','4'))):
    xx=x+trial*70001
    d2=bytearray(d); d2[xx]^=0x55; d2=bytes(d2)
    B,sb=parse(d2,params)
    # commands as set of rows after xx
    sa_=set(map(tuple,A[A[:,0]>xx-100].tolist())); sb_=set(map(tuple,B[B[:,0]>xx-100].tolist()))
    diff=sorted(p for p in (sa_^sb_))
    posd=np.array(sorted(set(r[0] for r in diff)))
    print('perturb at',xx,'diff cmds',len(diff),'first',posd[0]-xx if len(posd) else None,'last',posd[-1]-xx if len(posd) else None, 'size',sb)
    if len(posd):
        h=np.histogram(posd-xx,bins=[0,1<<10,1<<12,1<<14,1<<16,1<<17,1<<18,1<<19,1<<20,1<<21,1<<22,1<<23,1<<26])
        print('  hist',h[0].tolist())
        cl=[];
        for p in (posd-xx).tolist():
            if cl and p-cl[-1][1]<512: cl[-1][1]=p; cl[-1][2]+=1
            else: cl.append([p,p,1])
        print('  clusters',len(cl),[(a,b-a,c) for a,b,c in cl][:60])
