import orc, emu, time, glob, os, sys, random
L = emu.lib()

def check(name, data, q=5, w=22, size_hint=None, catable=False, prefix=b"", seg=4096, lib=L):
    sh = len(data) if size_hint is None else size_hint
    params=[(1,q),(2,w),(5,sh)]
    if catable: params.append((167,1))
    t=time.time()
    mbs,st=emu.lz77_trace(lib,data,q,w,sh,catable,prefix,seg)
    te=time.time()-t
    c,tr=orc.stream_compress(data,params,prefix=prefix if prefix else None,collect_trace=True)
    # oracle trace positions are stream positions (incl. prefix)
    ok = len(tr)==len(mbs)
    if ok:
        for (kind,s,n,cm,dc),m in zip(tr,mbs):
            if (s,s+n)!=(m['start'],m['end']) or (kind!=0)!=m['uncompressed'] or dc!=m['dist_cache_after']:
                ok=False; print('  mb mismatch',(kind,s,n,len(cm),dc),(m['start'],m['end'],m['uncompressed'],len(m['cmds']),m['dist_cache_after'])); break
            if kind==0 and cm!=m['cmds']:
                ok=False
                for i,(x,y) in enumerate(zip(cm,m['cmds'])):
                    if x!=y: print('  cmd diff at',i,'of',len(cm),len(m['cmds']),x,y); break
                else: print('  cmd count', len(cm), len(m['cmds']))
                break
    else:
        print('  metablock count', len(tr), len(mbs), [(k,s,n) for k,s,n,_,_ in tr][:5], [(m['start'],m['end'],m['uncompressed']) for m in mbs][:5])
    print('%-28s q%d w%d n=%d seg=%d %s rounds=%d parsed=%d emu=%.2fs'%(name,q,w,len(data),seg,'OK' if ok else 'FAIL',st['rounds'],st['segments_parsed'],te))
    return ok

if __name__=='__main__':
    allok=True
    for f in sorted(glob.glob('/root/reference/testdata/*')):
        b=os.path.basename(f)
        if 'compressed' in b and b not in ('compressed_file','compressed_repeated'): continue
        d=open(f,'rb').read()
        if len(d)==0: continue
        for q in (5,6,7,8):
            allok&=check(b,d,q,22)
        allok&=check(b,d,5,18)
        allok&=check(b,d,5,22,seg=1024)
        allok&=check(b,d,5,22,seg=65536)
    print('ALL OK' if allok else 'SOME FAILED')
