import sys, os, time, json
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/rust-brotli_amd')
import large_cases, brotli_mi355x, hashlib
FROZEN=json.load(open('/root/repo/tests/golden/large_hashes.json'))
name='c4_silesia_128MiB_multi8_h5'
data=large_cases.make_input(name, FROZEN)
lib=brotli_mi355x.default_library()
for rep in range(2):
    t=time.time(); out=bytes(lib.BrotliCompress(data,{1:5,2:22},8)); dt=time.time()-t
    print('rep',rep,'%.1fs'%dt, hashlib.sha256(out).hexdigest()==FROZEN[name]['stream_sha256'], flush=True)
