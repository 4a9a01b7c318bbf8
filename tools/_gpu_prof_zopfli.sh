cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
rm -rf $OUT/r04_zopfli_kt
cat > /tmp/zp.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import emu, gpulib, synth
L = gpulib.lib()
a = synth.alice()
for q in (10, 11):
    t = time.time(); out, st = emu.encode_stream(L, a, [(1, q), (2, 22), (5, len(a))]); print(q, len(out), time.time() - t, flush=True)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_zopfli_kt -o r04 -- python /tmp/zp.py > $OUT/r04_zopfli_kt.log 2>&1
cp $(find $OUT/r04_zopfli_kt -name "*kernel_stats.csv" | head -1) $OUT/r04_zopfli_kernel_stats.csv
rm -rf $OUT/r04_zopfli_kt
head -12 $OUT/r04_zopfli_kernel_stats.csv | cut -c1-110
grep -v "^W\|^E\|^I" $OUT/r04_zopfli_kt.log | tail -3
