cd /root/repo
B="python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
for v in "" w64 occ4 occ6; do
  if [ -n "$v" ]; then export BROTLI_MI355X_LIB=/root/repo/rust-brotli_amd/libbrotli_mi355x_$v.so; else unset BROTLI_MI355X_LIB; fi
  echo "== variant [$v]"
  timeout 120 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('launch_ms'), d.get('stages'))"
done
unset BROTLI_MI355X_LIB
BROTLI_MI355X_TIMELINE=1 timeout 120 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | grep "round timeline" | tail -2
