mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_quality_2_4.py::test_one_shot_and_multi_shard_gpu tests/test_quality_2_4.py::test_past_the_ring_and_shards_gpu tests/test_quality_2_4.py::test_flushes_gpu tests/test_fuzz_smoke.py::test_api_sweep_quality_2_4_device -x -q -m gpu --durations=4 > gpurun_out/r04_f3_gpu4.log 2>&1
tail -8 gpurun_out/r04_f3_gpu4.log
python - <<'PY' > gpurun_out/r04_f3_multi16.jsonl 2>&1
import sys, time, json
sys.path.insert(0, "tests"); sys.path.insert(0, "rust-brotli_amd")
import torch  # noqa
import orc, synth, test_cabi
lib = test_cabi._load("gpu")
big = synth.markov_text(16 << 20, 77)
lib.compress(big[:65536], 2, 22)
for q in (2, 3, 4):
    t = time.time(); out = bytes(lib.BrotliCompress(big, {1: q, 2: 22}, 16)); dt = time.time() - t
    t = time.time(); want = orc.compress_multi(big, [(1, q), (2, 22)], 16); cpu = time.time() - t
    print(json.dumps({"workload": "q%d_text_16MiB_multi16" % q, "seconds": round(dt, 3), "value": round(len(big) / dt / 1e6, 2), "unit": "MB/s",
                      "identical_to_cpu_oracle": out == want, "cpu_oracle_MBps_one_core": round(len(big) / cpu / 1e6, 1)}), flush=True)
PY
cat gpurun_out/r04_f3_multi16.jsonl
