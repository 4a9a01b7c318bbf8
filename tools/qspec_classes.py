"""tools/qspec_classes.py [MiB] -- qualities 2..4 on the speculative path over input classes (text, random, mixed, Silesia-like, zeros, stretches,
repeated excerpts): time, launches of the parse (lz77_rounds), identity with the oracle.  One JSON object per line."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc, synth, test_cabi
lib = test_cabi._load("gpu")
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 16) << 20
lib.compress(synth.markov_text(1 << 16), 5, 22)
classes = [("text", synth.markov_text(n, 5)), ("random", synth.random_bytes(n)), ("mixed", synth.mixed(n, 7)), ("silesia_like", synth.silesia_like(n, 4)),
           ("zeros", bytes(n)), ("stretches", synth.stretches(n, 9)), ("repeated_excerpts", synth.repeated_excerpts(n, 35))]
for name, d in classes:
    for q in (2, 3, 4):
        os.environ["BROTLI_MI355X_DEBUG"] = "1"
        r, w = os.pipe()
        saved = os.dup(2)
        os.dup2(w, 2)
        t = time.time()
        out = lib.compress(d, q, 22)
        dt = time.time() - t
        os.dup2(saved, 2)
        os.close(w)
        os.close(saved)
        log = b""
        os.set_blocking(r, False)
        try:
            while True:
                chunk = os.read(r, 1 << 20)
                if not chunk:
                    break
                log += chunk
        except BlockingIOError:
            pass
        os.close(r)
        rounds = log.count(b"quick round")
        serial = b"serial path takes over" in log
        coarse = b"one chain per block from here" in log
        t = time.time()
        want = orc.compress(d, q, 22)
        cpu = time.time() - t
        print(json.dumps({"input": name, "quality": q, "MiB": n >> 20, "ms": round(dt * 1e3, 1), "MBps": round(len(d) / dt / 1e6, 1), "rounds": rounds, "fell_back_to_serial": serial, "coarse_restart": coarse,
                          "identical": out == want, "cpu_oracle_MBps": round(len(d) / cpu / 1e6, 1), "ratio": round(len(out) / len(d), 3)}), flush=True)
