#!/bin/bash
# tools/seg_probe.sh -- segment size experiments: alice29 at quality 5 (one small call) and text at qualities 2 / 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
for seg in 256 512 1024; do echo "== alice q5, segment $seg"; BROTLI_MI355X_SEGMENT_BYTES=$seg python tools/small_trace.py 30 | tail -1; done
for seg in 256 512 1024; do for q in 2 4; do echo "== 2 MiB q$q, segment $seg"; BROTLI_MI355X_SEGMENT_BYTES=$seg python tools/qspec_rounds.py 2 $q 2>&1 | tail -1; done; done
for seg in 1024 2048 4096; do for q in 2 4; do echo "== 64 MiB q$q, segment $seg"; BROTLI_MI355X_SEGMENT_BYTES=$seg python tools/qspec_rounds.py 64 $q 2>&1 | tail -1; done; done
for seg in 256 512; do for q in 2 4; do echo "== alice q$q, segment $seg"; SMALL_Q=$q BROTLI_MI355X_SEGMENT_BYTES=$seg python tools/small_trace.py 30 | tail -1; done; done
