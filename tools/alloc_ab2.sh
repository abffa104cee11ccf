#!/bin/bash
# spectral 512^3: piece size x chunked / whole-box passes, several fresh processes each
for pol in $1; do for ch in "" 0; do for p in 1 2 3; do
  printf "PFHIP_ALLOC=%-14s PFHIP_FFT3D_CHUNK=%-3s process %d: " $pol "${ch:-def}" $p
  if [ -z "$ch" ]; then unset PFHIP_FFT3D_CHUNK; else export PFHIP_FFT3D_CHUNK=$ch; fi
  PFHIP_ALLOC=$pol python tools/stream_vs_memory_probe.py --chunked --keep 0 --streams 1 --handles 3 2>&1 | grep "handle   [012]" | awk '{printf "%s ", $(NF-1)} END {print ""}'
done; done; done
