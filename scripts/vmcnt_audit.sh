#!/bin/bash
# Static audit of the gfx950 ISA of every kernel in csrc/: global loads / stores against vector-memory drains (s_waitcnt vmcnt(0)).
# A kernel whose drains are of the order of its loads pays one exposed memory round trip per load: invisible where many waves share
# a CU, the whole run time of a small-grid launch.  (Counts are static: over all branches of a kernel, loops counted once.)
# usage: scripts/vmcnt_audit.sh [file.hip ...]   (no GPU needed; hipcc cross-compiles)
HERE="$(cd "$(dirname "$0")/.." && pwd)"; CSRC=$HERE/adaptive_voice_conversion_amd/csrc
OUT=${AVC_AUDIT_DIR:-/tmp/avc_audit}; mkdir -p $OUT
FILES=("$@"); [ ${#FILES[@]} -eq 0 ] && FILES=($CSRC/*.hip)
for f in "${FILES[@]}"; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I$CSRC -I$HERE/include -S --cuda-device-only -o $OUT/$b.s $f 2>/dev/null &
done
wait
for f in "${FILES[@]}"; do
  b=$(basename $f .hip)
  [ -s $OUT/$b.s ] || continue
  awk -v file=$b '
    /^_Z[A-Za-z0-9_]*:/ { name = $1; sub(":", "", name) }
    /s_waitcnt vmcnt\(0\)/ { w0[name]++ }
    /s_waitcnt vmcnt/ { w[name]++ }
    /global_load_lds|buffer_load.*lds/ { dma[name]++ ; next }
    /global_load|buffer_load|flat_load/ { l[name]++ }
    /global_store|buffer_store|flat_store/ { st[name]++ }
    /\.vgpr_count:/ { }
    END { for (n in l) printf "%-14s %-100s loads=%-4d dma=%-3d stores=%-4d waits=%-4d drains=%-4d\n", file, substr(n, 1, 100), l[n], dma[n], st[n], w[n], w0[n] }' $OUT/$b.s
done | sort -t= -k6 -n -r | c++filt 2>/dev/null | cut -c1-230
