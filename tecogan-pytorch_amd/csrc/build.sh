#!/bin/bash
# Build libtecogan_hip.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -Wall -Wno-unused-function"
# objects are reused only when they were compiled with the same flags
STAMP=".build_flags"
if [ ! -f "$STAMP" ] || [ "$(cat $STAMP)" != "$FLAGS ${EXTRA_FLAGS:-}" ]; then
  rm -f tg_*.o
  echo "$FLAGS ${EXTRA_FLAGS:-}" > "$STAMP"
fi
OBJS=()
PIDS=()
for f in tg_*.hip; do
  o="${f%.hip}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ tg_common.h -nt "$o" ] || [ ../../include/tecogan_hip.h -nt "$o" ]; then
    # per-file flags: a line `// TG_FILE_FLAGS: ...` in the source (e.g. -fno-slp-vectorize for a hand-scheduled loop)
    ff=$(sed -n 's/^\/\/ TG_FILE_FLAGS: *//p' "$f" | head -1)
    echo "hipcc $f $ff"
    rm -f "$o"                       # a failed compile must not leave a stale object to link
    $HIPCC $FLAGS $ff ${EXTRA_FLAGS:-} -c "$f" -o "$o" &
    PIDS+=($!)
  fi
  OBJS+=("$o")
done
for pid in "${PIDS[@]:-}"; do      # bare `wait` returns 0 even when a job failed
  [ -z "$pid" ] || wait "$pid" || { echo "build.sh: a compile job failed" >&2; exit 1; }
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libtecogan_hip.so "${OBJS[@]}" -ldl
echo "built $(cd .. && pwd)/libtecogan_hip.so"
