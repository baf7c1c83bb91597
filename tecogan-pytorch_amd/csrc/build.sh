#!/bin/bash
# Build libtecogan_hip.so for gfx950 (MI355X).  Cross-compiles without a GPU.
#
# The in-tree library (../libtecogan_hip.so, the one the package loads) is ALWAYS built with exactly the flags below:
# EXTRA_FLAGS / lab switches are refused for it.  A lab build (ablation / A-B variants, timing only) is
#   TG_LAB_BUILD=1 OUT=/some/dir EXTRA_FLAGS="-DWR_RDFORM=1" bash build.sh
# which compiles with -DTG_LAB=1 into OUT (objects and libtecogan_lab.so), never into the package; load it with
# TECOGAN_HIP_LIB.  tg_build_info() of either library says which one it is (tests/test_abi_cpu.py asserts lab=0).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=on -Wall -Wno-unused-function"
OBJDIR=.
LIB=../libtecogan_hip.so
if [ "${TG_LAB_BUILD:-0}" = "1" ]; then
  [ -n "${OUT:-}" ] || { echo "build.sh: TG_LAB_BUILD=1 needs OUT=<directory outside the package>" >&2; exit 1; }
  mkdir -p "$OUT"
  OUT=$(cd "$OUT" && pwd)
  case "$OUT/" in "$(cd .. && pwd)"/*) echo "build.sh: a lab build may not write into the package ($OUT)" >&2; exit 1;; esac
  FLAGS="$FLAGS -DTG_LAB=1 ${EXTRA_FLAGS:-}"
  OBJDIR="$OUT"
  LIB="$OUT/libtecogan_lab.so"
elif [ -n "${EXTRA_FLAGS:-}" ]; then
  echo "build.sh: EXTRA_FLAGS is refused for the in-tree library (set TG_LAB_BUILD=1 OUT=<dir> for a lab build)" >&2
  exit 1
fi
# objects are reused only when they were compiled with the same flags
STAMP="$OBJDIR/.build_flags"
if [ ! -f "$STAMP" ] || [ "$(cat "$STAMP")" != "$FLAGS" ]; then
  rm -f "$OBJDIR"/tg_*.o
  echo "$FLAGS" > "$STAMP"
fi
# per-file flags: a line `// TG_FILE_FLAGS: ...` in a source (e.g. -fno-slp-vectorize for a hand-scheduled loop)
FILE_FLAGS_STR=""
for f in tg_*.hip; do
  ff=$(sed -n 's/^\/\/ TG_FILE_FLAGS: *//p' "$f" | head -1)
  [ -z "$ff" ] || FILE_FLAGS_STR="$FILE_FLAGS_STR$f: $ff; "
done
OBJS=()
PIDS=()
for f in tg_*.hip; do
  o="$OBJDIR/${f%.hip}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ tg_common.h -nt "$o" ] || [ ../../include/tecogan_hip.h -nt "$o" ] || [ build.sh -nt "$o" ]; then
    ff=$(sed -n 's/^\/\/ TG_FILE_FLAGS: *//p' "$f" | head -1)
    echo "hipcc $f $ff"
    rm -f "$o"                       # a failed compile must not leave a stale object to link
    extra=()
    [ "$f" != tg_api.hip ] || extra=("-DTG_BUILD_FLAGS_STR=\"$FLAGS\"" "-DTG_FILE_FLAGS_STR=\"$FILE_FLAGS_STR\"")
    # the kernel-resource remarks of every file are kept: the check below reads them
    $HIPCC $FLAGS $ff "${extra[@]}" -Rpass-analysis=kernel-resource-usage -c "$f" -o "$o" 2> "$OBJDIR/${f%.hip}.remarks" &
    PIDS+=($!)
  fi
  OBJS+=("$o")
done
fail=0
for pid in "${PIDS[@]:-}"; do      # bare `wait` returns 0 even when a job failed
  [ -z "$pid" ] || wait "$pid" || fail=1
done
for f in tg_*.hip; do              # warnings / errors (everything that is not a resource remark)
  r="$OBJDIR/${f%.hip}.remarks"
  [ ! -f "$r" ] || grep -v "kernel-resource-usage\|^ *[0-9]* | \|^ *| *\^\|^[0-9]* warnings\? generated" "$r" >&2 || true
done
[ $fail = 0 ] || { echo "build.sh: a compile job failed" >&2; exit 1; }
# Kernels whose hand-written asm the register allocator must not disturb: the LDS-resident SRNet body issues
# its weight loads and their waits in inline asm the compiler cannot see -- a spill or scratch use between a request
# and its s_waitcnt would read registers that have not landed.  No scratch, no spilled registers, or no library.
python3 - "$OBJDIR" <<'PY'
import re, sys, os
d = sys.argv[1]
must_be_clean = {'tg_conv3x3_wino_res': ['conv3x3_wino_resident_kernel']}
rows, bad = [], []
for fn in sorted(os.listdir(d)):
    if not fn.endswith('.remarks'):
        continue
    cur = None
    for ln in open(os.path.join(d, fn), errors='replace'):
        m = re.search(r'remark: Function Name: (\S+)', ln)
        if m:
            cur = {'file': fn[:-8], 'kernel': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass-analysis', ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
with open(os.path.join(d, '.kernel_resources.tsv'), 'w') as out:
    out.write('file\tkernel\tVGPRs\tAGPRs\tSGPRs\tscratch_bytes_per_lane\tVGPR_spill\tSGPR_spill\toccupancy_waves_per_SIMD\n')
    for r in rows:
        out.write('\t'.join([r['file'], r['kernel'], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('TotalSGPRs', '?'),
                             r.get('ScratchSize [bytes/lane]', '?'), r.get('VGPRs Spill', '?'), r.get('SGPRs Spill', '?'),
                             r.get('Occupancy [waves/SIMD]', '?')]) + '\n')
for f, kernels in must_be_clean.items():
    for k in kernels:
        hit = [r for r in rows if r['file'] == f and k in r['kernel']]
        if not hit and os.path.isfile(os.path.join(d, f + '.remarks')):
            bad.append(f'{k}: no resource remark found in {f}.remarks')
        for r in hit:
            if r.get('ScratchSize [bytes/lane]') != '0' or r.get('VGPRs Spill') != '0' or r.get('SGPRs Spill') != '0':
                bad.append(f"{k}: scratch {r.get('ScratchSize [bytes/lane]')} B/lane, VGPR spill {r.get('VGPRs Spill')}, "
                           f"SGPR spill {r.get('SGPRs Spill')} -- the hand-written vmcnt waits are unsafe with spills")
if bad:
    print('build.sh: ' + '; '.join(bad), file=sys.stderr)
    sys.exit(1)
PY
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$LIB" "${OBJS[@]}" -ldl
echo "built $(cd "$(dirname "$LIB")" && pwd)/$(basename "$LIB")"
