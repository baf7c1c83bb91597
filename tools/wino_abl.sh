for a in ${ABLS:-0 64 32}; do echo "ABL $a"; TG_WINO_ABL=$a bash tools/wino_prof.sh 134 320 64 64 2>&1 | grep -E "wino_kernel|err"; done
