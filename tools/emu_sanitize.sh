#!/bin/bash
# The emulated kernel suite (tests/test_emu_parity.py: the product's kernel sources on tests/emu's CPU model, one OS thread per
# HIP thread) under a sanitizer:   tools/emu_sanitize.sh address|undefined|thread [pytest -k expression]
# address / undefined: out-of-range LDS and global accesses, index arithmetic; thread: LDS hand-overs between waves without a
# block barrier (the one expected report: dc_backward.h's turn map, an intended last-writer-wins store).  CPU only; ~2-25 min.
set -e
cd "$(dirname "$0")/.."
S=${1:?address|undefined|thread}
K=${2:-}
case $S in
  address) LIB=libasan.so; OPT="ASAN_OPTIONS=detect_leaks=0";;
  undefined) LIB=libubsan.so; OPT="UBSAN_OPTIONS=print_stacktrace=1";;
  thread) LIB=libtsan.so; OPT="TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:history_size=4";;
  *) echo "unknown sanitizer $S"; exit 2;;
esac
EXTRA=""
[ $S = undefined ] && EXTRA="-fno-sanitize=vptr,float-cast-overflow"
(cd tests/emu && g++ -O1 -g -fsanitize=$S $EXTRA -fno-omit-frame-pointer -std=c++20 -fPIC -shared -pthread -DMFN_EMU -I . \
   -Wno-unused-but-set-variable -o libmfn_emu.so emu_api.cpp)
trap 'python tests/emu/build_emu.py > /dev/null' EXIT   # the plain build comes back whatever happens
env LD_PRELOAD=$(g++ -print-file-name=$LIB) $OPT python -m pytest tests/test_emu_parity.py -q -s ${K:+-k "$K"} 2>&1 |
  grep -E "ERROR: AddressSanitizer|runtime error|WARNING: ThreadSanitizer|#0 mfn::|passed|failed" | sed -E 's/\(libmfn_emu.so[^)]*\)//g' | sort | uniq -c | sort -rn | head -40
