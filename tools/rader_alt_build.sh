#!/bin/bash
# Builds rustfft_amd/lib/libmi355fft_alt.so: the shipped library with the Rader kernel lists of an experiment of
# tools/gen_rader_kernels.py (RADER_ALT=1 rows loop wherever it can be instantiated, 2 one butterfly per thread, 3 rows side by side
# wherever the layout allows, 5 every side-by-side body with the register hand-over), for tools/ab_lengths.py --b libmi355fft_alt.so
# and tools/prime_sweep.py --lib.  Only the eight generated Rader translation units differ; their objects go to csrc/build_alt, every
# other object is copied from csrc/build (build the shipped library first).  The tracked lists are restored afterwards.
#   bash tools/rader_alt_build.sh 5
set -eu
ALT=${1:?usage: rader_alt_build.sh <RADER_ALT value>}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/rustfft_amd/csrc
test -d $CSRC/build || { echo "build the shipped library first (make -C rustfft_amd/csrc)"; exit 1; }
rm -rf $CSRC/build_alt && cp -a $CSRC/build $CSRC/build_alt
stamp=$(mktemp)
for f in $CSRC/kernels_rader_*.hip; do echo "$(stat -c %Y $f) $f" >> $stamp; done
RADER_ALT=$ALT python $ROOT/tools/gen_rader_kernels.py
rc=0
make -C $CSRC BUILD=build_alt LIBNAME=libmi355fft_alt.so -j"$(nproc)" -s || rc=$?
python $ROOT/tools/gen_rader_kernels.py > /dev/null       # the shipped choice again (identical content) ...
while read t f; do touch -d @$t $f; done < $stamp           # ... with the old time stamps, so the shipped build stays up to date
rm -f $stamp
exit $rc
