#!/bin/bash
# alternative libraries for the A/B of non-temporal accesses in EVERY compiled whole-row schedule (kernels_smooth*_f{32,64}_*): the units are
# copied with the ABL argument of their MI_K1 / MI_K1X lines OR-ed with $1 (16 = loads, 48 = loads + stores), compiled like the originals and
# linked with the shipped objects -> rustfft_amd/lib/libmi355fft_snt$1.so
set -eu
NT=$1
cd "$(dirname "$0")/../../rustfft_amd/csrc"
D=build_snt$NT
mkdir -p $D
NOSLP=$(grep "^NOSLP :=" Makefile | cut -d= -f2)
jobs=0
for f in kernels_smooth*_f32_*.hip kernels_smooth*_f64_*.hip; do
  u=${f%.hip}
  python3 - "$f" "$D/$u.hip" "$NT" <<'PY'
import re,sys
src,dst,nt=sys.argv[1],sys.argv[2],int(sys.argv[3])
s=open(src).read()
s=re.sub(r'MI_K1X\((\w+), (\d+), (\d+), (true|false), (\d+), "(\w*)",', lambda m: f'MI_K1X({m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}, {int(m.group(5))|nt}, "{m.group(6)}",', s)
s=re.sub(r'MI_K1\((\w+), (\d+), (\d+), (true|false),', lambda m: f'MI_K1X({m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}, {nt}, "",', s)
open(dst,'w').write(s)
PY
  extra=""
  case " $NOSLP " in *" $u "*) extra="-fno-slp-vectorize";; esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-value --offload-compress $extra -c $D/$u.hip -o $D/$u.o ) &
  jobs=$((jobs+1))
  if [ $jobs -ge 8 ]; then wait -n; jobs=$((jobs-1)); fi
done
wait
OBJS=$(ls build/*.o | grep -v "kernels_smooth")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmi355fft_snt$NT.so $OBJS $D/*.o
ls -la ../lib/libmi355fft_snt$NT.so
