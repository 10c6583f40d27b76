#!/bin/bash
# register / scratch / LDS use of the kernels in telr_amd/libtelrhip.so whose names match the given patterns (default: all)
# usage: tools/kernel_regs.sh [pattern ...]
set -e
D=$(mktemp -d); SO=${SO:-$(dirname "$0")/../telr_amd/libtelrhip.so}
objcopy -O binary --only-section=.hip_fatbin "$SO" $D/fat.bin
B=/opt/rocm/lib/llvm/bin
T=$($B/clang-offload-bundler --list --type=o --input=$D/fat.bin | grep gfx950)
$B/clang-offload-bundler --unbundle --type=o --input=$D/fat.bin --targets=$T --output=$D/dev.co
$B/llvm-readelf --notes $D/dev.co | python3 -c "
import sys, re
pats = sys.argv[1:]
txt = sys.stdin.read()
for blk in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    n = g('name')
    if pats and not any(p in n for p in pats): continue
    print('%-70s vgpr %3s sgpr %3s scratch %4s lds %6s' % (n[:70], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
" "$@"
rm -rf $D
