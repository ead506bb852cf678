#!/bin/bash
# VGPR / SGPR / scratch / LDS of the kernels in a built library (from the notes of its gfx950 code objects).
# usage: bash tools/kernel_regs.sh [lib] [name filter]
LIB=$(readlink -f "${1:-polishpathplanning_amd/libppp_hip.so}"); PAT=${2:-k_win}
TMP=$(mktemp -d); cd $TMP || exit 2
objcopy --dump-section .hip_fatbin=fb.bin "$LIB" || exit 2
python3 - <<'PY'
data = open("fb.bin", "rb").read()
magics = [b"__CLANG_OFFLOAD_BUNDLE__", b"CCOB"]
pos = sorted(i for mg in magics for i in range(len(data)) if data.startswith(mg, i)) if len(data) < (1 << 20) else None
if pos is None:
    pos, i = [], 0
    while True:
        js = [j for j in (data.find(mg, i) for mg in magics) if j >= 0]
        if not js: break
        pos.append(min(js)); i = min(js) + 4
for n, p in enumerate(pos):
    open("bundle%d.bin" % n, "wb").write(data[p:(pos[n + 1] if n + 1 < len(pos) else len(data))])
PY
for b in bundle*.bin; do
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$b --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$b.co 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $b.co | PAT="$PAT" python3 -c "
import sys, re, os
pat = os.environ['PAT']
txt = sys.stdin.read()
for blk in re.split(r'\n\s*- \.agpr_count', txt)[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk)
    if not name or pat not in name.group(1): continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', blk) or [0, '?'])[1]
    print('%-64s vgpr %3s sgpr %3s spills %s scratch %s static-lds %s' % (name.group(1)[:64], g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
"
done
cd /; rm -rf $TMP
