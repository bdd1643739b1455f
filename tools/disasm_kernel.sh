#!/bin/bash
# usage: tools/disasm_kernel.sh <substring of the mangled kernel name> [out.s]
# extracts the gfx950 code objects of the built library and disassembles the kernels whose name contains the substring;
# prints registers / LDS / scratch from the notes and an instruction-class histogram
set -e
LIB=${LIB:-/root/repo/medicalimageanalysis_amd/libvmf_hip.so}
T=$(mktemp -d)
cp "$LIB" $T/lib.so
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1)
for f in $T/*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | python3 -c "
import sys,re
pat=sys.argv[1]
rec={}
def flush():
    if pat in rec.get('name',''):
        print(rec['name'],{k:rec.get(k) for k in ('vgpr_count','sgpr_count','group_segment_fixed_size','private_segment_fixed_size','vgpr_spill_count')})
for line in sys.stdin:
    line=line.strip()
    if line.startswith('- .'):
        flush(); rec={}; line=line[2:]
    m=re.match(r'\.(\w+):\s*(.*)',line)
    if m: rec[m.group(1)]=m.group(2)
flush()
" "$1"
  for sym in $(/opt/rocm/lib/llvm/bin/llvm-readelf --symbols --wide $f | awk '$4=="FUNC"{print $8}' | grep "$1"); do
    /opt/rocm/lib/llvm/bin/llvm-objdump -d $f --disassemble-symbols=$sym > ${2:-$T/out.s}
    echo "$sym: $(grep -cE '^\s+[a-z_0-9]+ ' ${2:-$T/out.s}) instructions"
    grep -E '^\s+[a-z_0-9]+ ' ${2:-$T/out.s} | awk '{print $1}' | sed -E 's/_e32|_e64|_dpp|_sdwa//' | sort | uniq -c | sort -rn | head -${TOP:-25}
  done
done
rm -rf $T
