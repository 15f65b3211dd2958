#!/bin/bash
# kinfo.sh <object.o> [kernel-name-substring]: VGPR/SGPR/spill/LDS metadata of the gfx950 kernels in an object, and the
# disassembly written next to it (/tmp/kinfo/<name>.s)
set -e
o=$1; pat=${2:-}
d=/tmp/kinfo; mkdir -p $d; b=$(basename $o .o)
objcopy -O binary --only-section=.hip_fatbin $o $d/$b.fat
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$d/$b.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$d/$b.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d $d/$b.co > $d/$b.s
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $d/$b.co | python3 -c "
import sys,re
cur={}
def flush():
    if cur.get('name') and '$pat' in cur['name']:
        print(cur['name'][:90], 'vgpr',cur.get('vgpr_count'),'agpr',cur.get('agpr_count'),'sgpr',cur.get('sgpr_count'),'vspill',cur.get('vgpr_spill_count'),'sspill',cur.get('sgpr_spill_count'),'lds',cur.get('group_segment_fixed_size'),'scratch',cur.get('private_segment_fixed_size'))
for l in sys.stdin:
    m=re.match(r'\s+(- )?\.(\w+):\s+(\S+)',l)
    if not m: continue
    if m.group(2)=='args' or (m.group(1) and m.group(2) in ('agpr_count','args')): pass
    k,v=m.group(2),m.group(3)
    if k=='agpr_count' and 'name' in cur and 'vgpr_count' in cur: flush(); cur={}
    if k in ('name','vgpr_count','agpr_count','sgpr_count','vgpr_spill_count','sgpr_spill_count','group_segment_fixed_size','private_segment_fixed_size'):
        if k=='name' and 'name' in cur and 'vgpr_count' in cur: flush(); cur={}
        if k=='name' and v.startswith('_Z')==False and 'name' in cur: continue
        cur[k]=v
flush()
"
