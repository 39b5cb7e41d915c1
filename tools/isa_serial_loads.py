"""Which global loads of a kernel are ALONE in flight when the wave waits for them (s_waitcnt vmcnt(0)) — i.e. dependent round trips the source
did not ask for (the compiler sinks loads behind branches: DESIGN 9c.8).  Input: the device assembly of a translation unit compiled with
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -gline-tables-only -Iinclude --save-temps=obj -c lvio_fusion_amd/csrc/<tu>.hip
usage: python tools/isa_serial_loads.py <kernel name substring> [path to the *-gfx950.s]      prints (source line, count) pairs"""
import re,sys,collections
s=open(sys.argv[2] if len(sys.argv) > 2 else '/tmp/isag/solver_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read()
funcs=[(m.start(), m.group(1)) for m in re.finditer(r'\n(_Z\w+):\s*; @', s)]
want=sys.argv[1]
for k,(pos,name) in enumerate(funcs):
    if want in name and '_b' not in name.split(want)[1][:3]:
        end=funcs[k+1][0] if k+1<len(funcs) else len(s)
        lines=s[pos:end].split('\n')
        cur=None; pending=[]; out=collections.Counter()
        for l in lines:
            t=l.strip()
            m=re.match(r'\.loc\s+\d+\s+(\d+)\s+(\d+)', t)
            if m: cur=int(m.group(1)); continue
            if t.startswith('global_load') or t.startswith('flat_load'):
                pending.append(cur)
            elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                if len(pending)==1: out[pending[0]]+=1
                pending=[]
        print(name[:60], sorted(out.items()))
        break
