import re,sys
from collections import Counter
t=open(sys.argv[1]).read()
m=re.search(r'^_Z10k_attend32ILi0EEvPK15HIP_vector_typeIjLj4EES3_S3_PKfPfi:(.*?)^\.Lfunc_end', t, re.S|re.M)
body=m.group(1)
ops=[l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.',';'))]
c=Counter(ops); print(len(ops), [(k,v) for k,v in c.most_common(16)])
def short(o):
    if 'mfma' in o: return 'M'
    if o.startswith('v_exp'): return 'e'
    if o.startswith('v_cvt_pk'): return 'c'
    if o.startswith('v_max'): return 'x'
    if o.startswith('ds_read'): return 'L'
    if o.startswith('s_waitcnt'): return 'w'
    if o.startswith('s_nop'): return 'n'
    if o.startswith('v_accvgpr'): return 'a'
    if o.startswith('s_cbranch') or o.startswith('s_branch'): return '|'
    if o.startswith('v_'): return 'v'
    if o.startswith('s_'): return 's'
    return '?'
tr=''.join(short(o) for o in ops)
for i in range(0,len(tr),160): print(tr[i:i+160])
for k in ('vgpr','accum','scratch','spill'):
    for l in re.findall(r'^\s*\.amdhsa_\w*%s\w*\s+\S+'%k, t, re.M)[:2]: print(l.strip())
