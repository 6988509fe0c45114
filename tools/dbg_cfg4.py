import os,sys,numpy as np,torch
sys.path.insert(0,os.getcwd()); sys.path.insert(0,'tests/support')
import forgex_amd as fx, oracle_lib
from forgex_amd import synth
pat=synth.PATTERNS['cfg4'].encode()
rows=synth.batch('cfg4',0,6000,torch.device('cuda'))
p=fx.Program(pat,fx.OP_SEARCH)
f,a,b=p.match_device(rows); torch.cuda.synchronize()
f,a,b=f.cpu().numpy(),a.cpu().numpy(),b.cpu().numpy()
rn=rows.cpu().numpy()
of,oa,ob=oracle_lib.batch(2,pat,rn,32)
bad=np.nonzero((f!=of)|(a!=oa)|(b!=ob))[0]
print('mismatches',len(bad),'path',p.last_path())
for i in bad[:6]:
    r=rn[i]; print(i,'gpu',f[i],a[i],b[i],'oracle',of[i],oa[i],ob[i],'nonascii',int((r>=128).sum()),'tileidx',i//64, 'head',bytes(r[:12]).hex(),'around_to',bytes(r[max(0,ob[i]-6):ob[i]+4]).hex())
