import sys; sys.path.insert(0,'tests')
import numpy as np, zkgpu_loader, oracle_lib
zk=zkgpu_loader.load(); zk.init(0); orc=oracle_lib.load(); P=zk.P
pol_bits,step_bits=3,2
rng=np.random.default_rng(pol_bits*100+step_bits)
pol=rng.integers(0,P,size=3<<pol_bits,dtype=np.uint64); sx=rng.integers(0,P,size=3,dtype=np.uint64)
for name,si in (("49^-8",pow(pow(49,P-2,P),8,P)),("49^-1",pow(49,P-2,P)),("1",1)):
    for sxv in (sx, np.array([1,0,0],np.uint64), np.array([int(sx[0]),0,0],np.uint64), np.array([0,1,0],np.uint64)):
        got=zk.fri_fold(zk.DevArray.from_host(pol),pol_bits,step_bits,zk.DevArray.from_host(sxv),si).to_host().reshape(-1,3)
        exp=orc.fri_fold(pol,pol_bits,step_bits,sxv,si).reshape(-1,3)
        print(name, [int(v) for v in sxv][:3], (got==exp).tolist())
# isolate f3 ops: y = sx * sinv -> use pol giving result = y (c1=2,c0=0)
pol2=np.zeros(3<<pol_bits,np.uint64)
n2=4
for g in range(n2):
    pol2[(0*n2+g)*3]=1; pol2[(1*n2+g)*3]=P-1
si=pow(pow(49,P-2,P),8,P)
got=zk.fri_fold(zk.DevArray.from_host(pol2),pol_bits,step_bits,zk.DevArray.from_host(sx),si).to_host().reshape(-1,3)
exp=orc.fri_fold(pol2,pol_bits,step_bits,sx,si).reshape(-1,3)
print('y only', (got==exp).tolist())
