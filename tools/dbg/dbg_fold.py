import sys; sys.path.insert(0,'tests')
import numpy as np, zkgpu_loader, oracle_lib
zk=zkgpu_loader.load(); zk.init(0); orc=oracle_lib.load(); P=zk.P
for pol_bits,step_bits in ((3,2),(4,2),(5,2),(6,2),(7,2),(8,2)):
    n2=1<<step_bits; nx=1<<(pol_bits-step_bits)
    pol=np.zeros(3<<pol_bits,np.uint64)
    # coefficient polynomial = X (c1 = 1): values v_i = w_nx^i
    wnx=orc.root(pol_bits-step_bits)
    for g in range(n2):
        for i in range(nx):
            pol[(i*n2+g)*3]=pow(wnx,i,P)
    sx=np.array([1,0,0],np.uint64); shift_inv=pow(49,P-2,P)
    got=zk.fri_fold(zk.DevArray.from_host(pol),pol_bits,step_bits,zk.DevArray.from_host(sx),shift_inv).to_host().reshape(-1,3)
    wi=pow(orc.root(pol_bits),P-2,P)
    exp=[shift_inv*pow(wi,g,P)%P for g in range(n2)]
    print(pol_bits,step_bits,[int(v[0])==e for v,e in zip(got,exp)], [hex(int(v[0])) for v in got[:3]], [hex(e) for e in exp[:3]])
    o=orc.fri_fold(pol,pol_bits,step_bits,sx,shift_inv).reshape(-1,3)
    print('   oracle ok', [int(v[0])==e for v,e in zip(o,exp)])
