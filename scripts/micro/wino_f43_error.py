import numpy as np
F=np.float32
BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float64)
G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float64)
def run(Cin=256, Cout=64, T_=64, seed=0, scale_trick=False):
    rng=np.random.default_rng(seed)
    d=rng.standard_normal((T_,Cin,6,6)); d=np.maximum(d,0.01*d)
    g=rng.standard_normal((Cout,Cin,3,3))*np.sqrt(2/(9*Cin))
    ref=np.zeros((T_,Cout,4,4))
    for i in range(4):
        for j in range(4):
            ref[:,:,i,j]=np.einsum('tcxy,kcxy->tk', d[:,:,i:i+3,j:j+3], g)
    U=np.einsum('ar,kcrs,bs->kcab',G,g,G)
    V=np.einsum('ax,tcxy,by->tcab',BT,d,BT)
    M=np.einsum('tcab,kcab->tkab',V,U)
    Y=np.einsum('ia,tkab,jb->tkij',AT,M,AT)
    chk=np.abs(Y-ref).max()
    U32=U.astype(F); d32=d.astype(F); BT32=BT.astype(F); AT32=AT.astype(F)
    V32=np.einsum('ax,tcxy->tcay',BT32,d32).astype(F); V32=np.einsum('tcay,by->tcab',V32,BT32).astype(F)
    M32=np.zeros((T_,Cout,6,6),F)
    for c in range(Cin):
        M32+= V32[:,None,c]*U32[None,:,c]
    Y32=np.einsum('ia,tkab->tkib',AT32,M32).astype(F); Y32=np.einsum('tkib,jb->tkij',Y32,AT32).astype(F)
    e=np.abs(Y32-ref)
    # F(2,3) for comparison
    BT2=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64); G2=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]); AT2=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
    U2=np.einsum('ar,kcrs,bs->kcab',G2,g,G2).astype(F)
    e2=[]
    for (oi,oj) in [(0,0),(2,2)]:
        dd=d32[:,:,oi:oi+4,oj:oj+4]
        V2=np.einsum('ax,tcxy->tcay',BT2.astype(F),dd).astype(F); V2=np.einsum('tcay,by->tcab',V2,BT2.astype(F)).astype(F)
        M2=np.zeros((T_,Cout,4,4),F)
        for c in range(Cin): M2+=V2[:,None,c]*U2[None,:,c]
        Y2=np.einsum('ia,tkab->tkib',AT2.astype(F),M2).astype(F); Y2=np.einsum('tkib,jb->tkij',Y2,AT2.astype(F)).astype(F)
        e2.append(np.abs(Y2-ref[:,:,oi:oi+2,oj:oj+2]).max())
    print('Cin %d: f64 check %.1e | F(4,3) f32 max err %.3e rms %.3e | F(2,3) f32 max err %.3e | out rms %.3f max %.2f'%(Cin,chk,e.max(),np.sqrt((e**2).mean()),max(e2),np.sqrt((ref**2).mean()),np.abs(ref).max()))
run(128); run(256); run(512)
